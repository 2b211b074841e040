#!/bin/bash
# Full-size parity against the float64 oracle for the three level modes (plain / deferred / deferred + algebraic sums).
mkdir -p gpurun_out/r03
out=gpurun_out/r03/full_size_parity_modes.txt
: > $out
for mode in "HGVAE_DEFER=0" "HGVAE_ALG_SUMS=0" "HGVAE_DEFER=1"; do
  echo "== $mode (MODEL=skipcon GRAPHS=${GRAPHS:-16} F64=1)" >> $out
  env $mode MODEL=skipcon GRAPHS=${GRAPHS:-16} F64=1 timeout -k 10 500 python tools/parity_large.py >> $out 2>&1 || exit 1
done
cat $out
