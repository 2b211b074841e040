#!/bin/bash
# Full-size parity against the float64 oracle for the level modes: every matrix written (plain) / deferred, relu -> dropout
# and statistics in a pass of their own, sums over the gradient / the same with the backward sums from the consumers'
# products / the projection stores the activated level, statistics pass reads it / the level backward in passes of its own /
# default (column sums inside the product, level backward inside the consumers' grad-input products).
# Arguments, if any, replace the list of modes: tools/parity_modes.sh "HGVAE_BWD_FUSE=0" "HGVAE_DEFER=1"
mkdir -p gpurun_out/r03
out=gpurun_out/r03/full_size_parity_modes.txt
: > $out
modes=("HGVAE_DEFER=0" "HGVAE_NT_ACT=0 HGVAE_ALG_SUMS=0" "HGVAE_NT_ACT=0" "HGVAE_NT_STATS=0" "HGVAE_BWD_FUSE=0" "HGVAE_DEFER=1")
if [ $# -gt 0 ]; then modes=("$@"); fi
for mode in "${modes[@]}"; do
  echo "== $mode (MODEL=skipcon GRAPHS=${GRAPHS:-16} F64=1)" >> $out
  env $mode MODEL=skipcon GRAPHS=${GRAPHS:-16} F64=1 timeout -k 10 500 python tools/parity_large.py >> $out 2>&1 || exit 1
done
cat $out
