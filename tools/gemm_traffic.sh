#!/bin/bash
# HBM bytes of the projection kernels (FETCH_SIZE / WRITE_SIZE in separate passes) on the probe shapes
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/gtraffic
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c -d $out/$c -o p --output-format csv -- python3 tools/gemm_probe.py > $out/$c.log 2>&1 || exit 1
  python3 tools/pmc_kernels.py $(find $out/$c -name '*counter_collection.csv' | head -1) k_gemm > $out/$c.txt || exit 1
  rm -rf $out/$c
done
grep -A1 "k_gemm" $out/FETCH_SIZE.txt | grep -v "^--" | paste - - | awk '{print $1,$2,$3,$NF}'
echo; grep -A1 "k_gemm" $out/WRITE_SIZE.txt | grep -v "^--" | paste - - | awk '{print $1,$2,$3,$NF}'
