#!/bin/bash
# Per-step kernel summary of bench.py under rocprofv3 (kernel trace only); output in gpurun_out/kt/.
set -o pipefail
out=gpurun_out/kt
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --cpu-sample-graphs 0 > $out/bench.json 2> $out/err.log || exit 1
python3 tools/step_profile.py $out/kt_kernel_trace.csv > $out/summary.txt || exit 1
rm -f $out/kt_kernel_trace.csv
head -${1:-30} $out/summary.txt
