#!/bin/bash
# Per-step kernel summary of bench.py under rocprofv3 (kernel trace only): tools/kt_bench.sh [workload] [lines]
# output in gpurun_out/kt_<workload>/ (summary.txt = per-kernel totals of one step, seq.txt = the step's launches in order).
set -o pipefail
w=${1:-protein32}
out=gpurun_out/kt_$w
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --headline-only > $out/bench.json 2> $out/err.log || exit 1
python3 tools/step_profile.py $out/kt_kernel_trace.csv 60 $out/seq.txt > $out/summary.txt || exit 1
rm -f $out/kt_kernel_trace.csv
head -${2:-40} $out/summary.txt
