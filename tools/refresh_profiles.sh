#!/bin/bash
# Regenerates the evidence kept under profiles/ on a GPU box; outputs land in gpurun_out/refresh/ (copy what is to be
# published into profiles/).  One bench line per workload, a rocprofv3 kernel summary of the headline workloads, the
# two-rank rehearsal (two ranks share the GPU over gloo when the box has one).
set -o pipefail
out=gpurun_out/refresh
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
for w in protein256 protein32 large chig32 attention attention32 pool custom_pool; do
  timeout -k 10 300 python3 bench.py --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || exit 1
  echo "bench $w done"
done
for w in chig32 attention pool custom_pool; do
  timeout -k 10 300 python3 bench.py --workload $w --graph --no-cpu-baseline > $out/bench_${w}_graph.json 2> $out/bench_${w}_graph.err || exit 1
  echo "bench $w --graph done"
done
timeout -k 10 300 python3 bench.py --gpus 2 --workload protein32 --steps 10 --warmup 3 > $out/bench_protein32_2ranks.json 2> $out/bench_protein32_2ranks.err || exit 1
echo "2-rank rehearsal done"
for w in protein32 protein256 attention32 pool; do
  d=$out/kt_$w; mkdir -p $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o kt --output-format csv -- python3 bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --headline-only > $d/bench.json 2> $d/err.log || exit 1
  python3 tools/step_profile.py $d/kt_kernel_trace.csv 60 $d/seq.txt > $d/summary.txt || exit 1
  rm -f $d/kt_kernel_trace.csv
  echo "kernel trace $w done"
done
