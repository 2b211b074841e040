#!/bin/bash
# Copies what tools/refresh_profiles.sh left in gpurun_out/refresh/ into profiles/<round directory> (argument, e.g. r02).
set -e
cd "$(dirname "$0")/.."
r=gpurun_out/refresh; dst=profiles/${1:?round directory}
mkdir -p $dst
for w in protein256 protein32 large chig32 attention attention32 pool custom_pool; do cp $r/bench_$w.json $dst/bench_$w.json; done
for w in chig32 attention pool custom_pool; do cp $r/bench_${w}_graph.json $dst/; done
cp $r/bench_protein32_2ranks.json $dst/bench_protein32_2ranks_gloo_one_gpu.json
for w in protein32 protein256 attention32 pool; do
  cp $r/kt_$w/summary.txt $dst/kernels_${w}_step_summary.txt
  cp $r/kt_$w/seq.txt $dst/kernels_${w}_step_sequence.txt
  cp $r/kt_$w/kt_kernel_stats.csv $dst/kernels_${w}_rocprofv3_stats.csv
done
for f in $dst/bench_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], d['ms_per_step'], round(d['value'],1), r['frac'], r.get('step_hbm_frac'), r.get('cache_gather_lines_frac'))"; done
