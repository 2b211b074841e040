#!/bin/bash
# Regenerates the evidence kept under profiles/ on a GPU box; outputs land in gpurun_out/refresh/.
set -o pipefail
out=gpurun_out/refresh
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 200 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --cpu-sample-graphs 0 > $out/bench_under_rocprof.json 2> $out/kt.err || exit 1
for p in bf16x6 fp32; do
  HGVAE_PROJECTION=$p timeout -k 10 120 python3 tools/gemm_probe.py > $out/gemm_probe_$p.txt 2>/dev/null || exit 1
done
HGVAE_PROJECTION=bf16x6 timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $out/pmc -o pmc --output-format csv -- python3 tools/gemm_probe.py > $out/pmc.log 2>&1 || exit 1
f=$(find $out/pmc -name '*counter_collection.csv' | head -1)
python3 tools/pmc_kernels.py "$f" k_gemm > $out/gemm_pmc.txt || exit 1
find $out/kt -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/kt/*/*kernel_trace.csv $out/pmc
cat $out/bench.json | cut -c1-300
cat $out/gemm_probe_bf16x6.txt
