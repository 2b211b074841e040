#!/bin/bash
# Counter summaries (tools/pmc_bench.py) of the workloads bench.py prices with them; results in gpurun_out/pmc/ (copy into profiles/).
set -o pipefail
export TMPDIR=/tmp
python3 tools/pmc_bench.py --workload protein32 || exit 1
echo "protein32 done"
python3 tools/pmc_bench.py --workload protein256 || exit 1
echo "protein256 done"
python3 tools/pmc_bench.py --workload large --agg-kernel 'k_aggregate<4, false, 4, 2>' --agg-width 128 || exit 1
echo "large done"
python3 tools/pmc_bench.py --workload attention32 --agg-kernel 'k_aggregate<4, false, 4, 2>' --agg-width 256 || exit 1
echo "attention32 done"
python3 tools/pmc_bench.py --workload pool || exit 1
echo "pool done"
