#!/bin/bash
# HBM bytes of the aggregation kernel from the PMC counters (separate passes); result in gpurun_out/aggpmc/agg_pmc.json
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/aggpmc
rm -rf $out; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $out/f -o f --output-format csv -- python3 tools/agg_probe.py --workload protein256 > $out/f.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $out/w -o w --output-format csv -- python3 tools/agg_probe.py --workload protein256 > $out/w.log 2>&1 || exit 1
python3 tools/agg_pmc.py $(find $out/f -name '*counter_collection.csv' | head -1) $(find $out/w -name '*counter_collection.csv' | head -1) $out/agg_pmc.json || exit 1
grep -E "median" $out/f.log
rm -rf $out/f $out/w
