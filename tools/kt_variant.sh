#!/bin/bash
# Kernel summary of one model variant's train step (tools/variants_time.py) under rocprofv3: kt_variant.sh <model> [graphs] [lines]
set -o pipefail
out=gpurun_out/ktv_$1
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp MODELS=$1 GRAPHS=${2:-32}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o kt --output-format csv -- python3 tools/variants_time.py > $out/time.txt 2> $out/err.log || exit 1
cat $out/time.txt | cut -c1-100
python3 - $out/kt_kernel_stats.csv ${3:-25} <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
for r in rows[:n]:
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Name"])
    name = m.group(1) if m else r["Name"][:70]
    print(f"{name[:60]:60s} calls={int(r['Calls']):5d} avg={float(r['AverageNs'])/1e3:8.1f}us total%={r['Percentage']}")
PY
rm -f $out/kt_kernel_trace.csv
