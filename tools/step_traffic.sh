#!/bin/bash
# HBM bytes moved by one train step of bench.py: counter totals of a 22-step run minus those of a 12-step run, / 10
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/straffic
rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  for n in 10 20; do
    timeout -s KILL 200 rocprofv3 --pmc $c -d $out/$c$n -o p --output-format csv -- python3 bench.py --steps $n --warmup 2 --cpu-sample-graphs 0 > $out/$c$n.log 2>&1 || exit 1
    python3 - $(find $out/$c$n -name '*counter_collection.csv' | head -1) $c > $out/$c$n.txt <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        tot += float(r["Counter_Value"])
print(tot)
PY
    rm -rf $out/$c$n
  done
done
python3 - <<'PY'
o = "gpurun_out/straffic/"
f = (float(open(o + "FETCH_SIZE20.txt").read()) - float(open(o + "FETCH_SIZE10.txt").read())) / 10 * 1024 * 2   # KB -> bytes, x2: 16-byte reads count half
w = (float(open(o + "WRITE_SIZE20.txt").read()) - float(open(o + "WRITE_SIZE10.txt").read())) / 10 * 1024
print(f"per step: HBM read {f/1e9:.2f} GB (FETCH_SIZE x 2), written {w/1e9:.2f} GB, total {(f+w)/1e9:.2f} GB")
PY
