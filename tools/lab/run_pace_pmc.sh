#!/bin/bash
# memory reads of the two-part NT kernel with and without pacing (FETCH_SIZE in KB, 23 launches each)
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/pace_pmc
for v in never always; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pace_pmc/$v -o p --output-format csv -- tools/bin/nt16_$v 1285158 138 138 138 $v > gpurun_out/pace_pmc/$v.log 2>&1 || exit 1
done
python3 - <<'P'
import csv, glob, collections
for v in ("never", "always"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pace_pmc/{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in acc.items():
        print(v, k, "per launch: %.3f GB (x2 correction applied)" % (s / n * 1024 * 2 / 1e9), n)
P
