#!/bin/bash
# Core clock per kernel inside a train step: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) over the dispatch's
# duration from the same counter run.   tools/clock_probe.sh [workload]
set -o pipefail
export TMPDIR=/tmp
w=${1:-protein256}
out=gpurun_out/clock
rm -rf $out; mkdir -p $out
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/run -o p --output-format csv -- ${CMD:-python3 bench.py --workload $w --steps 3 --warmup 2 --headline-only --no-cpu-baseline} > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
cc = glob.glob("$out/run/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob("$out/run/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(cc)))
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if d is None:
        if "Start_Timestamp" in r: d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        else: continue
    m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
    key = m.group(1) if m else r["Kernel_Name"][:50]
    a = acc[key]; a[0] += float(r["Counter_Value"]) / 8; a[1] += d; a[2] += 1          # one row per dispatch: the counter summed over the 8 XCDs
with open("$out/clock.txt", "w") as f:
    for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        if ns > 0 and ns / n > 20000:
            line = f"{k:44s} n={n:4d} avg {ns / n / 1e3:8.1f} us  {cyc / ns:5.2f} GHz"
            print(line); f.write(line + "\n")
PY
