#!/usr/bin/env python3
"""Per-step kernel summary from a rocprofv3 kernel trace of bench.py:
step_profile.py <kernel_trace.csv> [lines] [sequence file]  (uses the last full step: kernels between Adam launches)."""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"(k_\w+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.search(r"(\w+)<", name.replace("void ", "").replace("at::native::", "").replace("(anonymous namespace)::", ""))
    return (m.group(1) if m else name)[:48]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "k_adam" in r["Kernel_Name"]]
    # Adam uses several multi_tensor launches per step; take boundaries where the gap in index is large
    bounds = [adam[0]] + [b for a, b in zip(adam, adam[1:]) if b - a > 20]
    lo, hi = bounds[-2], bounds[-1]
    step = rows[lo:hi]
    t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print(f"kernels in step: {len(step)}  wall {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms")
    acc = collections.defaultdict(lambda: [0, 0])
    for r in step:
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        print(f"{k:52s} x{n:3d}  {ns / 1e3:8.1f} us")
    if len(sys.argv) > 3:       # the step's launches in order: start offset, duration, gap to the previous kernel's end
        with open(sys.argv[3], "w") as f:
            prev = t0
            for r in step:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                f.write(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {(s - prev) / 1e3:6.1f}  {short(r['Kernel_Name'])}\n")
                prev = e


if __name__ == "__main__":
    main()
