#!/usr/bin/env python3
"""Median duration of consecutive runs of the same kernel in a rocprofv3 kernel trace:
kt_runs.py <kernel_trace.csv> [substring ...]"""
import csv
import re
import sys


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    rows = [r for r in csv.DictReader(open(path)) if not subs or any(s in r["Kernel_Name"] for s in subs)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    runs = []
    for r in rows:
        m = re.search(r"(k_\w+(<[^>]*>)?)", r["Kernel_Name"])
        key = (m.group(1) if m else r["Kernel_Name"][:50]) + f" grid={r['Grid_Size_X']}"
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if runs and runs[-1][0] == key:
            runs[-1][1].append(d)
        else:
            runs.append((key, [d]))
    for key, v in runs:
        v.sort()
        print(f"{key:52s} n={len(v):3d} median={v[len(v) // 2]:8.1f} us")


if __name__ == "__main__":
    main()
