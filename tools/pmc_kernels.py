#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter values per kernel: pmc_kernels.py <counter_collection.csv> [name filter]."""
import collections
import csv
import re
import sys


def main():
    path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        m = re.search(r"(k_\w+(<[^>]*>)?)", name)
        key = (m.group(1) if m else name[:60]) + f" grid={r['Grid_Size']} wg={r['Workgroup_Size']} lds={r.get('LDS_Block_Size', '?')} vgpr={r.get('VGPR_Count', '?')}"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, d in acc.items():
        n = max(len(v) for v in d.values())
        print(f"{key}  launches={n}")
        for c, v in sorted(d.items()):
            print(f"    {c:36s} {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main()
