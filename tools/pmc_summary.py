#!/usr/bin/env python3
"""Summarises a rocprofv3 --pmc counter CSV per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: [0.0, 0])
for f in files:
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:70], r["Counter_Name"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
for (k, c), (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:70s} {c:12s} dispatches={n:5d} mean={v / n:16.1f}")
