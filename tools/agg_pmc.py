#!/usr/bin/env python3
"""Builds profiles/agg_pmc_<workload>.json from two rocprofv3 counter runs of tools/agg_probe.py:
  agg_pmc.py <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> <out.json> [workload]
Counter units are KB; the 512 MiB calibration copy of the probe gives the factor between counter and real bytes
(gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane reads)."""
import csv
import json
import sys


def means(path, counter):
    acc = {"agg_fwd": [], "agg_bwd": [], "copy": []}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name, v = r["Kernel_Name"], float(r["Counter_Value"])
        if "k_aggregate<4, false" in name:
            acc["agg_fwd"].append(v)
        elif "k_aggregate<4, true" in name:
            acc["agg_bwd"].append(v)
        elif any(abs(v - t) < 0.02 * t for t in ((524288.0,) if counter == "WRITE_SIZE" else (262144.0, 524288.0))) \
                and "normal" not in name and "distribution" not in name:
            acc["copy"].append(v)                # the 512 MiB calibration copy (counter = all or half of its KB)
    return {k: (sum(v) / len(v) if v else None, len(v)) for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1], sys.argv[2], sys.argv[3]
    workload = sys.argv[4] if len(sys.argv) > 4 else "protein256"
    f, w = means(fetch, "FETCH_SIZE"), means(write, "WRITE_SIZE")
    copy_bytes = 512 * 1024 * 1024
    ff = copy_bytes / (f["copy"][0] * 1024) if f["copy"][0] else 2.0
    wf = copy_bytes / (w["copy"][0] * 1024) if w["copy"][0] else 1.0
    doc = {"workload": workload, "F": 138,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/agg_probe.py",
           "calibration": {"kernel": "512 MiB float4 device copy (known bytes)", "fetch_factor": ff, "write_factor": wf,
                           "note": "gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane reads; WRITE_SIZE is exact"},
           "raw": {k: {"fetch_KB_mean": f[k][0], "dispatches_fetch": f[k][1], "write_KB_mean": w[k][0],
                       "dispatches_write": w[k][1]} for k in f}}
    for k in ("agg_fwd", "agg_bwd"):
        rd, wr = int(f[k][0] * 1024 * ff), int(w[k][0] * 1024 * wf)
        doc[k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("agg_fwd", "agg_bwd")}), "fetch_factor", round(ff, 4), "write_factor", round(wf, 4))


if __name__ == "__main__":
    main()
