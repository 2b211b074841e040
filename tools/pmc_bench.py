#!/usr/bin/env python3
"""Counter summary of one bench.py workload -> gpurun_out/pmc/pmc_<workload>_b<graphs>.json; bench.py reads the copy
kept under profiles/.

  python3 tools/pmc_bench.py --workload protein256 [--agg-kernel 'k_aggregate<4, false, 3, 2>' --agg-width 138]

Three rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE), each at two step
counts; per-step figures are the difference of the two runs divided by the difference in steps, so warm-up, the
workload generator and the per-launch timing pass cancel.  This script only starts rocprofv3 as child processes
(the program after "--" is python3 itself) and reads their CSV files; it never touches the GPU.

Units: FETCH_SIZE / WRITE_SIZE count KB; on gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane reads, so
read bytes = FETCH_SIZE x 1024 x 2 (an upper estimate for kernels with narrower reads); WRITE_SIZE is exact.
MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counters, workload, steps, out, extra):
    d = os.path.join(out, f"{counters[0]}_{steps}")
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["timeout", "-s", "KILL", "240", "rocprofv3", "--pmc", *counters, "-d", d, "-o", "p", "--output-format", "csv",
           "--", "python3", os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", "2",
           "--no-cpu-baseline", "--headline-only"] + extra
    log = open(d + ".log", "w")
    rc = subprocess.run(cmd, stdout=log, stderr=subprocess.STDOUT, cwd=ROOT).returncode
    if rc != 0:
        sys.exit(f"pass {counters} steps {steps} failed with {rc}: see {d}.log")
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    line = [ln for ln in open(d + ".log") if ln.startswith("{\"metric\"")]
    shutil.rmtree(d, ignore_errors=True)
    return acc, (json.loads(line[-1]) if line else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="protein256")
    ap.add_argument("--agg-kernel", default="k_aggregate<4, false, 3, 2>")
    ap.add_argument("--agg-width", type=int, default=138)
    ap.add_argument("--steps", type=int, nargs=2, default=[3, 9])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    args, extra = ap.parse_known_args()
    os.makedirs(args.out, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    lo, hi = args.steps
    per_step = defaultdict(dict)      # kernel -> counter -> value per step
    per_disp = defaultdict(dict)      # kernel -> counter -> mean per dispatch (larger run)
    disp = {}
    line = None
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]):
        a_lo, _ = run_pass(counters, args.workload, lo, args.out, extra)
        a_hi, line = run_pass(counters, args.workload, hi, args.out, extra)
        for k, cs in a_hi.items():
            for c, (v, n) in cs.items():
                v0, n0 = a_lo.get(k, {}).get(c, (0.0, 0))
                per_step[k][c] = (v - v0) / (hi - lo)
                per_disp[k][c] = v / n if n else 0.0
                disp[k] = (n - n0) / (hi - lo)
        print(f"[pmc] {counters} done", flush=True)

    def rd(k):
        return per_step[k].get("FETCH_SIZE", 0.0) * 1024 * 2

    def wr(k):
        return per_step[k].get("WRITE_SIZE", 0.0) * 1024

    def busy(ks):
        b = sum(per_step[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks)
        g = sum(per_step[k].get("GRBM_GUI_ACTIVE", 0.0) for k in ks)
        return b / (1024.0 * g / 8.0) if g > 0 else None

    in_step = [k for k in per_step if disp.get(k, 0) > 0]
    graphs = line["config"]["graphs_per_gpu"] if line else None
    sys.path.insert(0, ROOT)
    import bench
    doc = {"workload": args.workload, "graphs_per_gpu": graphs, "source_hash": bench.source_hash(),
           "git_commit": os.environ.get("HGVAE_GIT_COMMIT"),
           "nodes": line["config"]["nodes_per_gpu"] if line else None, "edges": line["config"]["edges_per_gpu"] if line else None,
           "source": f"tools/pmc_bench.py: rocprofv3 --pmc <counter> -- python3 bench.py --workload {args.workload} "
                     f"--steps {{{lo},{hi}}} --warmup 2 (separate passes; per-step = difference / {hi - lo})",
           "units": "bytes; reads = FETCH_SIZE KB x 1024 x 2 (gfx950 counts 16-byte-per-lane reads at half), writes = WRITE_SIZE KB x 1024"}
    agg = [k for k in per_disp if args.agg_kernel in k]
    if agg:
        k = agg[0]
        r = per_disp[k].get("FETCH_SIZE", 0.0) * 1024 * 2
        w = per_disp[k].get("WRITE_SIZE", 0.0) * 1024
        doc["aggregation"] = {"kernel": k, "F": args.agg_width, "hbm_read_bytes": int(r), "hbm_write_bytes": int(w),
                              "hbm_bytes_per_launch": int(r + w), "dispatches_per_step": disp.get(k)}
    proj = {}
    for tag, pref in (("nt", ("k_gemm_nt",)), ("tn", ("k_gemm_tn", "k_reduce_tn"))):
        ks = [k for k in in_step if any(p in k for p in pref)]
        if ks:
            proj[tag] = {"kernels": sorted(ks), "mfma_busy_frac": round(busy([k for k in ks if "k_gemm" in k]) or 0.0, 4),
                         "hbm_bytes_per_step": int(sum(rd(k) + wr(k) for k in ks)),
                         "pmc_source": "profiles/" + os.path.basename(out_name(args.workload, graphs))}
    doc["projections"] = proj
    tr, tw = sum(rd(k) for k in in_step), sum(wr(k) for k in in_step)
    doc["step"] = {"hbm_read_bytes": int(tr), "hbm_write_bytes": int(tw), "hbm_bytes": int(tr + tw)}
    rows = sorted(in_step, key=lambda k: -(rd(k) + wr(k)))[:24]
    doc["kernels"] = [{"kernel": k[:96], "dispatches_per_step": round(disp[k], 2), "read_bytes_per_step": int(rd(k)),
                       "write_bytes_per_step": int(wr(k)), "mfma_busy_frac": (round(busy([k]), 4) if busy([k]) is not None else None)}
                      for k in rows]
    # written under gpurun_out/ (the directory that comes back from the GPU box); copy it to profiles/ to publish it
    path = os.path.join(args.out, os.path.basename(out_name(args.workload, graphs)))
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps({"written": path, "step": doc["step"], "aggregation": doc.get("aggregation"),
                      "projections": {t: {"mfma_busy_frac": p["mfma_busy_frac"], "hbm_bytes_per_step": p["hbm_bytes_per_step"]}
                                      for t, p in proj.items()}}))


def out_name(workload, graphs):
    return os.path.join("profiles", f"pmc_{workload}_b{graphs}.json")


if __name__ == "__main__":
    main()
