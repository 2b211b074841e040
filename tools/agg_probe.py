#!/usr/bin/env python3
"""Runs only the aggregation kernel (and a calibration copy) on a benchmark workload, for rocprofv3.

  rocprofv3 --kernel-trace --stats ... -- python3 tools/agg_probe.py --workload protein256
  rocprofv3 --pmc FETCH_SIZE ...       -- python3 tools/agg_probe.py --workload protein256
  rocprofv3 --pmc WRITE_SIZE ...       -- python3 tools/agg_probe.py --workload protein256

The calibration copy (a float4 clone of a 512 MiB tensor: known bytes read and written) gives the
factor between the FETCH_SIZE / WRITE_SIZE counters and real bytes for 16-byte-per-lane accesses.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="protein256")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--width", type=int, default=0, help="feature width (default: the model's hidden width)")
    args = ap.parse_args()
    from bench import agg_bytes, build_workload
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph, new_padded
    dev = torch.device("cuda:0")
    batch, _, dims = build_workload(args.workload, 0, 1, dev)
    f = args.width or dims[1]
    n, e = batch.x.size(0), batch.edge_index.size(1)
    g = Graph(batch.edge_index, n)
    g.by_src
    x = new_padded(n, f, dev)
    x.copy_(torch.randn(n, f, device=dev))
    big = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)  # 512 MiB
    big.normal_()
    torch.cuda.synchronize()
    for _ in range(3):
        big.clone()
    times_f, times_b = [], []
    for _ in range(args.iters):
        s, t, u = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        s.record()
        out = Fn._Aggregate.apply(x, g, True)
        t.record()
        gx = new_padded(n, f, dev, zero_pad=False)
        Fn._agg_launch_raw(g.by_src, n, out, f, 0, None, None, g.inv_deg, None, gx)
        u.record()
        torch.cuda.synchronize()
        times_f.append(s.elapsed_time(t))
        times_b.append(t.elapsed_time(u))
    times_f.sort(); times_b.sort()
    ab = agg_bytes(n, f, e)
    print(f"workload={args.workload} N={n} E={e} F={f} algorithmic_bytes={ab}")
    print(f"fwd median {times_f[len(times_f)//2]*1e3:.1f} us -> {ab/times_f[len(times_f)//2]/1e6:.0f} GB/s algorithmic")
    print(f"bwd median {times_b[len(times_b)//2]*1e3:.1f} us -> {(ab+4*e)/times_b[len(times_b)//2]/1e6:.0f} GB/s algorithmic")
    print(f"calibration copy: {big.numel()*4} bytes read, {big.numel()*4} bytes written per clone")


if __name__ == "__main__":
    main()
