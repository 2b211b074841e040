#!/usr/bin/env python3
"""CSR build (K1) of the headline batch, both directions, several times: run under `rocprofv3 --kernel-trace --stats` for
per-kernel times (tools/k1_time.py [workload])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hgvae_amd.graph import Graph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    batch, _, _ = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "protein256", 0, 1, dev)
    ei, n = batch.edge_index, batch.x.size(0)
    for _ in range(2):
        Graph(ei, n).by_src
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        Graph(ei, n).by_src
    torch.cuda.synchronize()
    print(f"N={n} E={ei.size(1)}  both CSRs: {(time.perf_counter() - t0) / reps * 1e6:.1f} us")


if __name__ == "__main__":
    main()
