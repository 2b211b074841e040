#!/usr/bin/env python3
"""GATConv forward + backward on the graph of the `large` workload (100 k nodes, 2 M edges, two nodes with ~2 100
out-edges): hub rows by a workgroup each against every row by its lane group.  Prints ms per forward + backward."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd import nn as HN  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    batch, _, _ = bench.build_workload("large", 0, 1, dev)
    c = 128
    torch.manual_seed(0)
    conv = HN.GATConv(c, c, heads=1, dropout=0.49).to(dev).train()
    x = batch.x[:, :c].contiguous().requires_grad_(True)
    g = torch.randn(x.size(0), c, device=dev)
    for long_rows in (False, True, False, True):
        Fn._gat_long = long_rows
        for _ in range(3):
            conv(x, batch.edge_index).backward(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            conv(x, batch.edge_index).backward(g)
        torch.cuda.synchronize()
        print(f"hub rows by a workgroup each: {long_rows}   {(time.perf_counter() - t0) * 50:.3f} ms per forward + backward", flush=True)


if __name__ == "__main__":
    main()
