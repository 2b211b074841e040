#!/usr/bin/env python3
"""Mini-batch assembly: GraphShard.batch (two gather launches, CSR included) against the host path
(Batch.from_data_list on CPU tensors, upload, CSR build by radix sort) on protein-like graphs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import Batch, Data, Graph, GraphShard, synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    ngraphs, bsz = int(os.environ.get("GRAPHS", 64)), int(os.environ.get("BATCH", 32))
    gen = torch.Generator(device=dev).manual_seed(7)
    sizes = torch.randint(4500, 5501, (ngraphs,), generator=gen, device=dev).tolist()
    graphs = [synthetic.protein_like_graph(n, 12, 105, gen, dev) for n in sizes]
    host = [Data(x=g.x.cpu(), edge_index=g.edge_index.cpu()) for g in graphs]
    shard = GraphShard.from_data_list(host, dev)
    ids = list(range(0, 2 * bsz, 2))[:bsz]
    for _ in range(3):
        b = shard.batch(ids)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record(); b = shard.batch(ids); e.record()
        torch.cuda.synchronize()
        ts.append((s.elapsed_time(e) * 1e3, (time.perf_counter() - t0) * 1e6))
    ts.sort()
    gpu_us, wall_us = ts[len(ts) // 2]
    n, e_ = b.x.size(0), b.edge_index.size(1)
    ld = b.x.stride(0)
    moved = 2 * 4 * n * ld + 8 * n + 2 * 2 * 4 * (n + 1) + e_ * (2 * 16 + 2 * 2 * 8)   # read + write of every array
    print(f"shard.batch: {bsz} graphs, N={n} E={e_}: {gpu_us:.1f} us on the GPU ({moved / gpu_us / 1e3:.0f} GB/s moved), "
          f"{wall_us:.0f} us wall incl. host index arithmetic")
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        hb = Batch.from_data_list([host[i] for i in ids])
        x, ei = hb.x.to(dev), hb.edge_index.to(dev)
        g = Graph(ei, x.size(0)); g.by_src
        torch.cuda.synchronize()
    host_us = (time.perf_counter() - t0) / reps * 1e6
    print(f"host collate + upload + CSR sort: {host_us:.0f} us wall  ({host_us / wall_us:.0f}x)")
    assert torch.equal(b.edge_index, ei) and torch.equal(b.graph.by_dst.col, g.by_dst.col)


if __name__ == "__main__":
    main()
