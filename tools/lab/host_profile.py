#!/usr/bin/env python3
"""cProfile of the host side of one workload's eager train steps (where the Python / ctypes / autograd time goes).
  python3 tools/lab/host_profile.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "chig32"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    import bench
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    from hgvae_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    batch, kind, dims = bench.build_workload(workload, 0, 1, dev)
    torch.manual_seed(0)
    hgvae_amd.manual_seed(1234)
    model, loss_fn = bench.make_model(kind, dims)
    model = model.to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-3)
    bucket = opt.bucket
    graph_of(batch.edge_index, batch.x.size(0)).check()

    def step():
        bucket.zero()
        loss = loss_fn(model, batch.x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():
            loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{workload}: host enqueue {enq / steps * 1e3:.3f} ms/step, wall {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
    torch.autograd.set_multithreading_enabled(False)      # backward on this thread, so that the profiler sees it
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main()
