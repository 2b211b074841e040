#!/usr/bin/env python3
"""Soak: a few hundred train steps per model variant (eager, then as a captured step where supported): loss finite and falling,
device memory flat.  python3 tools/lab/soak.py [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    import bench
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd.graphed import GraphedTrainStep
    from hgvae_amd.optim import FlatAdam
    dev = torch.device("cuda:0")
    for workload in ("protein32", "attention", "pool", "custom_pool", "chig32"):
        for graph in (False, True):
            batch, kind, dims = bench.build_workload(workload, 0, 1, dev)
            torch.manual_seed(0)
            hgvae_amd.manual_seed(1)
            model, loss_fn = bench.make_model(kind, dims)
            model = model.to(dev).train()
            opt = FlatAdam(model.parameters(), lr=1e-3, capturable=graph)
            bucket = opt.bucket

            def step():
                bucket.zero()
                loss = loss_fn(model, batch.x, batch.edge_index, batch.batch, 1.0)
                with Fn.direct_weight_grads():
                    loss.backward()
                opt.step()
                return loss

            run = step
            if graph:
                run = GraphedTrainStep(model, opt, batch.x, batch.edge_index, batch.batch, 1.0, loss_fn=loss_fn, bucket=bucket, seed=1, warmup=2)
            losses, mem = [], []
            n = steps if workload != "protein32" else max(steps // 4, 20)
            for i in range(n):
                l = run()
                if i % max(n // 10, 1) == 0 or i == n - 1:
                    losses.append(float(l))
                    mem.append(torch.cuda.memory_allocated() >> 20)
            ok = all(x == x and abs(x) < 1e30 for x in losses) and losses[-1] < losses[0] and max(mem[2:]) - min(mem[2:]) <= 64
            print(f"{workload:12s} {'graph' if graph else 'eager'} steps={n} loss {losses[0]:.4f} -> {losses[-1]:.4f}  mem MiB {mem[2]}..{mem[-1]}  {'ok' if ok else 'CHECK'}", flush=True)
            del model, opt, run
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
