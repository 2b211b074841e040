#!/usr/bin/env python3
"""Train-step time of the four model variants on protein-like graphs (eager), with a per-kernel summary."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgvae_amd  # noqa: E402
from hgvae_amd import synthetic  # noqa: E402
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd.optim import FlatAdam  # noqa: E402
from hgvae_amd.vae import (CustomPoolVAE, PoolVAE, SkipconAttentionVAE, SkipconVAE, custom_pool_step_loss,  # noqa: E402
                           train_step_loss)


def main():
    dev = torch.device("cuda:0")
    ngraphs = int(os.environ.get("GRAPHS", "8"))
    which = os.environ.get("MODELS", "skipcon,attention,pool,custom_pool").split(",")
    b105 = synthetic.protein_like_batch(ngraphs, seed=1000, device=dev)
    import copy
    b6 = copy.copy(b105)
    b6.x = torch.randn(b105.x.size(0), 6, device=dev)
    cases = {
        "skipcon": (lambda: SkipconVAE(105, 138, 28), b105, train_step_loss),
        "attention": (lambda: SkipconAttentionVAE(105, 256, 28), b105, train_step_loss),
        "pool": (lambda: PoolVAE(6, 138, 28), b6, train_step_loss),
        "custom_pool": (lambda: CustomPoolVAE(6, 138, 28), b6, None),
    }
    for name in which:
        make, batch, lossfn = cases[name]
        torch.manual_seed(0)
        hgvae_amd.manual_seed(int(os.environ.get("NOISE_SEED", "0")))     # same dropout / eps stream whatever ran before
        model = make().to(dev).train()
        opt = FlatAdam(model.parameters(), lr=1e-3)

        def step():
            opt.zero_grad()
            if name == "custom_pool":
                loss = custom_pool_step_loss(model, batch.x, batch.edge_index, batch.batch)
            else:
                loss = lossfn(model, batch.x, batch.edge_index, batch.batch, 1.0)
            loss.backward()
            opt.step()
            return loss

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.freeze()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:12s} nodes={batch.x.size(0)} edges={batch.edge_index.size(1)}  {dt * 1e3:8.2f} ms/step  "
              f"{ngraphs / dt:9.1f} graphs/s  loss={float(loss):.4f}", flush=True)


if __name__ == "__main__":
    main()
