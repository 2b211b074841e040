#!/usr/bin/env python3
"""Host cost of an eager chignolin-sized step: wall per step, and cProfile (tottime / cumtime tops)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn, synthetic  # noqa: E402
from hgvae_amd.optim import FlatAdam  # noqa: E402
from hgvae_amd.vae import SkipconVAE, train_step_loss  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 32, 60, seed=1, device=dev)
    model = SkipconVAE(60, 138, 28).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-3)
    bucket = opt.bucket

    def step():
        bucket.zero()
        loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():
            loss.backward()
        opt.step()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host enqueue {1e3 * (t1 - t0) / 100:.3f} ms/step, with sync {1e3 * (time.perf_counter() - t0) / 100:.3f}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)
    st.sort_stats("cumtime").print_stats(30)


if __name__ == "__main__":
    main()
