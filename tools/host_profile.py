#!/usr/bin/env python3
"""cProfile of the host side of eager train steps (where does enqueue time go?)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import synthetic  # noqa: E402
from hgvae_amd.distributed import FlatGradBucket  # noqa: E402
from hgvae_amd.optim import FlatAdam  # noqa: E402
from hgvae_amd.vae import SkipconVAE, train_step_loss  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    flat = os.environ.get("FLAT", "1") == "1"
    batch = synthetic.protein_like_batch(8, seed=1, device=dev)
    model = SkipconVAE(105, 138, 28).to(dev).train()
    if flat:
        opt = FlatAdam(model.parameters(), lr=1e-3)
        bucket = opt.bucket
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
        bucket = FlatGradBucket(model.parameters())

    def step():
        bucket.zero()
        loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
