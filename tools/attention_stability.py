#!/usr/bin/env python3
"""Loss after 13 Adam steps (lr 1e-3) of the attention variant on a small protein-like batch for several noise seeds:
CPU oracle (torch RNG) next to the HIP implementation (Philox), to tell training dynamics from implementation faults."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgvae_amd  # noqa: E402
from hgvae_amd import synthetic  # noqa: E402
from hgvae_amd.vae import SkipconAttentionVAE, train_step_loss  # noqa: E402
from oracle import models as OM  # noqa: E402
from oracle import pyg_ops as O  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    b = synthetic.protein_like_batch(int(os.environ.get("GRAPHS", 2)), seed=1000, device=dev, nodes=(900, 1100))
    x, ei, bt = b.x.cpu(), b.edge_index.cpu(), b.batch.cpu()
    for seed in range(5):
        torch.manual_seed(0)
        ref = OM.SkipconAttentionVAE(105, 256, 28)
        state = {k: v.clone() for k, v in ref.state_dict().items()}
        opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
        torch.manual_seed(100 + seed)
        for _ in range(13):
            opt.zero_grad()
            loss_r = OM.skipcon_loss(ref, x, ei, bt, 1.0, None)[0]
            loss_r.backward()
            opt.step()
        mod = SkipconAttentionVAE(105, 256, 28)
        mod.load_state_dict(state)
        mod = mod.to(dev).train()
        hgvae_amd.manual_seed(seed)
        opt = torch.optim.Adam(mod.parameters(), lr=1e-3)
        for _ in range(13):
            opt.zero_grad()
            loss = train_step_loss(mod, b.x, b.edge_index, b.batch, 1.0)
            loss.backward()
            opt.step()
        print(f"seed {seed}: oracle (CPU) loss {float(loss_r):10.3f}   HIP loss {float(loss):10.3f}", flush=True)


if __name__ == "__main__":
    main()
