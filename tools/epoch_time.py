#!/usr/bin/env python3
"""Epoch time of the reference's own training shape - frames of one 138-node molecule, 32 graphs per batch
(gvae/gvae_skipcon.py:262-263) - through hgvae_amd.vae.train_vae (eager launches) and through
hgvae_amd.graphed.CapturedTraining (one captured step, features copied in per batch), both fed by a device-resident
GraphShard loader.   python3 tools/epoch_time.py [frames] [epochs]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hgvae_amd  # noqa: E402
from hgvae_amd.data import Data  # noqa: E402
from hgvae_amd.graphed import CapturedTraining  # noqa: E402
from hgvae_amd.optim import FlatAdam  # noqa: E402
from hgvae_amd.shard import GraphShard  # noqa: E402
from hgvae_amd.vae import SkipconVAE, train_vae  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    a = torch.arange(137)
    ei = torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], 1)
    base = torch.randn(138, 60, generator=gen)
    data = [Data(x=base + 0.1 * torch.randn(138, 60, generator=gen), edge_index=ei) for _ in range(frames)]
    shard = GraphShard.from_data_list(data, dev)
    loader = shard.loader(32, shuffle=True, seed=1)
    for name in ("eager train_vae", "CapturedTraining"):
        torch.manual_seed(0)
        hgvae_amd.manual_seed(1)
        model = SkipconVAE(60, 138, 28).to(dev).train()
        opt = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
        cap = CapturedTraining(model, opt, seed=1) if name == "CapturedTraining" else None
        # the captured step carries its own CSR: its loader only has to gather features (and the edge list it is recognised by)
        lean = shard.loader(32, shuffle=True, seed=1, with_csr=False)
        run = (lambda: cap.train_epoch(lean)) if cap else (lambda: train_vae(model, loader, opt))
        first = run()          # warm-up epoch (captures, workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [run() for _ in range(epochs)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / epochs
        print(f"{name:18s}: {len(loader)} batches of 32 graphs per epoch, {dt * 1e3:8.1f} ms per epoch = {dt / len(loader) * 1e3:.3f} ms per step, "
              f"{frames / dt:9.0f} graphs/s; loss {first:.4f} -> {losses[-1]:.4f}"
              + (f"; captures {len(cap._entries)}, captured steps {cap.captured_steps}, eager steps {cap.eager_steps}" if cap else ""), flush=True)


def validation():
    """The same data through validate_vae (eager) and CapturedTraining.validate_epoch."""
    from hgvae_amd.vae import validate_vae
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    a = torch.arange(137)
    ei = torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], 1)
    base = torch.randn(138, 60, generator=gen)
    shard = GraphShard.from_data_list([Data(x=base + 0.1 * torch.randn(138, 60, generator=gen), edge_index=ei) for _ in range(frames)], dev)
    loader = shard.loader(32, with_csr=False)
    torch.manual_seed(0)
    model = SkipconVAE(60, 138, 28).to(dev)
    cap = CapturedTraining(model, FlatAdam(model.parameters(), lr=1e-3, capturable=True), seed=1)
    for name, run in (("eager validate_vae", lambda: validate_vae(model, loader)), ("captured validate_epoch", lambda: cap.validate_epoch(loader))):
        v = run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v = run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name:24s}: {len(loader)} batches of 32 graphs, {dt * 1e3:7.1f} ms per epoch = {dt / len(loader) * 1e3:.3f} ms per batch; loss {v:.4f}", flush=True)


if __name__ == "__main__":
    main()
    validation()
