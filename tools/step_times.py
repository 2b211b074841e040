#!/usr/bin/env python3
"""Per-step wall times (ms) of eager train steps, FLAT=1 (FlatAdam) or FLAT=0 (torch fused Adam)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import synthetic  # noqa: E402
from hgvae_amd.distributed import FlatGradBucket  # noqa: E402
from hgvae_amd.optim import FlatAdam  # noqa: E402
from hgvae_amd.vae import SkipconVAE, train_step_loss  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    flat = os.environ.get("FLAT", "1") == "1"
    batch = synthetic.protein_like_batch(32, seed=1000, device=dev)
    torch.manual_seed(0)
    model = SkipconVAE(105, 138, 28).to(dev).train()
    if flat:
        opt = FlatAdam(model.parameters(), lr=1e-3)
        bucket = opt.bucket
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
        bucket = FlatGradBucket(model.parameters())
    import gc
    if os.environ.get("NOGC") == "1":
        gc.disable()
    gc.callbacks.append(lambda phase, info: phase == "stop" and info["generation"] == 2 and print("gen2 gc", info, flush=True))
    ts, ls = [], []
    w0 = model.encoder2[0].lin_l.weight
    for i in range(int(os.environ.get("STEPS", "30"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bucket.zero()
        loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        ls.append(float(loss))
    print("flat" if flat else "torch", " ".join(f"{t:.1f}" for t in ts))
    print("losses", " ".join(f"{v:.2f}" for v in ls), "weight version", w0._version)


if __name__ == "__main__":
    main()
