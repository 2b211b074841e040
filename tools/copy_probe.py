#!/usr/bin/env python3
"""Lists the aten::copy_ / cat / add_ calls (with input shapes) of one eager train step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import synthetic, vae  # noqa: E402
from hgvae_amd.distributed import FlatGradBucket  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    batch = synthetic.protein_like_batch(2, seed=1, device=dev, nodes=(500, 600))
    model = vae.SkipconVAE(105, 138, 28).to(dev)
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd import functional as Fn
    opt = FlatAdam(model.parameters(), lr=1e-3)
    bucket = opt.bucket
    x = vae.padded_input(batch.x)

    def step():
        bucket.zero()
        loss = vae.train_step_loss(model, x, batch.edge_index, batch.batch, 1.0)
        with Fn.direct_weight_grads():
            loss.backward()
        bucket.all_reduce_mean()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    import collections
    cnt = collections.Counter()
    for ev in prof.events():
        if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::as_strided", "aten::view", "aten::slice", "aten::empty_strided", "aten::select", "aten::detach", "aten::alias", "aten::expand", "aten::t", "aten::transpose", "aten::unsqueeze", "aten::_unsafe_view", "aten::reshape", "aten::narrow", "aten::empty_like", "aten::result_type", "aten::to", "aten::lift_fresh", "aten::item", "aten::_local_scalar_dense", "aten::chunk", "aten::split"):
            stack = [s for s in (ev.stack or []) if "hgvae_amd" in s or "tools/" in s]
            cnt[(ev.name, str(ev.input_shapes)[:60], stack[0][-60:] if stack else "")] += 1
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
        print(v, k)


if __name__ == "__main__":
    main()
