#!/usr/bin/env python3
"""One train step (MODEL=attention|skipcon, F64=1 adds a float64 run of the oracle) on a protein-like batch of benchmark-like size against the CPU oracle with the same
explicit noise (the committed parity tests use small graphs)."""
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from hgvae_amd import synthetic  # noqa: E402
from hgvae_amd.noise import NoiseTape, use_noise  # noqa: E402
from hgvae_amd.vae import SkipconAttentionVAE, SkipconVAE, train_step_loss  # noqa: E402
from oracle import models as OM  # noqa: E402
from oracle import pyg_ops as O  # noqa: E402
from test_gpu_model import _fresh_state, randomize_bn, skipcon_tape  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    which = os.environ.get("MODEL", "attention")
    b = synthetic.protein_like_batch(int(os.environ.get("GRAPHS", 8)), seed=1000, device=dev)
    x, ei, bt = b.x.cpu(), b.edge_index.cpu(), b.batch.cpu()
    n = x.size(0)
    gen = torch.Generator().manual_seed(10)
    h = 256 if which == "attention" else 138
    torch.manual_seed(10)
    ref = (OM.SkipconAttentionVAE if which == "attention" else OM.SkipconVAE)(105, h, 28, 10)
    randomize_bn(ref, gen)
    ref.train()
    if which == "attention":
        loops = int((ei[0] != ei[1]).sum()) + n
        tape = skipcon_tape(n, h, 28, 10, 0.49, gen, gat_edges=loops)
    else:
        tape = skipcon_tape(n, h, 28, 10, 0.5, gen)
    state = _fresh_state(ref)
    loss_ref = OM.skipcon_loss(ref, x, ei, bt, 1.0, O.NoiseTape(tape))[0]
    loss_ref.backward()
    mod = (SkipconAttentionVAE if which == "attention" else SkipconVAE)(105, h, 28, 10)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, b.x, b.edge_index, b.batch, 1.0)
    loss.backward()
    print(f"{which}: N={n} E={ei.size(1)}  loss HIP {float(loss.detach()):.6f}  oracle {float(loss_ref.detach()):.6f}")
    rp = dict(ref.named_parameters())
    r64 = None
    if os.environ.get("F64"):        # the same restatement in float64: which of the two fp32 results is closer?
        ref64 = type(ref)(105, h, 28, 10).double()
        ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
        ref64.train()
        tape64 = [t.double() if t.is_floating_point() else t for t in tape]
        l64 = OM.skipcon_loss(ref64, x.double(), ei, bt, 1.0, O.NoiseTape(tape64))[0]
        l64.backward()
        r64 = dict(ref64.named_parameters())
        print(f"float64 oracle loss {float(l64.detach()):.6f}")
    worst = worst_o = 0.0
    for name, p in mod.named_parameters():
        if rp[name].grad is None:
            continue
        g = (r64[name].grad if r64 is not None else rp[name].grad).double()
        sc = g.abs().max().item() + 1e-30
        rel = (p.grad.double().cpu() - g).abs().max().item() / sc
        worst = max(worst, rel)
        if r64 is not None:
            worst_o = max(worst_o, (rp[name].grad.double() - g).abs().max().item() / sc)
    print(f"worst relative gradient error: HIP {worst:.2e}" + (f", fp32 CPU oracle {worst_o:.2e} (both against float64)" if r64 is not None else " (against the fp32 oracle)"))


if __name__ == "__main__":
    main()
