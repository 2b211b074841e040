#!/usr/bin/env python3
"""Where the device path's rounding error comes from: stage-by-stage distance of one skip-connection forward pass to the float64
CPU restatement, next to the fp32 CPU restatement's distance (same weights, same noise tape)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def main():
    from oracle import models as OM
    from oracle import pyg_ops as O
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconVAE
    from test_gpu_model import _fresh_state, randomize_bn, skipcon_tape
    f_in, h, z, ns = 105, int(sys.argv[1]) if len(sys.argv) > 1 else 30, 28, 10
    n, e = 755, int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    gen = torch.Generator().manual_seed(5)
    ei = torch.randint(0, n, (2, e), generator=gen)
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(7)
    ref32 = OM.SkipconVAE(f_in, h, z, ns)
    randomize_bn(ref32, gen)
    state = _fresh_state(ref32)
    tape = skipcon_tape(n, h, z, ns, 0.5, gen)
    ref64 = OM.SkipconVAE(f_in, h, z, ns).double()
    ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
    ref32.load_state_dict(state)
    ref32.train(); ref64.train()
    tape64 = [t.double() if t.is_floating_point() else t for t in tape]
    mod = SkipconVAE(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(DEV).train()
    with torch.no_grad():
        t32, t64 = O.NoiseTape(tape), O.NoiseTape(tape64)
        o32 = ref32.encode(x, ei, t32)
        o64 = ref64.encode(x.double(), ei, t64)
        z32 = OM.reparameterize(o32[2], o32[3], ns, t32)
        z64 = OM.reparameterize(o64[2], o64[3], ns, t64)
        r32 = ref32.decode(z32, o32[0], o32[1], t32)
        r64 = ref64.decode(z64, o64[0], o64[1], t64)
        with use_noise(NoiseTape(tape)):
            od = mod.encode(x.to(DEV), ei.to(DEV))
            zd = mod.reparameterize(od[2], od[3])
            rd = mod.decode(zd, od[0], od[1])
        names = ["x1 (level 1)", "x2 (level 2)", "mean", "log_std"]
        print(f"h={h} n={n} e={e}: stage, device vs f64, fp32 CPU vs f64")
        for k, nm in enumerate(names):
            print(f"  {nm:14s} {rel(od[k], o64[k]):.2e}   {rel(o32[k], o64[k]):.2e}")
        print(f"  {'z':14s} {rel(zd, z64):.2e}   {rel(z32, z64):.2e}")
        print(f"  {'reconstruction':14s} {rel(rd, r64):.2e}   {rel(r32, r64):.2e}")
        kl64 = OM.kl_divergence(o64[2], o64[3]); kl32 = OM.kl_divergence(o32[2], o32[3]); kld = mod.kl_divergence(od[2], od[3])
        print(f"  {'kl':14s} {abs(float(kld) - float(kl64)) / abs(float(kl64)):.2e}   {abs(float(kl32) - float(kl64)) / abs(float(kl64)):.2e}   (kl = {float(kl64):.6g})")
        # the operators alone on identical inputs: level-1 convolution, then its epilogue
        c64 = ref64.encoder1[0](x.double(), ei)
        c32 = ref32.encoder1[0](x, ei)
        cd = mod.encoder1[0](x.to(DEV), ei.to(DEV))
        print(f"  {'conv1 alone':14s} {rel(cd, c64):.2e}   {rel(c32, c64):.2e}")
        a64 = torch.relu(c64) * tape64[0] * 2
        bn64 = torch.nn.functional.batch_norm(a64, None, None, ref64.encoder1[2].weight, ref64.encoder1[2].bias, True, 0.1, 1e-5)
        a32 = torch.relu(c64.float()) * tape[0] * 2
        bn32 = torch.nn.functional.batch_norm(a32, None, None, ref32.encoder1[2].weight, ref32.encoder1[2].bias, True, 0.1, 1e-5)
        with use_noise(NoiseTape([tape[0]])):
            bnd = Fn.bn_act(c64.float().to(DEV), mod.encoder1[2], True, 0.5, True)
        print(f"  {'epilogue alone':14s} {rel(bnd, bn64):.2e}   {rel(bn32, bn64):.2e}   (input: the float64 convolution rounded to fp32)")


if __name__ == "__main__":
    main()
