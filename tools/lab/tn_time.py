#!/usr/bin/env python3
"""Time of the bf16 weight-gradient product (hgv_gemm_tn16 + reduction) on the benchmark shape; run with HGVAE_LIB pointing
at an ablation build to see what the launch spends its time on."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hgvae_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda:0")
Fn.set_projection("bf16x6")
for m in (80_492, 160_984, 321_968):
    gen = torch.Generator(device=dev).manual_seed(1)
    g = Fn.as_padded(torch.randn(m, 138, device=dev, generator=gen))
    xa = Fn.as_padded(torch.randn(m, 138, device=dev, generator=gen))
    xb = Fn.as_padded(torch.randn(m, 138, device=dev, generator=gen))
    for _ in range(5):
        Fn._gemm_tn2(g, xa, xb, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        Fn._gemm_tn2(g, xa, xb, True)
    e1.record()
    torch.cuda.synchronize()
    print(f"m={m}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", end="  ")
print()
