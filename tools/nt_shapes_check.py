#!/usr/bin/env python3
"""f16x3 projection against float64 over a list of small shapes (rows, K1, K2, N), plain and transposed weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd.graph import new_padded  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(3)
shapes = [(1422, 60, 60, 200), (1422, 200, 200, 200), (1422, 200, None, 4), (1422, 2, None, 200), (1422, 200, None, 60),
          (1422, 200, 60, 200), (1422, 4, None, 200), (1422, 200, None, 200), (1422, 60, None, 200), (1422, 200, 200, 60),
          (1422, 100, 100, 100), (1422, 104, 104, 100), (1422, 72, 40, 100), (300, 8, None, 8), (300, 12, 20, 24)]
for m, k1, k2, n in shapes:
    for zero_rows in (False, True):
        xa = new_padded(m, k1, dev); xa.copy_(torch.randn(m, k1, generator=gen, device=dev))
        xb = None
        if k2:
            xb = new_padded(m, k2, dev); xb.copy_(torch.randn(m, k2, generator=gen, device=dev))
        if zero_rows:
            xa[::3] = 0
        w = torch.randn(n, k1 + (k2 or 0), generator=gen, device=dev)
        bias = torch.randn(n, generator=gen, device=dev)
        for transposed in (False, True):
            if transposed:
                wt = w.t().contiguous()           # stored [K, N]; used through strides
                wa, wb = wt[:k1].t(), (wt[k1:].t() if k2 else None)
            else:
                wa, wb = w[:, :k1], (w[:, k1:] if k2 else None)
            img = Fn._image(wa, wb, False)
            y = Fn._gemm_nt(xa, img, xb, bias, n)
            ref = xa.double() @ w[:, :k1].double().t() + bias.double()
            if k2:
                ref = ref + xb.double() @ w[:, k1:].double().t()
            err = float((y.double() - ref).abs().max()) / float(ref.abs().max())
            flag = "" if err < 5e-6 else "   <-------- BAD"
            print(f"{img.kind} M={m} K=({k1},{k2}) N={n} zero_rows={zero_rows} transposed={transposed}: {err:.2e}{flag}", flush=True)
