#!/usr/bin/env python3
"""Accuracy and time of the scaled two-piece fp16 projection kernel (f16x3) next to the bf16 six-product kernel.
  python3 tools/f16x3_check.py
Errors are max |C - C64| over the largest sum of |terms| of the result (float64 on the device as the reference)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn  # noqa: E402

DEV = torch.device("cuda:0")


def product(kind, xa, wa, xb, wb, bias):
    Fn.set_projection(kind)
    with torch.no_grad():
        return Fn.dual_linear(xa, wa, xb, wb, bias)


def err(c, xa, wa, xb, wb, bias):
    want = xa.double() @ wa.double().t()
    terms = xa.double().abs() @ wa.double().abs().t()
    if xb is not None:
        want = want + xb.double() @ wb.double().t()
        terms = terms + xb.double().abs() @ wb.double().abs().t()
    if bias is not None:
        want = want + bias.double()
        terms = terms + bias.double().abs()
    per_row = ((c.double() - want).abs() / terms.clamp_min(1e-300)).max()
    return float((c.double() - want).abs().max() / terms.max()), float(per_row)


def case(name, m, k1, k2, n, make):
    g = torch.Generator(device=DEV).manual_seed(1)
    xa = make(m, k1, g)
    xb = make(m, k2, g) if k2 else None
    wa = torch.randn(n, k1, device=DEV, generator=g) * 0.1
    wb = torch.randn(n, k2, device=DEV, generator=g) * 0.1 if k2 else None
    bias = torch.randn(n, device=DEV, generator=g) * 0.01
    out = {}
    for kind in ("bf16x6", "f16x3"):
        c = product(kind, xa, wa, xb, wb, bias)
        out[kind] = err(c, xa, wa, xb, wb, bias)
    print(f"{name:34s} m={m:7d} k={k1}+{k2} n={n}: " + "  ".join(f"{k} {v[0]:.2e} (per element {v[1]:.2e})" for k, v in out.items()), flush=True)


def timing(m, k1, k2, n, reps=30):
    g = torch.Generator(device=DEV).manual_seed(2)
    xa = Fn.as_padded(torch.randn(m, k1, device=DEV, generator=g))
    xb = Fn.as_padded(torch.randn(m, k2, device=DEV, generator=g)) if k2 else None
    wa = torch.randn(n, k1, device=DEV, generator=g) * 0.1
    wb = torch.randn(n, k2, device=DEV, generator=g) * 0.1 if k2 else None
    res = {}
    for kind in ("bf16x6", "f16x3"):
        Fn.set_projection(kind)
        img = Fn._image(wa, wb, False)
        for _ in range(5):
            Fn._gemm_nt(xa, img, xb, None, n)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            Fn._gemm_nt(xa, img, xb, None, n)
        e1.record()
        torch.cuda.synchronize()
        res[kind] = e0.elapsed_time(e1) / reps * 1e3
    print(f"time m={m} k={k1}+{k2} n={n}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)


def tn_case(name, m, n, ka, kb, make_g, make_a, reps=20):
    """Weight gradients G^T [Xa | Xb] and column sums of G: error over the largest sum of |terms| / per element, time."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    g = Fn.as_padded(make_g(m, n, gen))
    xa = Fn.as_padded(make_a(m, ka, gen))
    xb = Fn.as_padded(make_a(m, kb, gen)) if kb else None
    x_all = torch.cat([xa[:, :ka], xb[:, :kb]], 1) if kb else xa[:, :ka]
    want = g[:, :n].double().t() @ x_all.double()
    terms = g[:, :n].double().abs().t() @ x_all.double().abs()
    wb = g[:, :n].double().sum(0)
    tb = g[:, :n].double().abs().sum(0)
    out = {}
    for kind in ("bf16x6", "f16x3"):
        Fn.set_projection(kind)
        Fn._tn_f16 = kind == "f16x3"          # the fp16 weight-gradient kernel is opt-in (HGVAE_TN_F16=1)
        run = (lambda: Fn._gemm_tn2(g, xa, xb, True)) if kb else (lambda: Fn._gemm_tn(g, xa, True))
        res = run()
        gw = torch.cat([res[0], res[1]], 1) if kb else res[0]
        gb = res[-1]
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        d = (gw.double() - want).abs()
        out[kind] = (float(d.max() / terms.max()), float((d / terms.clamp_min(1e-300)).max()),
                     float(((gb.double() - wb).abs() / tb.clamp_min(1e-300)).max()), e0.elapsed_time(e1) / reps * 1e3)
    print(f"tn {name:30s} m={m} n={n} k={ka}+{kb}: " + "  ".join(
        f"{k} {v[0]:.2e} (per element {v[1]:.2e}, bias {v[2]:.2e}) {v[3]:.1f} us" for k, v in out.items()), flush=True)


def tn_main():
    randn = lambda m, k, g: torch.randn(m, k, device=DEV, generator=g)
    tn_case("normal", 160984, 138, 138, 138, randn, randn)
    tn_case("normal single", 160984, 138, 138, 0, randn, randn)
    tn_case("gradient-sized G", 160984, 138, 138, 138, lambda m, k, g: randn(m, k, g) * 1e-9, randn)
    tn_case("ragged rows", 40001, 138, 105, 0, randn, randn)

    def col_scaled(m, k, g):     # columns of very different magnitude
        return randn(m, k, g) * (10.0 ** torch.randint(-6, 1, (1, k), device=DEV, generator=g).float())
    tn_case("columns 1e-6 .. 1", 160984, 138, 138, 138, col_scaled, randn)

    def growing(m, k, g):        # later rows far larger: the running scale is lowered again and again
        return randn(m, k, g) * (10.0 ** (torch.arange(m, device=DEV).float() / m * 12 - 6)).unsqueeze(1)
    tn_case("rows growing 1e-6 .. 1e6", 160984, 138, 138, 138, growing, growing)
    tn_case("rows falling", 160984, 138, 138, 138, lambda m, k, g: growing(m, k, g).flip(0), randn)

    def blocky(m, k, g):         # zeros first, then data
        x = randn(m, k, g)
        x[: m // 3] = 0
        return x
    tn_case("zero rows first", 160984, 138, 138, 138, blocky, blocky)
    tn_case("batch 256", 1285158, 138, 138, 138, randn, randn, reps=5)


def main():
    randn = lambda m, k, g: torch.randn(m, k, device=DEV, generator=g)
    if "--tn" in sys.argv:
        return tn_main()
    if "--time-only" in sys.argv:
        for shape in ((160984, 138, 138, 138), (160984, 138, 0, 138), (160984, 105, 0, 138), (1285158, 138, 138, 138)):
            timing(*shape)
        return
    case("normal", 4096, 138, 138, 138, randn)
    case("normal single", 4099, 138, 0, 138, randn)
    case("wide (streamed image)", 3000, 256, 256, 256, randn)
    case("narrow", 1000, 5, 0, 3, randn)
    case("gradient-sized (1e-8)", 4096, 138, 138, 138, lambda m, k, g: randn(m, k, g) * 1e-8)
    case("huge (1e30)", 4096, 138, 0, 138, lambda m, k, g: randn(m, k, g) * 1e30)

    def rising(m, k, g):          # every 32-wide chunk 300x larger than the one before: the row scale is lowered repeatedly
        x = randn(m, k, g)
        return x * (300.0 ** torch.arange(k, device=DEV).div(32, rounding_mode="floor")).float()
    case("rising chunks", 2048, 138, 138, 138, rising)

    def falling(m, k, g):
        x = randn(m, k, g)
        return x * (300.0 ** (-torch.arange(k, device=DEV).div(32, rounding_mode="floor"))).float()
    case("falling chunks", 2048, 138, 138, 138, falling)

    def sparse_rows(m, k, g):     # one-hot-like rows, rows of zeros, rows of different magnitude
        x = (torch.rand(m, k, device=DEV, generator=g) < 0.02).float()
        x[::7] = 0
        x = x * (10.0 ** torch.randint(-20, 20, (m, 1), device=DEV, generator=g).float())
        return x
    case("one-hot rows of mixed magnitude", 4096, 105, 0, 138, sparse_rows)

    def row_scaled(m, k, g):
        return randn(m, k, g) * (10.0 ** torch.randint(-30, 30, (m, 1), device=DEV, generator=g).float())
    case("rows of mixed magnitude", 4096, 138, 138, 138, row_scaled)
    for shape in ((160984, 138, 138, 138), (160984, 138, 0, 138), (160984, 105, 105, 138), (160984, 138, 138, 56), (160984, 256, 256, 256),
                  (4416, 138, 138, 138), (1285158, 138, 138, 138)):
        timing(*shape)


if __name__ == "__main__":
    main()
