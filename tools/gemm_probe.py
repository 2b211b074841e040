#!/usr/bin/env python3
"""Times the projection kernels (HGVAE_PROJECTION = bf16x6 | fp32) and the weight-gradient kernel on the
benchmark shapes (events on the launching stream; weight images are built outside the timed calls)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd.graph import new_padded  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def nt_call(xa, img, xb, bias, n):
    return Fn._gemm_nt(xa, img, xb, bias, n)


def main():
    dev = torch.device("cuda:0")
    m = int(os.environ.get("M", 160984))
    shapes = [(138, 138, 138), (138, None, 138), (105, 105, 138), (138, 138, 56), (276, None, 138), (28, None, 138)]
    if os.environ.get("WIDE"):     # hidden width 256 (attention variant)
        shapes = [(256, 256, 256), (256, None, 256), (105, 105, 256), (256, None, 56)]
    if os.environ.get("SHAPE"):        # "k1,k2,n" (k2 = 0: one source)
        k1, k2, n = (int(v) for v in os.environ["SHAPE"].split(","))
        shapes = [(k1, k2 or None, n)]
    only = os.environ.get("ONLY")
    if only is not None and not os.environ.get("SHAPE"):
        shapes = [shapes[int(only)]]
    for k1, k2, n in shapes:
        xa = new_padded(m, k1, dev); xa.normal_()
        xb = None
        if k2:
            xb = new_padded(m, k2, dev); xb.normal_()
        w = torch.randn(n, k1 + (k2 or 0), device=dev)
        wa, wb = w[:, :k1], (w[:, k1:] if k2 else None)
        bias = torch.randn(n, device=dev)
        g = new_padded(m, n, dev); g.normal_()
        img = Fn._image(wa, wb if xb is not None else None, False)
        if os.environ.get("CHECK"):        # error against float64 on the first, the last and 4096 random rows
            y = nt_call(xa, img, xb, bias, n)
            idx = torch.cat([torch.arange(0, min(m, 2048), device=dev), torch.randint(0, m, (4096,), device=dev),
                             torch.arange(max(m - 2048, 0), m, device=dev)])
            ref = xa[idx].double() @ wa.double().t() + bias.double()
            if k2:
                ref = ref + xb[idx].double() @ wb.double().t()
            err = float((y[idx].double() - ref).abs().max()) / float(ref.abs().max())
            print(f"  check M={m} K=({k1},{k2}) N={n}: max error / max |ref| = {err:.2e}", flush=True)
        t_nt = 1.0 if os.environ.get("TN_ONLY") else timeit(lambda: nt_call(xa, img, xb, bias, n))
        fl = 2.0 * m * (k1 + (k2 or 0)) * n
        t_tn = 1.0 if os.environ.get("NT_ONLY") else timeit(lambda: Fn._gemm_tn(g, xa, True))
        fl_tn = 2.0 * m * n * k1
        print(f"NT[{img.kind}] M={m} K=({k1},{k2}) N={n}: {t_nt:7.1f} us {fl / t_nt / 1e6:6.1f} TF/s | "
              f"TN N={n} K={k1}: {t_tn:7.1f} us {fl_tn / t_tn / 1e6:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
