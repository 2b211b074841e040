#!/usr/bin/env python3
"""Correctness and timing of the split-bf16 projection kernel against fp64 / the fp32 kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import _lib  # noqa: E402
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd.graph import ld_of, new_padded  # noqa: E402


def nt16(xa, wa, xb, wb, bias, n, t=False):
    lib = _lib.load()
    k1 = xa.size(1)
    k2 = xb.size(1) if xb is not None else 0
    img = torch.empty(lib.hgv_pack16_bytes(n, k1, k2) // 4, dtype=torch.float32, device=xa.device)
    st = _lib.stream()

    def strides(w):
        return (w.stride(1), w.stride(0)) if t else (w.stride(0), w.stride(1))
    sa = strides(wa)
    sb = strides(wb) if wb is not None else (0, 0)
    _lib.call("hgv_pack16_weights", wa.data_ptr(), sa[0], sa[1], k1, _lib.ptr(wb), sb[0], sb[1], k2, n, img.data_ptr(), st)
    out = new_padded(xa.size(0), n, xa.device, zero_pad=False)
    _lib.call("hgv_gemm_nt16", xa.data_ptr(), ld_of(xa), k1, _lib.ptr(xb), ld_of(xb) if xb is not None else 0, k2,
              img.data_ptr(), _lib.ptr(bias), xa.size(0), n, out.data_ptr(), ld_of(out), st)
    return out


def main():
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    shapes = [(1, 6, None, 1), (127, 28, None, 138), (129, 138, 138, 138), (1000, 105, 105, 138), (777, 138, 138, 56),
              (5000, 138, None, 105), (160984, 138, 138, 138), (160984, 138, None, 138), (160984, 105, 105, 138),
              (160984, 138, 138, 56), (70001, 276, None, 138)]
    for m, k1, k2, n in shapes:
        def rnd(r, c):
            t = new_padded(r, c, dev)
            t.copy_(torch.randn(r, c, generator=gen, device=dev))
            return t
        xa, xb = rnd(m, k1), (rnd(m, k2) if k2 else None)
        w = torch.randn(n, k1 + (k2 or 0), generator=gen, device=dev)
        bias = torch.randn(n, generator=gen, device=dev)
        wa, wb = w[:, :k1], (w[:, k1:] if k2 else None)
        if not _lib.load().hgv_gemm_nt16_supported(n, k1, k2 or 0):
            print(m, k1, k2, n, "unsupported")
            continue
        y = nt16(xa, wa, xb, wb, bias, n)
        ref = xa.double() @ wa.double().t() + bias.double()
        if k2:
            ref = ref + xb.double() @ wb.double().t()
        Fn.set_projection("fp32")
        img32 = Fn._image(wa, wb, False)
        Fn.set_projection("bf16x6")
        y32 = Fn._gemm_nt(xa, img32, xb, bias, n)
        scale = float(ref.abs().max())
        e16 = float((y.double() - ref).abs().max()) / scale
        e32 = float((y32.double() - ref).abs().max()) / scale
        pad_ok = True
        if n % 4:
            base = y.as_strided((m, (n + 3) // 4 * 4), (y.stride(0), 1))
            pad_ok = float(base[:, n:].abs().max()) == 0.0

        def timeit(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) * 100.0
        t16 = timeit(lambda: nt16(xa, wa, xb, wb, bias, n)) if m > 10000 else 0.0
        t32 = timeit(lambda: Fn._gemm_nt(xa, img32, xb, bias, n)) if m > 10000 else 0.0
        print(f"M={m} K=({k1},{k2}) N={n}: rel err bf16x6 {e16:.2e} fp32 {e32:.2e} pad_ok={pad_ok}  bf16x6 {t16:.1f} us  fp32 {t32:.1f} us", flush=True)


if __name__ == "__main__":
    main()
