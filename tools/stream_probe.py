#!/usr/bin/env python3
"""Plain streaming rates on this GPU for arrays of the benchmark's activation size (reference point for the
bandwidth-bound kernels): torch copy (read + write) and sum (read only), back-to-back launches."""
import torch


def rate(fn, nbytes, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    return us, nbytes / us / 1e6


def main():
    dev = torch.device("cuda:0")
    for rows in (160984,):
        for cols in (140, 256):
            a = torch.randn(rows, cols, device=dev)
            b = torch.empty_like(a)
            c = torch.randn(rows, cols, device=dev)
            n = a.numel() * 4
            us, tb = rate(lambda: b.copy_(a), 2 * n)
            print(f"[{rows} x {cols}] copy: {us:6.1f} us  {tb:5.2f} TB/s")
            us, tb = rate(lambda: torch.add(a, c, out=b), 3 * n)
            print(f"[{rows} x {cols}] add : {us:6.1f} us  {tb:5.2f} TB/s")
            us, tb = rate(lambda: a.sum(), n)
            print(f"[{rows} x {cols}] sum : {us:6.1f} us  {tb:5.2f} TB/s")
            # rotating through 8 buffers (640 MB+): nothing can stay in the 256 MB Infinity Cache
            bufs = [torch.randn(rows, cols, device=dev) for _ in range(8)]
            k = [0]

            def rot():
                k[0] = (k[0] + 1) % 8
                bufs[k[0]].copy_(bufs[(k[0] + 4) % 8])
            us, tb = rate(rot, 2 * n)
            print(f"[{rows} x {cols}] copy, rotating buffers: {us:6.1f} us  {tb:5.2f} TB/s")


if __name__ == "__main__":
    main()
