#!/usr/bin/env python3
"""Accuracy of the level epilogue (relu -> dropout -> batch norm) alone against float64, next to torch's fp32 batch norm, over
row counts and widths."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
DEV = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def main():
    from hgvae_amd import functional as Fn
    gen = torch.Generator(device=DEV).manual_seed(0)
    print("   n     h  relu | out: device, torch fp32 | grad: device, torch fp32 | worst column: n_nonzero, mean/std")
    for n in (64, 755, 5000, 40000, 160984):
        for h in (4, 30, 105, 138, 256):
            for relu in (True, False):
                bn = torch.nn.BatchNorm1d(h).to(DEV)
                y = (torch.randn(n, h, device=DEV, generator=gen) * 2 + 0.5).requires_grad_(True)
                out = Fn.bn_act(y, bn, relu, 0.0, True)
                go = torch.randn(n, h, device=DEV, generator=gen)
                out.backward(go)
                yd = y.detach().double().requires_grad_(True)
                a = torch.relu(yd) if relu else yd
                ref = torch.nn.functional.batch_norm(a, None, None, bn.weight.detach().double(), bn.bias.detach().double(), True, 0.1, bn.eps)
                ref.backward(go.double())
                y32 = y.detach().clone().requires_grad_(True)
                a32 = torch.relu(y32) if relu else y32
                t32 = torch.nn.functional.batch_norm(a32, None, None, bn.weight.detach(), bn.bias.detach(), True, 0.1, bn.eps)
                t32.backward(go)
                err = (out.double() - ref).abs().max(0).values
                c = int(err.argmax())
                col = a.detach()[:, c]
                print(f"{n:7d} {h:4d} {str(relu):5s} | {rel(out, ref):.2e} {rel(t32, ref):.2e} | {rel(y.grad, yd.grad):.2e} {rel(y32.grad, yd.grad):.2e} | "
                      f"{int((col != 0).sum())} {float(col.mean() / col.std()):.2f}")


if __name__ == "__main__":
    main()
