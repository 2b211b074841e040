#!/usr/bin/env python3
"""relu -> dropout -> batch-norm kernels in isolation on the benchmark's activation size (same buffers every
iteration, so inputs can stay in the Infinity Cache): forward and backward, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn  # noqa: E402
from hgvae_amd.graph import new_padded  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    n, h = int(os.environ.get("ROWS", 160984)), int(os.environ.get("H", 138))
    p = float(os.environ.get("P", 0.5))
    y = new_padded(n, h, dev); y.normal_()
    y.requires_grad_(True)
    g = new_padded(n, h, dev); g.normal_()
    bn = torch.nn.BatchNorm1d(h).to(dev).train()
    fwd_only = bool(os.environ.get("FWD_ONLY"))
    for _ in range(30):
        if fwd_only:
            with torch.no_grad():
                out = Fn.bn_act(y, bn, True, p, True)
            continue
        out = Fn.bn_act(y, bn, True, p, True)
        out.backward(g)
        y.grad = None
    torch.cuda.synchronize()
    print("done")


if __name__ == "__main__":
    main()
