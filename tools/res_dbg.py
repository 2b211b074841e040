import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn
from hgvae_amd.graph import new_padded
dev = torch.device("cuda:0")
m = 160984
xa = new_padded(m, 138, dev); xa.normal_()
xb = new_padded(m, 138, dev); xb.normal_()
w = torch.randn(138, 276, device=dev); b = torch.randn(138, device=dev)
for _ in range(3):
    Fn._gemm_nt(xa, w[:, :138], False, xb, w[:, 138:], b, 138)
torch.cuda.synchronize()
