import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hgvae_amd import functional as Fn
from hgvae_amd.graph import new_padded
dev = torch.device("cuda:0")
n = int(os.environ.get("NOUT", "138"))
w = torch.randn(n, 138, device=dev); b = torch.randn(n, device=dev)
for m in (4096 * 16 * 2, 3 * 4096 * 16):
    xa = new_padded(m, 138, dev); xa.normal_()
    img = Fn._image(w, None, False)
    for _ in range(6):
        Fn._gemm_nt(xa, img, None, b, n)
    torch.cuda.synchronize()
