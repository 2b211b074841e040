#!/usr/bin/env python3
"""Random shapes through the projection, aggregation and level-epilogue operators against float64 torch on the device.
  python3 tests/fuzz_ops.py [cases] [seed]     (prints the first failing case and exits 1)
``run(cases, seed)`` is what tests/test_gpu_fuzz.py calls."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DEV = torch.device("cuda:0")


def rel(a, b, scale=None):
    """max |a - b| over the natural scale of the result: max |b|, or for sums of products the largest sum of |terms| (a sum
    that cancels to ~0 has no relative accuracy of its own)."""
    a, b = a.detach().double(), b.detach().double()
    s = float(b.abs().max()) if scale is None else float(scale)
    return float((a - b).abs().max()) / max(s, 1e-30)


def fuzz_linear(rng, gen):
    from hgvae_amd import functional as Fn
    m = rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 100, 257, 1000, 4097, rng.randint(1, 20000)])
    k1 = rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 28, 56, 60, 105, 128, 138, 256, rng.randint(1, 288)])
    k2 = rng.choice([0, 0, 1, 5, 6, 28, 105, 138, 256, rng.randint(1, 288)])
    n = rng.choice([1, 2, 3, 4, 6, 28, 56, 60, 105, 128, 138, 256, rng.randint(1, 288)])
    bias = rng.random() < 0.7
    xa = torch.randn(m, k1, device=DEV, generator=gen).requires_grad_(True)
    wa = (torch.randn(n, k1, device=DEV, generator=gen) * 0.2).requires_grad_(True)
    xb = torch.randn(m, k2, device=DEV, generator=gen).requires_grad_(True) if k2 else None
    wb = (torch.randn(n, k2, device=DEV, generator=gen) * 0.2).requires_grad_(True) if k2 else None
    b = torch.randn(n, device=DEV, generator=gen).requires_grad_(True) if bias else None
    y = Fn.dual_linear(xa, wa, xb, wb, b)
    g = torch.randn(m, n, device=DEV, generator=gen)
    y.backward(g)
    want = xa.detach().double() @ wa.detach().double().t()
    if k2:
        want = want + xb.detach().double() @ wb.detach().double().t()
    if bias:
        want = want + b.detach().double()
    xa64, wa64, g64 = xa.detach().double(), wa.detach().double(), g.double()
    sy = xa64.abs() @ wa64.abs().t()
    if k2:
        sy = sy + xb.detach().double().abs() @ wb.detach().double().abs().t()
    if bias:
        sy = sy + b.detach().double().abs()
    errs = {"y": rel(y, want, sy.max()), "gxa": rel(xa.grad, g64 @ wa64, (g64.abs() @ wa64.abs()).max()),
            "gwa": rel(wa.grad, g64.t() @ xa64, (g64.abs().t() @ xa64.abs()).max())}
    if k2:
        xb64, wb64 = xb.detach().double(), wb.detach().double()
        errs["gxb"] = rel(xb.grad, g64 @ wb64, (g64.abs() @ wb64.abs()).max())
        errs["gwb"] = rel(wb.grad, g64.t() @ xb64, (g64.abs().t() @ xb64.abs()).max())
    if bias:
        errs["gb"] = rel(b.grad, g64.sum(0), g64.abs().sum(0).max())
    return f"linear m={m} k1={k1} k2={k2} n={n} bias={bias}", errs, 1e-6


def fuzz_aggregate(rng, gen):
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    n = rng.choice([1, 2, 7, 64, 65, 1000, rng.randint(1, 30000)])
    e = rng.choice([0, 1, n, 5 * n, rng.randint(0, 12 * n)])
    f = rng.choice([1, 2, 3, 4, 5, 6, 28, 56, 60, 105, 128, 138, 256, 276, rng.randint(1, 600)])
    ei = torch.randint(0, n, (2, e), device=DEV, generator=gen)
    if e and rng.random() < 0.3:                         # leave a range of nodes without in-edges / out-edges
        ei = ei % max(n // 2, 1)
    x = torch.randn(n, f, device=DEV, generator=gen).requires_grad_(True)
    mean = rng.random() < 0.7
    g = Graph(ei, n)
    out = Fn.aggregate(x, g, mean=mean)
    want = torch.zeros(n, f, dtype=torch.float64, device=DEV)
    if e:
        want.index_add_(0, ei[1], x.detach().double()[ei[0]])
    deg = torch.bincount(ei[1], minlength=n).clamp(min=1).double().unsqueeze(1) if mean else 1.0
    want = want / deg
    go = torch.randn(n, f, device=DEV, generator=gen)
    out.backward(go)
    wg = torch.zeros(n, f, dtype=torch.float64, device=DEV)
    if e:
        wg.index_add_(0, ei[0], (go.double() / deg)[ei[1]])
    return f"aggregate n={n} e={e} f={f} mean={mean}", {"out": rel(out, want), "gx": rel(x.grad, wg) if float(wg.abs().max()) > 0 else float(x.grad.abs().max())}, 2e-6


def fuzz_bn(rng, gen):
    from hgvae_amd import functional as Fn
    n = rng.choice([16, 17, 255, 1000, rng.randint(16, 40000)])   # a handful of rows: the input gradient cancels to ~0, no relative measure
    h = rng.choice([1, 2, 3, 4, 6, 28, 60, 105, 128, 138, 256, rng.randint(1, 288)])
    relu = rng.random() < 0.6
    bn = torch.nn.BatchNorm1d(h).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.randn(h, device=DEV, generator=gen))
        bn.bias.copy_(torch.randn(h, device=DEV, generator=gen))
    y = (torch.randn(n, h, device=DEV, generator=gen) * 2 + 0.5).requires_grad_(True)
    out = Fn.bn_act(y, bn, relu, 0.0, True)
    go = torch.randn(n, h, device=DEV, generator=gen)
    out.backward(go)
    yd = y.detach().double().requires_grad_(True)
    a = torch.relu(yd) if relu else yd
    w64, b64 = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
    ref = torch.nn.functional.batch_norm(a, None, None, w64, b64, True, 0.1, bn.eps)
    ref.backward(go.double())
    errs = {"out": rel(out, ref), "gy": rel(y.grad, yd.grad), "gw": rel(bn.weight.grad, w64.grad), "gb": rel(bn.bias.grad, b64.grad)}
    # running statistics after the step, then the evaluation-mode path that uses them (forward and input gradient)
    m64 = a.detach().mean(0)
    errs["running_mean"] = rel(bn.running_mean, 0.1 * m64, max(float(m64.abs().max()), 1e-3))
    errs["running_var"] = rel(bn.running_var, 0.9 + 0.1 * a.detach().var(0, unbiased=True))
    bn.eval()
    y2 = y.detach().clone().requires_grad_(True)
    out2 = Fn.bn_act(y2, bn, relu, 0.3, False)
    out2.backward(go)
    y2d = y.detach().double().requires_grad_(True)
    a2 = torch.relu(y2d) if relu else y2d
    ref2 = torch.nn.functional.batch_norm(a2, bn.running_mean.double(), bn.running_var.double(), w64.detach(), b64.detach(), False, 0.1, bn.eps)
    ref2.backward(go.double())
    errs["eval_out"] = rel(out2, ref2)
    errs["eval_gy"] = rel(y2.grad, y2d.grad)
    return f"bn n={n} h={h} relu={relu}", errs, 2e-5


def fuzz_csr(rng, gen):
    """Both CSR directions against torch's stable sort of the edge list (integer work: exact)."""
    from hgvae_amd.graph import Graph
    n = rng.choice([1, 2, 255, 256, 257, 4096, rng.randint(1, 200000)])
    e = rng.choice([0, 1, 2, n, rng.randint(0, 8 * n)])
    ei = torch.randint(0, n, (2, e), device=DEV, generator=gen)
    if rng.random() < 0.3:
        ei = ei.t().contiguous().t()                       # [2, E] view of an [E, 2] buffer
    g = Graph(ei, n)
    bad = 0
    for csr, key_row, other_row in ((g.by_dst, 1, 0), (g.by_src, 0, 1)):
        order = torch.sort(ei[key_row], stable=True).indices
        rp = torch.zeros(n + 1, dtype=torch.long, device=DEV)
        rp[1:] = torch.cumsum(torch.bincount(ei[key_row], minlength=n), 0)
        ok = torch.equal(csr.perm.long(), order) and torch.equal(csr.col.long(), ei[other_row][order]) and \
            torch.equal(csr.rowptr.long(), rp)
        bad += 0 if ok else 1
    return f"csr n={n} e={e}", {"mismatch": float(bad)}, 0.0


def fuzz_pool(rng, gen):
    """Per-graph top-k (ties included) and filter_adj against the CPU restatement (integer results: exact)."""
    from oracle import pyg_ops as O
    from hgvae_amd import functional as Fn
    b = rng.choice([1, 2, 5, 33])
    sizes = [rng.choice([1, 2, 3, 100, 1023, 1024, 1025, rng.randint(1, 6000)]) for _ in range(b)]
    n = sum(sizes)
    batch = torch.repeat_interleave(torch.arange(b), torch.tensor(sizes))
    cpu_gen = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    score = torch.tanh(torch.randn(n, generator=cpu_gen))
    if rng.random() < 0.5:
        score = (score * 4).round() / 4                    # many ties
    ratio = rng.choice([0.5, 0.3, 1.0, 0.01])
    ref = O.topk(score, ratio, batch)
    bad = 0
    for global_sort in (False, True):
        perm = Fn.topk_perm(score.to(DEV), ratio, batch.to(DEV), global_sort=global_sort)
        bad += 0 if torch.equal(perm.cpu(), ref) else 1
    e = rng.randint(0, 6 * n)
    ei = torch.randint(0, n, (2, e), generator=cpu_gen)
    ea = torch.randn(e, 1, generator=cpu_gen)
    ref_ei, ref_ea = O.filter_adj(ei, ea, ref, n)
    out_ei, out_ea = Fn.filter_adj(ei.to(DEV), ea.to(DEV), ref.to(DEV), n)
    bad += 0 if (torch.equal(out_ei.cpu(), ref_ei) and torch.equal(out_ea.cpu(), ref_ea)) else 1
    return f"pool sizes={sizes} ratio={ratio} e={e}", {"mismatch": float(bad)}, 0.0


def fuzz_gat(rng, gen):
    """Single-head graph attention (no attention dropout) against a masked dense softmax in float64: duplicate edges, self loops
    in the input (replaced by the layer's own), isolated nodes."""
    from oracle import pyg_ops as O
    from hgvae_amd.nn import GATConv
    n = rng.choice([1, 2, 17, 200, rng.randint(1, 900)])
    e = rng.choice([0, n, rng.randint(0, 10 * n)])
    f_in, c = rng.choice([1, 3, 30, 138]), rng.choice([1, 4, 30, 138, 256])
    cpu_gen = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    ei = torch.randint(0, n, (2, e), generator=cpu_gen)
    x = torch.randn(n, f_in, generator=cpu_gen)
    torch.manual_seed(rng.randint(0, 1 << 30))
    conv = GATConv(f_in, c)
    with torch.no_grad():
        conv.bias.copy_(torch.randn(c, generator=cpu_gen))
        conv.att_src.mul_(rng.choice([1.0, 8.0]))          # sharper softmax
    ref = {k: v.detach().double().requires_grad_(True) for k, v in conv.named_parameters()}
    xr = x.double().requires_grad_(True)
    want = O.dense_gat(xr, ei, ref["lin_src.weight"], ref["att_src"], ref["att_dst"], ref["bias"])
    go = torch.randn(n, c, generator=cpu_gen)
    want.backward(go.double())
    # the per-edge algorithm (what PyG runs) in fp32 on the CPU: a sharp softmax makes the attention gradients differences of
    # terms that cancel - per edge, so that parallel edges do not merge as they do in the dense form - and fp32 arithmetic
    # then sits far (relatively) from float64: the bar follows it (x4)
    cpu = O.GATConv(f_in, c)
    cpu.load_state_dict({k: v.detach().clone() for k, v in conv.state_dict().items()})
    ref32 = dict(cpu.named_parameters())
    x32 = x.clone().requires_grad_(True)
    cpu(x32, ei).backward(go)
    conv = conv.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    out = conv(xd, ei.to(DEV))
    out.backward(go.to(DEV))
    tol = 2e-5
    errs = {"out": rel(out.cpu(), want)}
    # a shift of all logits of a row leaves its softmax unchanged, so the gradient of att_dst is a sum that cancels to (almost)
    # zero wherever the leaky slope is the same along the row: both attention vectors are measured on their common scale
    att_scale = max(float(ref["att_src"].grad.abs().max()), float(ref["att_dst"].grad.abs().max()), 1e-6)
    pairs = [("gx", xd.grad, xr.grad, x32.grad, max(float(xr.grad.abs().max()), 1e-6))]
    # the projection's weight gradient is x^T (gradient of the projected features): on the scale of the sum of its |terms|
    # (a 1 x 1 layer over a few hundred nodes is one sum that may cancel)
    xw = (x.double() @ ref["lin_src.weight"].detach().t()).requires_grad_(True)
    O.dense_gat(xw, ei, torch.eye(c, dtype=torch.float64), ref["att_src"].detach(), ref["att_dst"].detach(),
                ref["bias"].detach()).backward(go.double())
    lin_terms = float((xw.grad.abs().t() @ x.double().abs()).max())
    for k, v in conv.named_parameters():
        scale = att_scale if k.startswith("att_") else max(float(ref[k].grad.abs().max()), 1e-6)
        if k == "lin_src.weight":
            scale = max(scale, lin_terms)
        pairs.append(("g_" + k.split(".")[0], v.grad, ref[k].grad, ref32[k].grad, scale))
    raw = {}
    for name, got, r64, r32, scale in pairs:
        raw[name] = (rel(got.cpu(), r64, scale), rel(r32, r64, scale))
        errs[name] = raw[name][0] / max(1.0, 4 * raw[name][1] / tol)
    if os.environ.get("FUZZ_GAT_DETAIL"):
        print(f"  gat detail n={n} e={e} f_in={f_in} c={c}: att scale {att_scale:.3e}, |go| {float(go.abs().max()):.2e}, out scale "
              f"{float(want.abs().max()):.2e}, dev/cpu32 errs " + " ".join(f"{k}={v[0]:.1e}/{v[1]:.1e}" for k, v in raw.items()), flush=True)
    return f"gat n={n} e={e} f_in={f_in} c={c}", errs, tol


def fuzz_convs(rng, gen):
    """SAGEConv / GraphConv / GCNConv modules (outputs, input and parameter gradients) against the CPU restatement in float64
    on random multigraphs with self loops and isolated nodes."""
    from oracle import pyg_ops as O
    from hgvae_amd import nn as HN
    kind = rng.choice(["SAGEConv", "GraphConv", "GCNConv"])
    n = rng.choice([1, 2, 33, 500, rng.randint(1, 3000)])
    e = rng.choice([0, n, rng.randint(0, 10 * n)])
    f_in, f_out = rng.choice([1, 3, 6, 60, 105, 138, 276]), rng.choice([1, 2, 28, 56, 138, 276])
    cpu_gen = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    ei = torch.randint(0, n, (2, e), generator=cpu_gen)
    if e and rng.random() < 0.3:
        ei = ei % max(n // 3, 1)                           # most nodes isolated
    x = torch.randn(n, f_in, generator=cpu_gen)
    torch.manual_seed(rng.randint(0, 1 << 30))
    ref = getattr(O, kind)(f_in, f_out).double()
    mod = getattr(HN, kind)(f_in, f_out)
    mod.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    mod = mod.to(DEV)
    xr = x.double().requires_grad_(True)
    want = ref(xr, ei)
    go = torch.randn(n, f_out, generator=cpu_gen)
    want.backward(go.double())
    xd = x.to(DEV).requires_grad_(True)
    out = mod(xd, ei.to(DEV))
    out.backward(go.to(DEV))
    # scale of a sum of products: the same network on absolute values
    with torch.no_grad():
        aref = getattr(O, kind)(f_in, f_out).double()
        aref.load_state_dict({k: v.abs() for k, v in ref.state_dict().items()})
        scale_out = float(aref(x.double().abs(), ei).abs().max())
    errs = {"out": rel(out.cpu(), want, max(scale_out, 1e-30)),
            "gx": rel(xd.grad.cpu(), xr.grad, max(float(xr.grad.abs().max()), 0.1 * float(go.abs().max())))}
    rp = dict(ref.named_parameters())
    for k, v in mod.named_parameters():
        errs["g_" + k] = rel(v.grad.cpu(), rp[k].grad, max(float(rp[k].grad.abs().max()), 0.1 * float(go.abs().max())))
    return f"{kind} n={n} e={e} f_in={f_in} f_out={f_out}", errs, 5e-6


def fuzz_losses(rng, gen):
    """Reparameterisation, KL term and mean squared error (values and gradients) against float64, with log-std values
    far from zero."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    n = rng.choice([1, 2, 63, 1000, rng.randint(1, 30000)])
    z, ns = rng.choice([1, 2, 28, 56]), rng.choice([1, 3, 10])
    spread = rng.choice([0.1, 1.0, 3.0])
    ml = torch.randn(n, 2 * z, device=DEV, generator=gen)
    ml[:, z:] *= spread
    ml.requires_grad_(True)
    eps = torch.randn(ns, n, z, device=DEV, generator=gen)
    with use_noise(NoiseTape([eps.cpu()])):
        zz = Fn.reparameterize(ml, z, ns)
    kl = Fn.kl_divergence(ml, z)
    gz = torch.randn(n, z, device=DEV, generator=gen)
    (zz * gz).sum().backward(retain_graph=True)
    g_rep = ml.grad.clone(); ml.grad = None
    kl.backward()
    g_kl = ml.grad.clone()
    m64 = ml.detach().double().requires_grad_(True)
    mean, ls = m64[:, :z], m64[:, z:]
    z64 = (mean.unsqueeze(0) + ls.exp().unsqueeze(0) * eps.double()).mean(0)
    kl64 = (-0.5 * torch.sum(1 + 2 * ls - mean.pow(2) - ls.exp().pow(2), dim=-1)).mean()
    kl_terms = (0.5 * torch.sum(1 + 2 * ls.abs() + mean.pow(2) + ls.exp().pow(2), dim=-1)).mean().detach()
    (z64 * gz.double()).sum().backward(retain_graph=True)
    r_rep = m64.grad.clone(); m64.grad = None
    kl64.backward()
    r_kl = m64.grad.clone()
    a = torch.randn(n, z, device=DEV, generator=gen).requires_grad_(True)
    b = torch.randn(n, z, device=DEV, generator=gen)
    mse = Fn.mse_loss(a, b)
    mse.backward()
    a64 = a.detach().double().requires_grad_(True)
    mse64 = torch.nn.functional.mse_loss(a64, b.double())
    mse64.backward()
    errs = {"z": rel(zz, z64), "g_reparam": rel(g_rep, r_rep),
            # on the scale of the terms it sums (1 + 2 ls - mean^2 - exp(2 ls) cancels for small ls, mean: a one-element
            # batch has no relative accuracy of its own)
            "kl": abs(float(kl.detach()) - float(kl64.detach())) / max(float(kl_terms), 1e-30), "g_kl": rel(g_kl, r_kl),
            "mse": abs(float(mse.detach()) - float(mse64.detach())) / max(float(mse64.detach()), 1e-30), "g_mse": rel(a.grad, a64.grad)}
    return f"losses n={n} z={z} ns={ns} spread={spread}", errs, 2e-6


def fuzz_adam(rng, gen):
    """A few optimiser steps on random parameter shapes against torch.optim.Adam + clip_grad_norm_ in float64; torch's own
    fp32 run of the same steps sets how far fp32 arithmetic may land (clipped, tiny gradients against a large eps are touchy)."""
    from hgvae_amd.optim import FlatAdam
    shapes = [tuple(rng.randint(1, 40) for _ in range(rng.choice([1, 2]))) for _ in range(rng.randint(1, 6))]
    lr, wd = rng.choice([1e-3, 1e-2, 0.3]), rng.choice([0.0, 0.0, 1e-2])
    eps, betas = rng.choice([1e-8, 1e-5]), rng.choice([(0.9, 0.999), (0.5, 0.9)])
    clip = rng.choice([None, None, 0.5, 10.0])
    ps = [torch.nn.Parameter(torch.randn(sh, device=DEV, generator=gen)) for sh in shapes]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    rs = [torch.nn.Parameter(p.detach().double().clone()) for p in ps]
    opt = FlatAdam(ps, lr=lr, betas=betas, eps=eps, weight_decay=wd, max_grad_norm=clip)
    t32 = torch.optim.Adam(qs, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    t64 = torch.optim.Adam(rs, lr=lr, betas=betas, eps=eps, weight_decay=wd)
    for _ in range(rng.randint(1, 5)):
        for p, q, r in zip(ps, qs, rs):
            g = torch.randn(p.shape, device=DEV, generator=gen) * rng.choice([1e-3, 1.0, 30.0])
            p.grad = g.clone() if p.grad is None else p.grad.copy_(g)
            q.grad, r.grad = g.clone(), g.double()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(qs, clip)
            torch.nn.utils.clip_grad_norm_(rs, clip)
        opt.step(); t32.step(); t64.step()
    dev_err = max(rel(p, r) for p, r in zip(ps, rs))
    cpu_err = max(rel(q, r) for q, r in zip(qs, rs))
    tol = 1e-6
    return (f"adam shapes={shapes} lr={lr} wd={wd} eps={eps} betas={betas} clip={clip} (torch fp32 {cpu_err:.1e})",
            {"param": dev_err / max(1.0, 4 * cpu_err / tol)}, tol)


def fuzz_model(rng, gen):
    """One train step (loss and every parameter gradient) of one of the four model variants on a random batch of random
    graphs with random widths against the float64 CPU restatement driven by the same noise tape."""
    from oracle import models as OM
    from oracle import pyg_ops as O
    from hgvae_amd import vae as V
    from hgvae_amd.noise import NoiseTape, use_noise
    from test_gpu_model import _fresh_state, pool_tape, randomize_bn, skipcon_tape
    kind = rng.choice(["skipcon", "skipcon", "attention", "pool", "custom_pool"])
    f_in, h = rng.choice([3, 6, 60, 105]), rng.choice([4, 30, 64, 138, 200])
    if kind in ("pool", "custom_pool"):
        h = min(h, 138)                                    # their second level is 2h wide; projections stop at 288 columns
    z, ns = rng.choice([1, 2, 28]), rng.choice([1, 3, 10])
    sizes = [rng.choice([1, 2, 40, 138, rng.randint(1, 700)]) for _ in range(rng.choice([1, 3, 8]))]
    n = max(sum(sizes), 64)
    sizes[-1] += n - sum(sizes)
    cpu_gen = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    e = rng.choice([0, n, rng.randint(0, 8 * n)])
    src = torch.randint(0, n, (e,), generator=cpu_gen)
    lo = torch.cumsum(torch.tensor([0] + sizes[:-1]), 0)[batch[src]]                     # keep every edge inside its graph
    dst = lo + (torch.rand(e, generator=cpu_gen) * torch.tensor(sizes)[batch[src]]).long()
    ei = torch.stack([src, dst])
    if kind == "attention":
        ei = ei[:, src != dst]                             # the attention layer replaces self loops by its own: keep the count simple
    e = ei.size(1)
    x = torch.randn(n, f_in, generator=cpu_gen)
    torch.manual_seed(rng.randint(0, 1 << 30))
    if kind == "skipcon":
        make, dev_cls = (lambda: OM.SkipconVAE(f_in, h, z, ns)), V.SkipconVAE
        tape = None
        ref_loss = lambda m, xx, t: OM.skipcon_loss(m, xx, ei, batch, 1.0, t)[0]       # noqa: E731
        dev_loss = lambda m, xx, ee, bb: V.train_step_loss(m, xx, ee, bb, 1.0)           # noqa: E731
        mk_tape = lambda: skipcon_tape(n, h, z, ns, 0.5, cpu_gen)                        # noqa: E731
    elif kind == "attention":
        make, dev_cls = (lambda: OM.SkipconAttentionVAE(f_in, h, z, ns)), V.SkipconAttentionVAE
        ref_loss = lambda m, xx, t: OM.skipcon_loss(m, xx, ei, batch, 1.0, t)[0]       # noqa: E731
        dev_loss = lambda m, xx, ee, bb: V.train_step_loss(m, xx, ee, bb, 1.0)           # noqa: E731
        mk_tape = lambda: skipcon_tape(n, h, z, ns, 0.49, cpu_gen, gat_edges=e + n)      # noqa: E731
    elif kind == "pool":
        make, dev_cls = (lambda: OM.PoolVAE(f_in, h, z, ns)), V.PoolVAE
        ref_loss = lambda m, xx, t: OM.pool_loss(m, xx, ei, batch, 1.0, t)[0]          # noqa: E731
        dev_loss = lambda m, xx, ee, bb: V.train_step_loss(m, xx, ee, bb, 1.0)           # noqa: E731
        mk_tape = lambda: pool_tape(n, h, z, ns, cpu_gen)                                # noqa: E731
    else:
        make, dev_cls = (lambda: OM.CustomPoolVAE(f_in, h, z, ns, beta=0.5)), (lambda *a: V.CustomPoolVAE(*a, beta=0.5))
        ref_loss = lambda m, xx, t: OM.custom_pool_loss(m, xx, ei, batch, t)[0]        # noqa: E731
        dev_loss = lambda m, xx, ee, bb: V.custom_pool_step_loss(m, xx, ee, bb)          # noqa: E731
        mk_tape = lambda: pool_tape(n, h, z, ns, cpu_gen, second_encode=False, second_decode=True)   # noqa: E731
    ref32 = make()
    randomize_bn(ref32, cpu_gen)
    state = _fresh_state(ref32)
    tape = mk_tape()
    ref = make().double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
    ref.train()
    mod = dev_cls(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(DEV).train()
    ref32.load_state_dict(state)
    ref32.train()
    # sometimes a random part of the parameters is frozen and / or the input asks for its gradient
    frozen = set()
    if rng.random() < 0.35:
        frozen = {name for name, _ in mod.named_parameters() if rng.random() < 0.35}
        for m in (ref, ref32, mod):
            for name, prm in m.named_parameters():
                if name in frozen:
                    prm.requires_grad_(False)
    want_gx = rng.random() < 0.35
    x64, x32, xd = x.double(), x.clone(), x.to(DEV)
    if want_gx:
        x64.requires_grad_(True); x32.requires_grad_(True); xd.requires_grad_(True)
    tape64 = [t.double() if t.is_floating_point() else t for t in tape]
    loss_ref = ref_loss(ref, x64, O.NoiseTape(tape64))
    loss_ref.backward()
    with use_noise(NoiseTape(tape)):
        loss = dev_loss(mod, xd, ei.to(DEV), batch.to(DEV))
    loss.backward()
    # the same step by the fp32 CPU restatement: how far plain fp32 arithmetic lands from float64 on this case
    # (narrow layers on few nodes are ill-conditioned through the batch statistics); the device path may be 4x that
    loss32 = ref_loss(ref32, x32, O.NoiseTape(tape))
    loss32.backward()
    rp, rp32 = dict(ref.named_parameters()), dict(ref32.named_parameters())
    floor = 1e-6 * max(abs(float(loss_ref.detach())), 1.0)
    dev_err = cpu_err = 0.0
    for name, prm in mod.named_parameters():
        if rp[name].grad is None:                          # frozen, or outside the loss (the discarded pooling branch)
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        assert prm.grad is not None, name
        r = rp[name].grad.double()
        scale = max(float(r.abs().max()), floor)
        dev_err = max(dev_err, float((prm.grad.double().cpu() - r).abs().max()) / scale)
        cpu_err = max(cpu_err, float((rp32[name].grad.double() - r).abs().max()) / scale)
    if want_gx:
        assert xd.grad is not None
        scale = max(float(x64.grad.abs().max()), floor)
        dev_err = max(dev_err, float((xd.grad.double().cpu() - x64.grad).abs().max()) / scale)
        cpu_err = max(cpu_err, float((x32.grad.double() - x64.grad).abs().max()) / scale)
    lscale = max(abs(float(loss_ref.detach())), 1e-30)
    dev_l, cpu_l = abs(float(loss.detach()) - float(loss_ref.detach())) / lscale, abs(float(loss32.detach()) - float(loss_ref.detach())) / lscale
    # bars: loss 1e-5, gradients 2e-4 (reported on the gradient scale, hence the factor 20), each widened to 4x the fp32 CPU error
    # A widened bar is capped at WIDEN_CAP x: beyond that the case says nothing about the device path and is reported as
    # a failure of the harness rather than passed.  run() counts the cases that needed any widening.
    widen_l, widen_g = min(WIDEN_CAP, max(1.0, 4 * cpu_l / 1e-5)), min(WIDEN_CAP, max(1.0, 4 * cpu_err / 2e-4))
    errs = {"loss": 20 * dev_l / widen_l, "grad": dev_err / widen_g}
    if errs["grad"] > 2e-4:
        # Before calling it a failure: is the case decided by roundings?  Narrow levels over a few hundred rows, next to
        # one- and two-node graphs, can hold a batch-norm column that is almost constant (1 / std up to 316) in front of a
        # relu: a perturbation of a few fp32 ulps then flips one mask bit and moves some gradient by a fixed 1e-3 .. 1e-2,
        # under every arithmetic (measured: additive noise of 3e-7 on the first projection's output reproduces the exact
        # same error for two noise seeds out of four).  The float64 restatement itself is asked: its gradients on inputs
        # perturbed by 2e-7 (relative) - if THEY move by more than the bar, no fp32 path can be held to it; the bar then
        # follows 4x that movement, as it follows the fp32 CPU restatement's own error.
        cond = 0.0
        for _ in range(4):
            refp = make().double()
            refp.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
            refp.train()
            for name, prm in refp.named_parameters():
                if name in frozen:
                    prm.requires_grad_(False)
            xp = x.double() * (1 + 2e-7 * torch.randn(x.shape, generator=cpu_gen).double())
            if want_gx:
                xp.requires_grad_(True)
            ref_loss(refp, xp, O.NoiseTape(tape64)).backward()
            for name, prm in refp.named_parameters():
                if rp[name].grad is not None:
                    r = rp[name].grad.double()
                    cond = max(cond, float((prm.grad.double() - r).abs().max()) / max(float(r.abs().max()), floor))
            if want_gx:
                cond = max(cond, float((xp.grad - x64.grad).abs().max()) / max(float(x64.grad.abs().max()), floor))
        widen_g = min(WIDEN_CAP, max(1.0, 4 * cpu_err / 2e-4, 4 * cond / 2e-4))
        errs["grad"] = dev_err / widen_g
        errs["fragile_cases"] = 0.0 if cond <= 2e-4 else 1e-300        # shows up in the worst-error summary when it happened
    errs["_widening"] = max(widen_l, widen_g)                           # bookkeeping for run(), not an error
    if os.environ.get("FUZZ_MODEL_DETAIL"):
        print(f"  model detail: {kind} loss f64 {float(loss_ref.detach()):.9g}; rel dev {dev_l:.2e} cpu32 {cpu_l:.2e}; "
              f"grad dev {dev_err:.2e} cpu32 {cpu_err:.2e}; h={h} n={n} e={e}", flush=True)
    return (f"model {kind} f_in={f_in} h={h} z={z} ns={ns} sizes={sizes} e={e} frozen={sorted(frozen)} input_grad={want_gx} "
            f"(fp32 CPU grad err {cpu_err:.1e})"), errs, 2e-4


WIDEN_CAP = 50.0              # largest factor by which a whole-step bar may follow the fp32 CPU restatement / the conditioning
# Largest share of a seed's whole-step cases that may need a widened bar (at least one case).  A seed of ~330 cases holds
# 22-28 whole-step cases; measured on the committed seeds 11 / 12 / 5 / 15: 1 of 24, 0 of 23, 2 of 28 (one of them at the
# cap), 0 of 22 - 3 % over all four.
WIDENED_SHARE = 0.08


def run(cases: int, seed: int, verbose: bool = False):
    """(failure message or None, worst error per quantity).  The whole-step cases whose bar had to be widened are
    counted (``fuzz_model.widened_cases`` of ``fuzz_model.cases``, largest factor ``fuzz_model.max_widening``); more of
    them than WIDENED_SHARE of the seed's whole-step cases is a failure: a harness that widens often tests nothing."""
    rng = random.Random(seed)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    worst = {}
    model_cases = widened = 0
    for i in range(cases):
        fn = rng.choice([fuzz_linear, fuzz_linear, fuzz_aggregate, fuzz_bn, fuzz_csr, fuzz_pool, fuzz_model, fuzz_gat, fuzz_losses, fuzz_adam, fuzz_convs, fuzz_convs])
        name, errs, tol = fn(rng, gen)
        w = errs.pop("_widening", None)
        if w is not None:
            model_cases += 1
            widened += 1 if w > 1.0 else 0
            worst["fuzz_model.max_widening"] = max(worst.get("fuzz_model.max_widening", 1.0), w)
            worst["fuzz_model.cases"], worst["fuzz_model.widened_cases"] = float(model_cases), float(widened)
        for k, v in errs.items():
            key = fn.__name__ + "." + k
            worst[key] = max(worst.get(key, 0.0), v)
            if not v <= tol:
                return f"case {i} (seed {seed}): {name}: {k} = {v:.3e} (> {tol:g})", worst
        if verbose and i % 50 == 49:
            print(f"{i + 1} cases ok", flush=True)
    if widened > max(1, int(WIDENED_SHARE * model_cases + 0.5)):
        return (f"seed {seed}: {widened} of {model_cases} whole-step cases needed a widened bar "
                f"(more than {WIDENED_SHARE:.0%})"), worst
    return None, worst


def run_models(cases: int, seed: int):
    """Whole-step cases only (fuzz_model): (failure message or None, worst errors, cases whose bar was widened)."""
    rng = random.Random(seed)
    gen = torch.Generator(device=DEV).manual_seed(seed)
    worst, widened = {}, 0
    for i in range(cases):
        name, errs, tol = fuzz_model(rng, gen)
        w = errs.pop("_widening", None)
        widened += 1 if (w is not None and w > 1.0) else 0
        for k, v in errs.items():
            worst[k] = max(worst.get(k, 0.0), v)
            if not v <= tol:
                return f"case {i} (seed {seed}): {name}: {k} = {v:.3e} (> {tol:g})", worst, widened
    return None, worst, widened


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    fail, worst = run(cases, seed, verbose=True)
    if fail:
        print("FAIL", fail)
        sys.exit(1)
    print("worst relative errors:", {k: f"{v:.2e}" for k, v in sorted(worst.items())})
    print(f"whole-step cases: {int(worst.get('fuzz_model.cases', 0))}, widened bars: {int(worst.get('fuzz_model.widened_cases', 0))}, "
          f"largest widening {worst.get('fuzz_model.max_widening', 1.0):.1f}x (cap {WIDEN_CAP:.0f}x)")


if __name__ == "__main__":
    main()
