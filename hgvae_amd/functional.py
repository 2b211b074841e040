"""Autograd operators over the HIP kernels (forward and hand-written backward).

Every function here launches kernels from ``libhgvae_hip.so`` on the current
stream; dense projections go through ``torch.mm`` into padded buffers.  Nothing
falls back to a CPU or eager re-implementation.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .graph import Graph, as_padded, ld_of, new_padded, rowmajor
from .noise import current_noise

_ws_cache: "dict[tuple, torch.Tensor]" = {}


def _workspace(nbytes: int, device, tag: str) -> torch.Tensor:
    """Per-(device, stream, tag) scratch buffer, grown on demand and reused across calls."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 4096), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# --------------------------------------------------------------------------------------
# neighbour aggregation
# --------------------------------------------------------------------------------------
def _agg_launch(csr, n_rows, x, F, row_mode, row_scale, edge_w, nbr_w, add, out):
    _lib.call("hgv_agg_fwd", csr.rowptr.data_ptr(), csr.col.data_ptr(), n_rows, x.data_ptr(), ld_of(x), F, row_mode,
              _lib.ptr(row_scale), _lib.ptr(edge_w), _lib.ptr(nbr_w), _lib.ptr(add),
              ld_of(add) if add is not None else 0, out.data_ptr(), ld_of(out), _lib.stream())


class _Aggregate(torch.autograd.Function):
    """out_i = (mean | sum) over in-neighbours j of x_j."""

    @staticmethod
    def forward(ctx, x, graph: Graph, mean: bool):
        _lib.require_cuda(x)
        x = as_padded(x)
        out = new_padded(graph.num_nodes, x.size(1), x.device, zero_pad=False)
        _agg_launch(graph.by_dst, graph.num_nodes, x, x.size(1), 1 if mean else 0, None, None, None, None, out)
        ctx.graph, ctx.mean = graph, mean
        return out

    @staticmethod
    def backward(ctx, g):
        graph = ctx.graph
        g = as_padded(g)
        gx = new_padded(graph.num_nodes, g.size(1), g.device, zero_pad=False)
        _agg_launch(graph.by_src, graph.num_nodes, g, g.size(1), 0, None, None,
                    graph.inv_deg if ctx.mean else None, None, gx)
        return gx, None, None


def aggregate(x: torch.Tensor, graph: Graph, mean: bool = True) -> torch.Tensor:
    if x.size(0) != graph.num_nodes:
        raise _lib.HgvError(f"x has {x.size(0)} rows but the graph has {graph.num_nodes} nodes")
    return _Aggregate.apply(x, graph, mean)


class _WeightedAggregate(torch.autograd.Function):
    """out_i = sum_k w_k x[col_k] over the by-target CSR; w is per CSR position (no grad for w here)."""

    @staticmethod
    def forward(ctx, x, graph: Graph, w):
        x = as_padded(x)
        out = new_padded(graph.num_nodes, x.size(1), x.device, zero_pad=False)
        _agg_launch(graph.by_dst, graph.num_nodes, x, x.size(1), 0, None, w, None, None, out)
        ctx.graph = graph
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, g):
        graph = ctx.graph
        (w,) = ctx.saved_tensors
        g = as_padded(g)
        w_t = w.index_select(0, graph.src_pos_to_dst_pos)
        gx = new_padded(graph.num_nodes, g.size(1), g.device, zero_pad=False)
        _agg_launch(graph.by_src, graph.num_nodes, g, g.size(1), 0, None, w_t, None, None, gx)
        return gx, None, None


# --------------------------------------------------------------------------------------
# dense projection pair:  y = xa @ Wa^T + xb @ Wb^T + bias   (xb / Wb optional)
# --------------------------------------------------------------------------------------
class _DualLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xa, wa, xb, wb, bias):
        _lib.require_cuda(xa, wa)
        xa = rowmajor(xa)
        y = new_padded(xa.size(0), wa.size(0), xa.device)
        if bias is not None:
            torch.addmm(bias, xa, wa.t(), out=y)
        else:
            torch.mm(xa, wa.t(), out=y)
        if xb is not None:
            xb = rowmajor(xb)
            torch.addmm(y, xb, wb.t(), out=y)
        ctx.has_b, ctx.has_bias = xb is not None, bias is not None
        ctx.save_for_backward(xa, wa, xb, wb)
        return y

    @staticmethod
    def backward(ctx, g):
        xa, wa, xb, wb = ctx.saved_tensors
        g = rowmajor(g)
        need = ctx.needs_input_grad
        gxa = gwa = gxb = gwb = gb = None
        if need[0]:
            gxa = new_padded(g.size(0), wa.size(1), g.device)
            torch.mm(g, wa, out=gxa)
        if need[1]:
            gwa = torch.mm(g.t(), xa)
        if ctx.has_b:
            if need[2]:
                gxb = new_padded(g.size(0), wb.size(1), g.device)
                torch.mm(g, wb, out=gxb)
            if need[3]:
                gwb = torch.mm(g.t(), xb)
        if ctx.has_bias and need[4]:
            gb = g.sum(0)
        return gxa, gwa, gxb, gwb, gb


def dual_linear(xa, wa, xb=None, wb=None, bias=None):
    return _DualLinear.apply(xa, wa, xb, wb, bias)


# --------------------------------------------------------------------------------------
# (relu) -> dropout -> batch norm
# --------------------------------------------------------------------------------------
class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, p, mask, seed, offset, eps, momentum):
        _lib.require_cuda(y)
        y = rowmajor(y)
        n, h = y.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        out = new_padded(n, h, y.device, zero_pad=False)
        save_mean = torch.empty(h, dtype=torch.float32, device=y.device)
        save_rstd = torch.empty(h, dtype=torch.float32, device=y.device)
        _lib.call("hgv_bn_act_fwd", y.data_ptr(), ld_of(y), n, h, int(relu), float(p), _lib.ptr(mask), seed, offset,
                  _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(running_mean), _lib.ptr(running_var),
                  float(momentum), save_mean.data_ptr(), save_rstd.data_ptr(), out.data_ptr(), ld_of(out),
                  ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.cfg = (relu, p, seed, offset)
        ctx.save_for_backward(y, gamma, save_mean, save_rstd, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        y, gamma, save_mean, save_rstd, mask = ctx.saved_tensors
        relu, p, seed, offset = ctx.cfg
        g = rowmajor(g)
        n, h = y.shape
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        dy = new_padded(n, h, y.device, zero_pad=False)
        dgamma = torch.empty(h, dtype=torch.float32, device=y.device)
        dbeta = torch.empty(h, dtype=torch.float32, device=y.device)
        _lib.call("hgv_bn_act_bwd", g.data_ptr(), ld_of(g), y.data_ptr(), ld_of(y), n, h, int(relu), float(p),
                  _lib.ptr(mask), seed, offset, _lib.ptr(gamma), save_mean.data_ptr(), save_rstd.data_ptr(),
                  dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), ws.numel(),
                  _lib.stream())
        need = ctx.needs_input_grad
        return (dy, dgamma if need[1] else None, dbeta if need[2] else None,
                None, None, None, None, None, None, None, None, None)


def bn_act(y: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool, p: float, training: bool) -> torch.Tensor:
    """relu? -> Dropout(p) -> BatchNorm1d, fused.  ``bn`` supplies affine parameters and running buffers."""
    _lib.require_cuda(y)
    if not training:
        y = rowmajor(y)
        out = new_padded(y.size(0), y.size(1), y.device, zero_pad=False)
        _lib.call("hgv_bn_act_eval", y.data_ptr(), ld_of(y), y.size(0), y.size(1), int(relu), _lib.ptr(bn.weight),
                  _lib.ptr(bn.bias), float(bn.eps), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                  out.data_ptr(), ld_of(out), _lib.stream())
        return out
    mask, seed, offset = None, 0, 0
    if p > 0.0:
        mask, seed, offset = current_noise().dropout(y.shape, y.device)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    out = _BnAct.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, p, mask, seed, offset, bn.eps,
                       momentum)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return out


# --------------------------------------------------------------------------------------
# reparameterisation and losses
# --------------------------------------------------------------------------------------
class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ml, z_dim, n_samples, eps, seed, offset):
        ml = rowmajor(ml)
        n = ml.size(0)
        z = new_padded(n, z_dim, ml.device)
        eps_bar = torch.empty(n, z_dim, dtype=torch.float32, device=ml.device)
        _lib.call("hgv_reparam_fwd", ml.data_ptr(), ld_of(ml), n, z_dim, n_samples, _lib.ptr(eps), seed, offset,
                  z.data_ptr(), ld_of(z), eps_bar.data_ptr(), _lib.stream())
        ctx.z_dim = z_dim
        ctx.save_for_backward(ml, eps_bar)
        return z

    @staticmethod
    def backward(ctx, gz):
        ml, eps_bar = ctx.saved_tensors
        gz = rowmajor(gz)
        gml = new_padded(ml.size(0), 2 * ctx.z_dim, ml.device)
        _lib.call("hgv_reparam_bwd", gz.data_ptr(), ld_of(gz), ml.data_ptr(), ld_of(ml), ml.size(0), ctx.z_dim,
                  eps_bar.data_ptr(), gml.data_ptr(), ld_of(gml), _lib.stream())
        return gml, None, None, None, None, None


def reparameterize(ml: torch.Tensor, z_dim: int, n_samples: int) -> torch.Tensor:
    """z = mean over n_samples of (mean + exp(log_std) * eps_k); ml = [mean | log_std]."""
    _lib.require_cuda(ml)
    eps, seed, offset = current_noise().normal((n_samples, ml.size(0), z_dim), ml.device)
    return _Reparam.apply(ml, z_dim, n_samples, eps, seed, offset)


class _KL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ml, z_dim):
        ml = rowmajor(ml)
        out = torch.empty((), dtype=torch.float32, device=ml.device)
        ws = _workspace(_lib.load().hgv_loss_ws(), ml.device, "loss")
        _lib.call("hgv_kl_fwd", ml.data_ptr(), ld_of(ml), ml.size(0), z_dim, out.data_ptr(), ws.data_ptr(),
                  ws.numel(), _lib.stream())
        ctx.z_dim = z_dim
        ctx.save_for_backward(ml)
        return out

    @staticmethod
    def backward(ctx, g):
        (ml,) = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        gml = new_padded(ml.size(0), 2 * ctx.z_dim, ml.device)
        _lib.call("hgv_kl_bwd", ml.data_ptr(), ld_of(ml), ml.size(0), ctx.z_dim, g.data_ptr(), gml.data_ptr(),
                  ld_of(gml), 0, _lib.stream())
        return gml, None


def kl_divergence(ml: torch.Tensor, z_dim: int) -> torch.Tensor:
    _lib.require_cuda(ml)
    return _KL.apply(ml, z_dim)


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r, x):
        r, x = rowmajor(r), rowmajor(x)
        out = torch.empty((), dtype=torch.float32, device=r.device)
        ws = _workspace(_lib.load().hgv_loss_ws(), r.device, "loss")
        _lib.call("hgv_mse_fwd", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1),
                  out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.save_for_backward(r, x)
        return out

    @staticmethod
    def backward(ctx, g):
        r, x = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        gr = new_padded(r.size(0), r.size(1), r.device)
        _lib.call("hgv_mse_bwd", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1), g.data_ptr(),
                  gr.data_ptr(), ld_of(gr), _lib.stream())
        return gr, None


def mse_loss(recon: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    _lib.require_cuda(recon, x)
    if recon.shape != x.shape:
        raise _lib.HgvError(f"mse_loss: shapes differ {tuple(recon.shape)} vs {tuple(x.shape)}")
    return _MSE.apply(recon, x)
