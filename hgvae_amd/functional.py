"""Autograd operators over the HIP kernels (forward and hand-written backward).

Every function here launches kernels from ``libhgvae_hip.so`` on the current
stream, the fp32-MFMA projections included.  Nothing falls back to a CPU or eager
re-implementation.
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from . import _lib
from .graph import Graph, as_padded, is_vec4, ld_of, new_padded, rowmajor
from .noise import current_noise

_ws_cache: "dict[tuple, torch.Tensor]" = {}


def _workspace(nbytes: int, device, tag: str) -> torch.Tensor:
    """Per-(device, stream, tag) scratch buffer, grown on demand and reused across calls."""
    key = (device.index, _lib.stream(), tag)        # raw handle of the current stream: no Stream object per call
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 4096), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


# --------------------------------------------------------------------------------------
# neighbour aggregation
# --------------------------------------------------------------------------------------
class LaunchProbe:
    """Optional per-launch timing (events on the launching stream) of the aggregation kernel and of the
    projection products: records are ``(kind, shape, start, end)`` with kind "agg" / "agg_w" (shape
    ``(rows, F, E)``), "nt" (``(M, K, N)``: C[M,N] = A[M,K] W^T) or "tn" (``(M, N, K)``: GW[N,K] = G^T A)."""

    def __init__(self):
        self.records = []

    def add(self, kind, shape):
        s, t = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.append((kind, shape, s, t))
        return s, t

    def durations_ms(self, kind=None):
        torch.cuda.synchronize()
        return [(k, shape, s.elapsed_time(t)) for (k, shape, s, t) in self.records if kind is None or k == kind]


_probe: Optional[LaunchProbe] = None


def set_probe(probe: Optional[LaunchProbe]) -> None:
    global _probe
    _probe = probe


def _agg_launch(csr, n_rows, x, F, row_mode, row_scale, edge_w, nbr_w, add, out):
    if _probe is not None:
        weighted = edge_w is not None or nbr_w is not None
        s, t = _probe.add("agg_w" if weighted else "agg", (n_rows, F, csr.col.numel()))
        s.record()
        _agg_launch_raw(csr, n_rows, x, F, row_mode, row_scale, edge_w, nbr_w, add, out)
        t.record()
        return
    _agg_launch_raw(csr, n_rows, x, F, row_mode, row_scale, edge_w, nbr_w, add, out)


def _agg_launch_raw(csr, n_rows, x, F, row_mode, row_scale, edge_w, nbr_w, add, out):
    # rows far longer than the average (hubs) go to a workgroup-per-row pass, which is launched unless the device
    # has already reported that this CSR has none
    lr = csr.long_rows(n_rows)
    _lib.call("hgv_agg_fwd_long", csr.rowptr.data_ptr(), csr.col.data_ptr(), n_rows, x.data_ptr(), ld_of(x), F, row_mode,
              _lib.ptr(row_scale), _lib.ptr(edge_w), _lib.ptr(nbr_w), _lib.ptr(add),
              ld_of(add) if add is not None else 0, out.data_ptr(), ld_of(out), lr.threshold, lr.rows.data_ptr(),
              lr.count.data_ptr(), lr.cap, 1 if lr.maybe_any() else 0, _lib.stream())


def _agg_affine_launch(csr, n_rows, d, F, col_s, col_t, out):
    """Mean aggregation of x = d * col_s + col_t (never written) over ``csr``: see hgv_agg_fwd_affine."""
    ev = _probe.add("agg", (n_rows, F, csr.col.numel())) if _probe is not None else None
    if ev:
        ev[0].record()
    lr = csr.long_rows(n_rows)
    _lib.call("hgv_agg_fwd_affine", csr.rowptr.data_ptr(), csr.col.data_ptr(), n_rows, d.data_ptr(), ld_of(d), F,
              col_s.data_ptr(), col_t.data_ptr(), out.data_ptr(), ld_of(out), lr.threshold, lr.rows.data_ptr(),
              lr.count.data_ptr(), lr.cap, 1 if lr.maybe_any() else 0, _lib.stream())
    if ev:
        ev[1].record()


def _agg_ind_launch(csr, n_rows, d, F, out, ind_col):
    """Mean aggregation of d over ``csr`` with the has-neighbours indicator in pad column ``ind_col`` of ``out``."""
    ev = _probe.add("agg", (n_rows, F, csr.col.numel())) if _probe is not None else None
    if ev:
        ev[0].record()
    lr = csr.long_rows(n_rows)
    _lib.call("hgv_agg_fwd_ind", csr.rowptr.data_ptr(), csr.col.data_ptr(), n_rows, d.data_ptr(), ld_of(d), F, out.data_ptr(),
              ld_of(out), ind_col, lr.threshold, lr.rows.data_ptr(), lr.count.data_ptr(), lr.cap, 1 if lr.maybe_any() else 0,
              _lib.stream())
    if ev:
        ev[1].record()


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
    x = plain(x)
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
# levels whose normalisation is deferred to their consumers
# --------------------------------------------------------------------------------------
# A level is  y = projection(...) -> relu? -> dropout -> batch norm  and its output feeds further projections / mean
# aggregations.  Inside :func:`deferred_levels` (the loss functions of the train loops enter it) the pass that takes the
# column statistics overwrites y with d = dropout(relu(y)), and the level hands out a :class:`Deferred` - d plus the two
# vectors of  out = d * s + t  - whose consumers fold s / t into their weights (``_image_affine``), apply them to finished
# aggregation rows, and correct their weight gradients (``_wgrad``): the normalised matrix is never written or read (one
# read and one write of the level's matrix instead of two reads and a write).  HGVAE_DEFER=0 switches it off (A/B runs).
_DEFER_ENABLED = os.environ.get("HGVAE_DEFER", "1") != "0"
DEFER_MIN_P = 0.2        # folding s / t into the weights turns  (d - mean) * rstd  into a difference of two sums; dropout
                         # with probability p bounds |mean| / std of d by sqrt((1 - p) / p), so nothing cancels for p >= 0.2
DEFER_MIN_ROWS = int(os.environ.get("HGVAE_DEFER_MIN_ROWS", "65536"))
# Below this many rows a level is bound by its launches, not by its bytes, and deferral adds two small ones per consumer
# (the scaled image, the weight-gradient finish): measured on one MI355X, captured steps of 4 416 rows 0.73 -> 0.77 ms and
# 70 825 rows 1.98 -> 1.96 ms, eager steps of 160 984 rows 4.95 -> 4.69 ms, of 1 283 529 rows 36.1 -> 35.3 ms.
_defer = False
# The backward pass of a deferred level takes its two column sums from the consumers' weight-gradient products when it
# can (hgv_bn_act_bwd_pre_alg); HGVAE_ALG_SUMS=0 always takes them from the matrices (A/B runs).
_alg_sums = os.environ.get("HGVAE_ALG_SUMS", "1") != "0"


def set_deferred_levels(enabled: bool) -> None:
    global _DEFER_ENABLED
    _DEFER_ENABLED = bool(enabled)


@contextlib.contextmanager
def deferred_levels(enabled: bool = True):
    global _defer
    prev, _defer = _defer, bool(enabled) and _DEFER_ENABLED and _projection == "f16x3"
    try:
        yield
    finally:
        _defer = prev


class _GradParts:
    """What the consumers of one deferred level leave behind for its backward pass.  A consumer that took the level as
    (d, s, t) sends back g_p = G_p W_p and has, for its own weight gradient, formed raw_p = G_p^T d and the column sums
    of G_p; with those of ALL consumers the level's backward needs no pass over the matrices to get the column sums of g
    and of g * xhat (hgv_bn_act_bwd_pre_alg).  ``uses`` counts the consumers that registered in the forward pass;
    ``tainted``: some consumer read the level another way (the written-out matrix, frozen weights, an aggregate of the
    normalised operand), so the sums must be taken from the matrices.  ``handles``: operators that took the level as
    (d, s, t) in the forward pass; ``level``: what the level's backward needs (set by the level); ``fused``: the level's
    backward as already applied by its only (or last) consumer's grad-input product (:func:`_grad_input`); ``pending``:
    the shares of the incoming gradient that earlier consumers' products have written; ``shared``: one handle went to two
    operators (autograd adds their gradients before the level sees them, so no product may stand in for the level)."""
    __slots__ = ("parts", "uses", "tainted", "handles", "level", "fused", "pending", "shared")

    def __init__(self):
        self.parts, self.uses, self.tainted = [], 0, False
        self.handles, self.level, self.fused, self.pending, self.shared = 0, None, None, [], False

    def add(self, w, raw, csum) -> None:
        self.parts.append((w, raw, csum))

    def complete(self) -> bool:
        return not self.tainted and 0 < self.uses == len(self.parts) <= 4


class Deferred:
    """Output of a level, kept as d = dropout(relu(y)) [N, H] and the column vectors s, t of  out = d * s + t.

    ``d`` is connected to the autograd graph; by convention the gradient that flows back into it is the gradient with
    respect to ``out`` (the consumers in this module compute exactly that).  ``tensor()`` writes ``out`` for a consumer
    that needs the matrix."""
    __slots__ = ("d", "s", "t", "mean", "rstd", "gamma", "beta", "parts", "_out", "taken")

    def __init__(self, d, s, t, mean, rstd, gamma, beta, parts=None):
        self.d, self.s, self.t, self.mean, self.rstd, self.gamma, self.beta = d, s, t, mean, rstd, gamma, beta
        self.parts = parts
        self._out = None
        self.taken = 0               # operators that took this handle as (d, s, t)

    @property
    def shape(self):
        return self.d.shape

    @property
    def device(self):
        return self.d.device

    @property
    def requires_grad(self):
        return self.d.requires_grad

    def size(self, dim=None):
        return self.d.size() if dim is None else self.d.size(dim)

    def dim(self):
        return self.d.dim()

    def record_stream(self, stream) -> None:
        for t in (self.d, self.s, self.mean, self.rstd):          # s and t share one buffer
            t.record_stream(stream)

    def tensor(self) -> torch.Tensor:
        if self.parts is not None:
            self.parts.tainted = True
        if self._out is None:
            self._out = _Materialize.apply(self.d, self.mean, self.rstd, self.gamma, self.beta)
        return self._out


class _Materialize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d, mean, rstd, gamma, beta):
        out = new_padded(d.size(0), d.size(1), d.device, zero_pad=not is_vec4(d))
        _lib.call("hgv_bn_apply", d.data_ptr(), ld_of(d), d.size(0), d.size(1), _lib.ptr(gamma), _lib.ptr(beta),
                  mean.data_ptr(), rstd.data_ptr(), out.data_ptr(), ld_of(out), _lib.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None, None          # the gradient with respect to `out` is what flows into d


def plain(x):
    """``x`` as an ordinary tensor (writes the normalised matrix of a :class:`Deferred`)."""
    return x.tensor() if isinstance(x, Deferred) else x


def _affine_ok(n: int, k1: int, k2: int) -> bool:
    """A consumer with an n x (k1 + k2) weight can fold column scales into its image (the scaled-fp16 kernels take it)."""
    return _projection == "f16x3" and _split_image_floats(n, k1, k2, "f16x3") > 0


def _image_affine(wa, aff_a, wb, aff_b, bias, ind_a: bool = False):
    """(scaled-fp16 image of [wa diag(s_a) | wb diag(s_b)], bias + wa t_a + wb t_b); ``aff`` = (s, t) or None.  With
    ``ind_a`` operand a carries one more column than ``wa``, a {0, 1} indicator per row, and ``wa t_a`` becomes that
    column's weight instead of a part of the bias."""
    n, k1 = wa.size(0), wa.size(1) + (1 if ind_a else 0)
    k2 = wb.size(1) if wb is not None else 0
    buf = torch.empty(_split_image_floats(n, k1, k2, "f16x3"), dtype=torch.float32, device=wa.device)
    bias_out = torch.empty(n, dtype=torch.float32, device=wa.device)
    sa, ta = (aff_a[0], aff_a[1]) if aff_a is not None else (None, None)
    sb, tb = (aff_b[0], aff_b[1]) if aff_b is not None else (None, None)
    _lib.call("hgv_pack16h_weights_affine", wa.data_ptr(), wa.stride(0), wa.stride(1), k1, _lib.ptr(sa), _lib.ptr(ta),
              1 if ind_a else 0, _lib.ptr(wb), wb.stride(0) if wb is not None else 0, wb.stride(1) if wb is not None else 0, k2,
              _lib.ptr(sb), _lib.ptr(tb), n, _lib.ptr(bias), bias_out.data_ptr(), buf.data_ptr(), _lib.stream())
    return _Image("f16x3", buf), bias_out


def _taint(aff) -> None:
    if aff is not None and len(aff) > 2 and aff[2] is not None:
        aff[2].tainted = True


def _contribute(aff, w, raw, csum) -> None:
    """Leave (W, G^T d, column sums of G) of a product with a deferred operand for that level's backward pass."""
    if aff is not None and len(aff) > 2 and aff[2] is not None:
        aff[2].add(w, raw, csum)


def _wgrad_finish(jobs, colsum, accumulate: bool) -> None:
    """dst (+)= raw * s + c t^T for up to three jobs in one launch.  A job is ``(dst, raw, (s, t) or None)`` with
    c = ``colsum``, or ``(dst, raw, (s, t), c)`` with its own vector c (any stride)."""
    nj = len(jobs)
    vp_t, i64_t, i32_t = _ctypes_arrays(nj)
    D, R, S, T, CS = vp_t(), vp_t(), vp_t(), vp_t(), vp_t()
    ldd, ldr, css = i64_t(), i64_t(), i64_t()
    N, K = i32_t(), i32_t()
    for q, job in enumerate(jobs):
        d, r, aff = job[0], job[1], job[2]
        c = job[3] if len(job) > 3 else colsum
        if tuple(d.shape) != tuple(r.shape) or (d.size(1) > 1 and (d.stride(1) != 1 or r.stride(1) != 1)):
            raise _lib.HgvError(f"weight-gradient finish: {tuple(d.shape)} / {d.stride()} from {tuple(r.shape)} / {r.stride()}")
        D[q], R[q], ldd[q], ldr[q], N[q], K[q] = d.data_ptr(), r.data_ptr(), max(d.stride(0), d.size(1)), max(r.stride(0), r.size(1)), r.size(0), r.size(1)
        css[q] = 1
        if aff is not None:
            S[q], T[q], CS[q], css[q] = aff[0].data_ptr(), aff[1].data_ptr(), c.data_ptr(), (c.stride(0) if c.numel() > 1 else 1)
    _lib.call("hgv_wgrad_finish", nj, D, ldd, R, ldr, N, K, S, T, CS, css, 1 if accumulate else 0, _lib.stream())


def _wgrad(g, ops, bias, want_bias: bool):
    """Weight (and bias) gradients of  y = sum_i x_i @ w_i^T + bias  for one or two operands.

    ``ops`` = [(x, w, aff, want)]: ``x`` the stored operand, ``aff`` = (s, t) when the operand the product really saw was
    x * s + t column by column (then G^T (x s + t) = (G^T x) diag(s) + colsum(G) t^T is formed from the raw product).
    Returns ([gw_i ...], gb) for autograd with None where a gradient was added straight into ``.grad`` or not wanted."""
    live = [(x, w, aff) for (x, w, aff, want) in ops if want]
    for (_, _, aff, want) in ops:
        if aff is not None and not want:
            _taint(aff)                   # no product with this operand is formed: its level takes its sums the long way
    if not live:
        if want_bias:
            _, gb = _gemm_tn(g, ops[0][0], True)
            return [None] * len(ops), gb
        return [None] * len(ops), None
    if all(aff is None for (_, _, aff) in live):
        if len(live) == 2:
            ga, gb_, gbias = _weight_grad2(g, live[0][0], live[0][1], live[1][0], live[1][1], bias, want_bias)
            res = [ga, gb_]
        else:
            ga, gbias = _weight_grad(g, live[0][0], live[0][1], bias, want_bias)
            res = [ga]
    else:
        if len(live) == 2:
            ra, rb, colsum = _gemm_tn2(g, live[0][0], live[1][0], True)
            raws = [ra, rb]
        else:
            ra, colsum = _gemm_tn(g, live[0][0], True)
            raws = [ra]
        for r, (_, w, aff) in zip(raws, live):
            _contribute(aff, w, r, colsum)
        bufs = [_grad_buffer(w) for (_, w, _) in live]
        bbuf = _grad_buffer(bias) if want_bias else None
        direct = all(b is not None for b in bufs) and (not want_bias or bbuf is not None)
        dsts = bufs if direct else [torch.empty(w.shape, dtype=torch.float32, device=g.device) for (_, w, _) in live]
        jobs = [(d, r, aff) for d, r, (_, _, aff) in zip(dsts, raws, live)]
        gbias = None
        if want_bias:
            gbias = bbuf if direct else torch.empty_like(colsum)
            jobs.append((gbias.view(-1, 1), colsum.view(-1, 1), None))
        _wgrad_finish(jobs, colsum, accumulate=direct)
        res = [None] * len(live) if direct else dsts
        gbias = None if direct else gbias
    out, it = [], iter(res)
    for (_, _, _, want) in ops:
        out.append(next(it) if want else None)
    return out, gbias


class _BnStats(torch.autograd.Function):
    """relu? -> dropout -> column statistics of a level, in place: the buffer of ``y`` holds d = dropout(relu(y))
    afterwards (see :class:`Deferred`).  Returns one or two handles on it and the vectors [s | t], mean, rstd; the backward
    pass is that of relu? -> dropout -> batch norm, read off d.  The caller must own ``y`` (nothing else may need the
    pre-activation: the projections of this module keep only their inputs)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, p, mask, seed, offset, ctr, eps, momentum, tracked, fork,
                parts=None, pre=False):
        ctx.parts = parts
        n, h = y.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        hp = (h + 3) // 4 * 4
        st = torch.empty(2, hp, dtype=torch.float32, device=y.device)
        save_mean = torch.empty(h, dtype=torch.float32, device=y.device)
        save_rstd = torch.empty(h, dtype=torch.float32, device=y.device)
        d = y.detach()                 # same storage: overwritten behind autograd's back (y has no other reader)
        if isinstance(pre, torch.Tensor):      # ... and left its partial column statistics: merge them
            _lib.call("hgv_bn_stats_merge", pre.data_ptr(), pre.size(0), 16, n, h, _lib.ptr(gamma), _lib.ptr(beta), float(eps),
                      _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), _lib.ptr(tracked), save_mean.data_ptr(),
                      save_rstd.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), _lib.stream())
        elif pre:                      # the producing projection stored d already: statistics only
            _lib.call("hgv_bn_stats_pre", d.data_ptr(), ld_of(d), n, h, _lib.ptr(gamma), _lib.ptr(beta), float(eps),
                      _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), _lib.ptr(tracked), save_mean.data_ptr(),
                      save_rstd.data_ptr(), st[0].data_ptr(), st[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
        else:
            _lib.call("hgv_bn_stats_act", d.data_ptr(), ld_of(d), n, h, int(relu), float(p), _lib.ptr(mask), seed, offset, ctr,
                      _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum),
                      _lib.ptr(tracked), save_mean.data_ptr(), save_rstd.data_ptr(), st[0].data_ptr(), st[1].data_ptr(),
                      ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.cfg = (bool(relu), float(p))
        if parts is not None:          # (a forked level whose second handle nobody takes has one consumer too)
            parts.level = (d, gamma, beta, save_mean, save_rstd, bool(relu), float(p), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        ctx.save_for_backward(d, gamma, save_mean, save_rstd, beta)
        ctx.mark_non_differentiable(st, save_mean, save_rstd)
        ctx.set_materialize_grads(False)
        if fork:
            return y.view_as(y), y.view_as(y), st, save_mean, save_rstd
        return y.view_as(y), st, save_mean, save_rstd

    @staticmethod
    def backward(ctx, *grads):
        d, gamma, save_mean, save_rstd, beta = ctx.saved_tensors
        relu, p = ctx.cfg
        g, g2 = (grads[0], grads[1]) if len(grads) == 5 else (grads[0], None)
        if g is None:
            g, g2 = g2, None
        if g is None:
            return (None,) * 17
        n, h = d.shape
        need = ctx.needs_input_grad
        parts = ctx.parts
        if parts is not None and parts.fused is not None:
            # the only consumer's grad-input product has stored dy itself (and left dgamma / dbeta): pass it on
            dy, dgamma, dbeta = parts.fused
            waiting = len(parts.pending)
            parts.fused = None
            parts.parts.clear()
            parts.pending.clear()
            # the other gradient of a level with two consumers is the share the product has already added
            if waiting == 1 and g2 is not None and g2.data_ptr() == dy.data_ptr():
                g, g2 = g2, None
            elif waiting == 1 and g2 is not None and g.data_ptr() == dy.data_ptr():
                g2 = None
            if g2 is not None or g.data_ptr() != dy.data_ptr() or g.shape != dy.shape:
                raise _lib.HgvError("a level's backward was applied by a product whose result is not the level's incoming gradient")
            return (g, dgamma if need[1] else None, dbeta if need[2] else None) + (None,) * 14
        g = rowmajor(g)
        g2 = rowmajor(g2) if g2 is not None else None
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), d.device, "bn")
        vec = is_vec4(d) and is_vec4(g) and (g2 is None or is_vec4(g2))
        dy = new_padded(n, h, d.device, zero_pad=not vec)
        gbuf, bbuf = (_grad_buffer(gamma) if need[1] else None), (_grad_buffer(beta) if need[2] else None)
        direct = gbuf is not None and bbuf is not None
        dgamma = gbuf if direct else torch.empty(h, dtype=torch.float32, device=d.device)
        dbeta = bbuf if direct else torch.empty(h, dtype=torch.float32, device=d.device)
        if _alg_sums and parts is not None and parts.complete() and all(w.size(1) == h for (w, _, _) in parts.parts):
            # every consumer was a projection on (d, s, t): the column sums come from their weight-gradient products
            nparts = len(parts.parts)
            vp_t, i64_t, i32_t = _ctypes_arrays(nparts)
            W, R, CS = vp_t(), vp_t(), vp_t()
            ldw, ldr, css = i64_t(), i64_t(), i64_t()
            nn_ = i32_t()
            for q, (w, raw, cs) in enumerate(parts.parts):
                W[q], R[q], CS[q] = w.data_ptr(), raw.data_ptr(), cs.data_ptr()
                ldw[q], ldr[q], css[q], nn_[q] = w.stride(0), raw.stride(0), (cs.stride(0) if cs.numel() > 1 else 1), w.size(0)
            _lib.call("hgv_bn_act_bwd_pre_alg", g.data_ptr(), ld_of(g), _lib.ptr(g2), ld_of(g2) if g2 is not None else 0,
                      d.data_ptr(), ld_of(d), n, h, int(relu), float(p), _lib.ptr(gamma), save_mean.data_ptr(),
                      save_rstd.data_ptr(), nparts, W, ldw, R, ldr, CS, css, nn_, dy.data_ptr(), ld_of(dy), dgamma.data_ptr(),
                      dbeta.data_ptr(), 1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
        else:
            _lib.call("hgv_bn_act_bwd_pre", g.data_ptr(), ld_of(g), _lib.ptr(g2), ld_of(g2) if g2 is not None else 0, d.data_ptr(),
                      ld_of(d), n, h, int(relu), float(p), _lib.ptr(gamma), save_mean.data_ptr(), save_rstd.data_ptr(),
                      dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(), 1 if direct else 0, ws.data_ptr(),
                      ws.numel(), _lib.stream())
        if parts is not None:
            parts.parts.clear()
            parts.pending.clear()
        if direct:
            dgamma = dbeta = None
        return (dy, dgamma if need[1] else None, dbeta if need[2] else None) + (None,) * 14


_defer_pair = os.environ.get("HGVAE_DEFER_PAIR", "1") != "0"            # 0: the shared first level is written out (A/B)


class _BnStatsPair(torch.autograd.Function):
    """Two dropout draws of one pre-activation as two deferred levels, over one read of it: the buffer of ``y`` holds the
    first draw's d afterwards, a new buffer the second's.  Returns two handles per draw and [2][s | t], mean [2, H],
    rstd [2, H]; the backward pass handles both draws in one pass and returns the sum of the two input gradients."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, p, seed, offset, offset2, ctr, eps, momentum, tracked,
                parts_a, parts_b, pre=None):
        n, h = y.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        lib = _lib.load()
        hp = (h + 3) // 4 * 4
        st = torch.empty(2, 2, hp, dtype=torch.float32, device=y.device)            # [s | t][draw][hp]
        stats = torch.empty(2, 2, h, dtype=torch.float32, device=y.device)          # [mean | rstd][draw][h]
        d1 = y.detach()
        if pre is not None:
            # the producing projection has stored both draws (the first in y's place) and left their partial column statistics:
            # the two merges update the running statistics in the order of the draws
            partial, d2 = pre
            for k in range(2):
                _lib.call("hgv_bn_stats_merge", partial[k].data_ptr(), partial.size(1), 16, n, h, _lib.ptr(gamma), _lib.ptr(beta),
                          float(eps), _lib.ptr(running_mean), _lib.ptr(running_var), float(momentum), _lib.ptr(tracked),
                          stats[0, k].data_ptr(), stats[1, k].data_ptr(), st[0, k].data_ptr(), st[1, k].data_ptr(), _lib.stream())
        else:
            ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
            d2 = torch.empty_strided((n, h), (ld_of(y), 1), dtype=torch.float32, device=y.device)
            _lib.call("hgv_bn_stats_act2", d1.data_ptr(), d2.data_ptr(), ld_of(d1), n, h, int(relu), float(p), seed, offset, offset2,
                      ctr, _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(running_mean), _lib.ptr(running_var),
                      float(momentum), _lib.ptr(tracked), stats[0].data_ptr(), stats[1].data_ptr(), st[0].data_ptr(),
                      st[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.cfg = (bool(relu), float(p))
        ctx.parts = (parts_a, parts_b)
        ctx.save_for_backward(d1, d2, gamma, stats, beta)
        ctx.mark_non_differentiable(st, stats)
        ctx.set_materialize_grads(False)
        return y.view_as(y), y.view_as(y), d2, d2.view_as(d2), st, stats

    @staticmethod
    def backward(ctx, g, g2, h, h2, _st, _stats):
        d1, d2, gamma, stats, beta = ctx.saved_tensors
        relu, p = ctx.cfg
        if g is None:
            g, g2 = g2, None
        if h is None:
            h, h2 = h2, None
        if g is None and h is None:
            return (None,) * 17
        n, hh = d1.shape
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, hh), d1.device, "bn")
        need = ctx.needs_input_grad
        gbuf, bbuf = (_grad_buffer(gamma) if need[1] else None), (_grad_buffer(beta) if need[2] else None)
        direct = gbuf is not None and bbuf is not None
        dgamma = gbuf if direct else torch.empty(hh, dtype=torch.float32, device=d1.device)
        dbeta = bbuf if direct else torch.empty(hh, dtype=torch.float32, device=d1.device)
        dy = new_padded(n, hh, d1.device, zero_pad=False)
        grads = [as_padded(rowmajor(t)) if t is not None else None for t in (g, g2, h, h2)]
        pa, pb = ctx.parts

        def part_arrays(lists):
            flat = [q for lst in lists for q in lst]
            vp_t, i64_t, i32_t = _ctypes_arrays(max(len(flat), 1))
            W, R, CS = vp_t(), vp_t(), vp_t()
            ldw, ldr, css = i64_t(), i64_t(), i64_t()
            nn_ = i32_t()
            for q, (w, raw, cs) in enumerate(flat):
                W[q], R[q], CS[q] = w.data_ptr(), raw.data_ptr(), cs.data_ptr()
                ldw[q], ldr[q], css[q], nn_[q] = w.stride(0), raw.stride(0), (cs.stride(0) if cs.numel() > 1 else 1), w.size(0)
            return W, ldw, R, ldr, CS, css, nn_

        def usable(parts):
            return _alg_sums and parts is not None and parts.complete() and all(w.size(1) == hh for (w, _, _) in parts.parts)

        if grads[0] is None or grads[2] is None:
            # only one of the two draws reached the loss: the single-draw backward on that draw
            first = grads[0] is not None
            a, a2 = (grads[0], grads[1]) if first else (grads[2], grads[3])
            k, d, parts = (0, d1, pa) if first else (1, d2, pb)
            if usable(parts):
                W, ldw, R, ldr, CS, css, nn_ = part_arrays([parts.parts])
                _lib.call("hgv_bn_act_bwd_pre_alg", a.data_ptr(), ld_of(a), _lib.ptr(a2), ld_of(a2) if a2 is not None else 0,
                          d.data_ptr(), ld_of(d), n, hh, int(relu), float(p), _lib.ptr(gamma), stats[0, k].data_ptr(),
                          stats[1, k].data_ptr(), len(parts.parts), W, ldw, R, ldr, CS, css, nn_, dy.data_ptr(), ld_of(dy),
                          dgamma.data_ptr(), dbeta.data_ptr(), 1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
            else:
                _lib.call("hgv_bn_act_bwd_pre", a.data_ptr(), ld_of(a), _lib.ptr(a2), ld_of(a2) if a2 is not None else 0,
                          d.data_ptr(), ld_of(d), n, hh, int(relu), float(p), _lib.ptr(gamma), stats[0, k].data_ptr(),
                          stats[1, k].data_ptr(), dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(),
                          1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
        else:
            a, a2, b, b2 = grads
            alg = usable(pa) and usable(pb)
            n0, n1 = (len(pa.parts), len(pb.parts)) if alg else (0, 0)
            W, ldw, R, ldr, CS, css, nn_ = part_arrays([pa.parts, pb.parts] if alg else [])
            _lib.call("hgv_bn_act_bwd2_pre", a.data_ptr(), ld_of(a), _lib.ptr(a2), ld_of(a2) if a2 is not None else 0,
                      b.data_ptr(), ld_of(b), _lib.ptr(b2), ld_of(b2) if b2 is not None else 0, d1.data_ptr(), d2.data_ptr(),
                      ld_of(d1), n, hh, int(relu), float(p), _lib.ptr(gamma), stats[0].data_ptr(), stats[1].data_ptr(), n0, n1,
                      W, ldw, R, ldr, CS, css, nn_, dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(),
                      1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
        for parts in (pa, pb):
            if parts is not None:
                parts.parts.clear()
        if direct:
            dgamma = dbeta = None
        return (dy, dgamma if need[1] else None, dbeta if need[2] else None) + (None,) * 14


_pair_in_product = os.environ.get("HGVAE_PAIR_IN_PRODUCT", "1") != "0"    # 0: the two-draw level reads the pre-activation in a pass of its own (A/B)


def bn_level_pair(y, bn: torch.nn.BatchNorm1d, relu: bool, p: float, rows: Optional[int] = None):
    """Training-mode relu? -> Dropout(p) -> BatchNorm1d for two successive dropout draws of the same fresh matrix ``y``
    (the two encode passes of a train step over the shared first convolution), each result as a pair of handles for its
    two consumers: ``((a, a'), (b, b'))`` - :class:`Deferred` levels inside :func:`deferred_levels`, else tensors
    (:func:`bn_act_pair`) - or None when neither form applies (explicit noise tapes, p = 0, unaligned rows, no gradient)
    and the caller makes two ordinary calls.

    ``y`` may be a callable that builds the matrix with exactly one projection operator of this module (and ``rows`` its
    row count): the product may then store both draws and their column sums itself, and the result is ``(pair or None,
    y)``."""
    global _act_request, _act_done, _act_ready
    made = callable(y)
    pre = None
    if made:
        make = y
        ask = (_defer and _defer_pair and _pair_in_product and _nt_act and _nt_stats and torch.is_grad_enabled() and p >= DEFER_MIN_P
               and (rows is None or rows >= max(2, DEFER_MIN_ROWS)) and noise_is_counter_based() and not on_branch_stream())
        _act_done = None
        _act_request = (relu, p, "pair") if ask else None
        try:
            y = plain(make())
        finally:
            _act_request = None
            _act_ready = None
        done, _act_done = _act_done, None
        if done is not None:
            if done[:5] != (y.data_ptr(), y.size(0), y.size(1), relu, p) or len(done) < 7:
                raise _lib.HgvError("a projection stored two activated draws that are not this level's input")
            pre = done
    if _defer and _defer_pair and torch.is_grad_enabled() and p >= DEFER_MIN_P and y.dim() == 2 \
            and y.size(0) >= max(2, DEFER_MIN_ROWS) and y.requires_grad and y.grad_fn is not None and is_vec4(y) \
            and y.size(1) <= 512 and noise_is_counter_based() and not on_branch_stream():
        if pre is not None:
            out2, seed, offset, offset2, ctr = pre[6]
            stored = (pre[5], out2)
        else:
            noise = current_noise()
            _, seed, offset, ctr = noise.dropout(y.shape, y.device)
            _, _, offset2, _ = noise.dropout(y.shape, y.device)
            stored = None
        momentum = bn.momentum if bn.momentum is not None else 0.1
        pa, pb = _GradParts(), _GradParts()
        a, a2, b, b2, st, stats = _BnStatsPair.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, p, seed, offset,
                                                     offset2, ctr, bn.eps, momentum, bn.num_batches_tracked, pa, pb, stored)

        def wrap(h, k, parts):
            return Deferred(h, st[0, k], st[1, k], stats[0, k], stats[1, k], bn.weight, bn.bias, parts)
        pair = (wrap(a, 0, pa), wrap(a2, 0, pa)), (wrap(b, 1, pb), wrap(b2, 1, pb))
        return (pair, y) if made else pair
    if pre is not None:
        raise _lib.HgvError("a projection stored two activated draws for a level that cannot take them")
    pair = bn_act_pair(y, bn, relu, p)
    return (pair, y) if made else pair


def bn_level(make_y, bn: torch.nn.BatchNorm1d, relu: bool, p: float, training: bool, fork: bool = False, act_ok: bool = False,
             rows: Optional[int] = None):
    """One level: ``make_y()`` -> relu? -> Dropout(p) -> BatchNorm1d.  ``make_y`` builds the pre-activation (a fresh
    matrix that nothing else reads); the result is a tensor, or inside :func:`deferred_levels` a :class:`Deferred`
    (a pair of either with ``fork``).  ``act_ok``: ``make_y`` is exactly one projection operator of this module (SAGEConv,
    dual / concatenated linear) and nothing else, so its product may store relu -> dropout itself.  ``rows``: the row count
    of the level when the caller knows it (lets small batches skip the set-up)."""
    global _act_request, _act_done
    want = _defer and training and p >= DEFER_MIN_P and (rows is None or rows >= max(2, DEFER_MIN_ROWS)) and torch.is_grad_enabled()
    if not want:                       # small batches and evaluation: the fused epilogue, nothing to set up
        return bn_act(plain(make_y()), bn, relu, p, training, fork=fork)
    _act_done = None
    _act_request = (relu, p) if (want and act_ok and _nt_act) else None
    global _act_ready
    try:
        y = plain(make_y())
    finally:
        _act_request = None
        _act_ready = None
    done, _act_done = _act_done, None
    pre = done is not None and done[:5] == (y.data_ptr(), y.size(0), y.size(1), relu, p)
    if done is not None and not pre:
        raise _lib.HgvError("a projection stored an activated level that is not this level's input")
    stored = pre
    if pre:
        pre = done[5] if done[5] is not None else True           # the partial statistics, or "stored, not summed"
    if stored or (want and y.dim() == 2 and y.size(0) >= max(2, DEFER_MIN_ROWS) and y.requires_grad and y.grad_fn is not None
               and is_vec4(y)):
        mask, seed, offset, ctr = (None, 0, 0, None) if stored else current_noise().dropout(y.shape, y.device)
        momentum = bn.momentum if bn.momentum is not None else 0.1
        rm, rv, tracked = bn.running_mean, bn.running_var, bn.num_batches_tracked
        if on_branch_stream():         # see bn_act: the trunk may be updating the running statistics right now
            stand_in = torch.zeros(2, y.size(1), dtype=torch.float32, device=y.device)
            _branch_updates.append((bn, stand_in, momentum))
            rm, rv, tracked = stand_in[0], stand_in[1], None
        parts = _GradParts()
        outs = _BnStats.apply(y, bn.weight, bn.bias, rm, rv, relu, p, mask, seed, offset, ctr, bn.eps, momentum, tracked, fork,
                              parts, pre)
        st, mean, rstd = outs[-3], outs[-2], outs[-1]
        made = [Deferred(h, st[0], st[1], mean, rstd, bn.weight, bn.bias, parts) for h in outs[:-3]]
        return (made[0], made[1]) if fork else made[0]
    return bn_act(y, bn, relu, p, training, fork=fork)


# --------------------------------------------------------------------------------------
# dense projection pair:  y = xa @ Wa^T + xb @ Wb^T + bias   (xb / Wb optional)
# --------------------------------------------------------------------------------------
MAX_GEMM_N = 288


def _pack(*specs):
    """Packed images for the NT kernel, built by ONE launch at every use: ``specs`` are up to four
    ``(weight, transposed)`` pairs (None entries are skipped and returned as None); ``transposed``
    packs w^T, the operand of the grad-input product.

    Images are not cached across calls: ``Tensor._version`` is the only cheap change signal and it
    misses real updates (``torch.optim.Adam(fused=True)`` and writes through ``.data`` leave it
    untouched), and a stale image silently trains against old weights.  An operator packs what its
    forward AND backward need when it runs forward (the weights backward must use are the forward
    ones anyway) and keeps the transposed images until backward.
    """
    import ctypes as C
    live = [(i, w, t) for i, sp in enumerate(specs) if sp is not None for (w, t) in [sp]]
    out = [None] * len(specs)
    if not live:
        return out
    lib = _lib.load()
    dims = [((w.size(1), w.size(0)) if t else (w.size(0), w.size(1))) for (_, w, t) in live]
    sizes = [lib.hgv_pack_bytes(n, k) // 4 for (n, k) in dims]
    buf = torch.empty(sum(sizes), dtype=torch.float32, device=live[0][1].device)
    nj = len(live)
    Wp, Op = (C.c_void_p * nj)(), (C.c_void_p * nj)()
    sn, sk = (C.c_int64 * nj)(), (C.c_int64 * nj)()
    N, K = (C.c_int * nj)(), (C.c_int * nj)()
    off = 0
    for q, ((i, w, t), (n, k), sz) in enumerate(zip(live, dims, sizes)):
        piece = buf[off:off + sz]
        off += sz
        out[i] = piece
        Wp[q], Op[q] = w.data_ptr(), piece.data_ptr()
        sn[q], sk[q] = (w.stride(1), w.stride(0)) if t else (w.stride(0), w.stride(1))
        N[q], K[q] = n, k
    _lib.call("hgv_pack_weights", nj, Wp, sn, sk, N, K, Op, _lib.stream())
    return out


# Projection arithmetic.  "f16x3" (default): every operand is scaled by a power of two (per row of A, per row of W;
# per workgroup in the weight gradients) and split into two fp16 pieces, three piece products on the fp16 matrix
# cores, fp32 accumulation, exact rescaling: dropped terms <= 3 * 2^-22 per product, measured error at or below the
# other two kinds.  "bf16x6": three bf16 pieces per operand, six piece products (no scaling, twice the matrix-core
# work).  "fp32" keeps v_mfma_f32_16x16x4_f32.  Shapes whose split weight image does not fit in LDS use the fp32
# kernels either way.  "auto": "f16x3" guarded per launch - a pass over the row-count-sized operands looks for a row that
# holds a non-zero entry more than 2^16 below its largest one (outside that range the two fp16 pieces keep an absolute,
# not a relative, accuracy); both arithmetics are enqueued behind the device-side flag and exactly one of them runs, with
# no host round trip.  The test is on the operand alone, so it is conservative: dense normalised activations almost
# always contain some entry that close to zero and then run as "bf16x6"; it is meant for callers who need the fp32-level
# bound to hold whatever the weights are, at the price of one more read of the operands and a second image.
_DEFAULT_PROJECTION = os.environ.get("HGVAE_PROJECTION", "f16x3")
_projection = _DEFAULT_PROJECTION
_SPLIT_KINDS = {"bf16x6": ("hgv_gemm_nt16_supported", "hgv_pack16_bytes", "hgv_pack16_weights_multi", "hgv_gemm_nt16"),
                "f16x3": ("hgv_gemm_nt16h_supported", "hgv_pack16h_bytes", "hgv_pack16h_weights_multi", "hgv_gemm_nt16h")}


def default_projection() -> str:
    return _DEFAULT_PROJECTION


def set_projection(kind: str) -> None:
    global _projection
    if kind not in ("bf16x6", "f16x3", "fp32", "auto"):
        raise ValueError("projection arithmetic is 'bf16x6', 'f16x3', 'auto' or 'fp32'")
    _projection = kind


_array_types: dict = {}
_split_sizes: dict = {}


def _split_kind() -> Optional[str]:
    """The split arithmetic single images are packed for now ('auto' packs 'f16x3' and 'bf16x6' side by side)."""
    return _projection if _projection in _SPLIT_KINDS else ("f16x3" if _projection == "auto" else None)


def _split_image_floats(n: int, k1: int, k2: int, kind: Optional[str] = None) -> int:
    """Size (in floats) of the split image of an n x (k1 + k2) weight, 0 when the NT kernel does not take the shape."""
    kind = kind or _split_kind()
    key = (kind, n, k1, k2)
    v = _split_sizes.get(key)
    if v is None:
        lib = _lib.load()
        supported, nbytes, _, _ = _SPLIT_KINDS[kind]
        v = _split_sizes[key] = getattr(lib, nbytes)(n, k1, k2) // 4 if getattr(lib, supported)(n, k1, k2) else 0
    return v



def _ctypes_arrays(n: int):
    """(void*[n], int64[n], int[n]) ctypes array types; building such a type costs microseconds, so they are made once."""
    t = _array_types.get(n)
    if t is None:
        import ctypes as C
        t = _array_types[n] = (C.c_void_p * n, C.c_int64 * n, C.c_int * n)
    return t


class _Image:
    """Packed weight operand of one NT product: a split-bf16 image of [Wa | Wb] or fp32 images of Wa, Wb."""
    __slots__ = ("kind", "a", "b")

    def __init__(self, kind, a, b=None):
        self.kind, self.a, self.b = kind, a, b


PACK_MAX_JOBS = 24           # images per hgv_pack16_weights_multi launch


def _pack_split(jobs, kind: Optional[str] = None) -> list:
    """Split images of ``jobs`` = [(wa, wb, transposed, n, k1, k2), ...], one launch per PACK_MAX_JOBS of them."""
    kind = kind or _split_kind()
    if _projection == "auto" and kind == "f16x3":
        # both arithmetics side by side; the product picks one per launch on the device
        return [_Image("auto", h.a, b.a) for h, b in zip(_pack_split(jobs, "f16x3+"), _pack_split(jobs, "bf16x6"))]
    kind = kind.rstrip("+")
    out = []
    for j0 in range(0, len(jobs), PACK_MAX_JOBS):
        part = jobs[j0:j0 + PACK_MAX_JOBS]
        nj = len(part)
        sizes = [_split_image_floats(n, k1, k2, kind) for (_, _, _, n, k1, k2) in part]
        buf = torch.empty(sum(sizes), dtype=torch.float32, device=part[0][0].device)
        vp_t, i64_t, i32_t = _ctypes_arrays(nj)
        W1, W2, Op = vp_t(), vp_t(), vp_t()
        sn1, sk1, sn2, sk2 = i64_t(), i64_t(), i64_t(), i64_t()
        K1, K2, N = i32_t(), i32_t(), i32_t()
        off = 0
        for q, ((wa, wb, t, n, k1, k2), sz) in enumerate(zip(part, sizes)):
            piece = buf[off:off + sz]
            off += sz
            out.append(_Image(kind, piece))
            W1[q], Op[q] = wa.data_ptr(), piece.data_ptr()
            sn1[q], sk1[q] = (wa.stride(1), wa.stride(0)) if t else (wa.stride(0), wa.stride(1))
            if wb is not None:
                W2[q] = wb.data_ptr()
                sn2[q], sk2[q] = (wb.stride(1), wb.stride(0)) if t else (wb.stride(0), wb.stride(1))
            K1[q], K2[q], N[q] = k1, k2, n
        _lib.call(_SPLIT_KINDS[kind][2], nj, W1, sn1, sk1, K1, W2, sn2, sk2, K2, N, Op, _lib.stream())
    return out


class _ImageScope:
    """Images packed up front for one loss evaluation, and the requests seen during it (next evaluation's plan)."""
    __slots__ = ("images", "requests")

    def __init__(self):
        self.images, self.requests = {}, {}


_image_scope: Optional[_ImageScope] = None


def _plannable(w) -> bool:
    """A weight the plan can find again: a leaf parameter or a view of one (e.g. a column block of a Linear weight)."""
    if w is None:
        return True
    base = w._base if w._base is not None else w
    return isinstance(base, torch.nn.Parameter) and base.is_leaf


def _wkey(w):
    return None if w is None else (w.data_ptr(), tuple(w.shape), w.stride())


def _wref(w):
    """(weak reference to the owning parameter, geometry of the view) - enough to find the same memory next time."""
    import weakref
    base = w._base if w._base is not None else w
    return weakref.ref(base), tuple(w.shape), w.stride(), w.storage_offset()


def _wresolve(entry):
    ref, shape, stride, offset = entry
    base = ref()
    if base is None or not base.is_cuda:
        return None
    if tuple(base.shape) == shape and base.stride() == stride and base.storage_offset() == offset:
        return base
    need = offset + sum((n - 1) * st for n, st in zip(shape, stride)) + 1 if all(shape) else 0
    if need > base.untyped_storage().nbytes() // base.element_size():
        return None
    return base.detach().as_strided(shape, stride, offset)


@contextlib.contextmanager
def weight_images(owner):
    """One loss evaluation of ``owner`` (a module) during which no parameter changes.

    Every projection operator packs the split-bf16 images of its weights when it runs - one small launch per
    operator, a dozen per train step.  Inside this context the images that the previous evaluation of the same
    owner (in the same grad mode) asked for are packed by ONE launch at entry, from the weights as they are now, and
    the operators pick them up; whatever is asked for beyond that is packed on the spot as usual and joins the plan
    of the next evaluation.  Nothing outlives the context except the plan (weak references and shapes), so an image
    can never be older than the evaluation that uses it.  Only leaf parameters are planned."""
    global _image_scope
    if _projection not in _SPLIT_KINDS:        # fp32 images, or 'auto' (two images per weight): packed where they are used
        yield
        return
    attr = "_hgv_image_plan_grad" if torch.is_grad_enabled() else "_hgv_image_plan_nograd"
    scope = _ImageScope()
    plan = getattr(owner, attr, None) or []
    jobs, keys = [], []
    for (ea, eb, t) in plan:
        wa, wb = _wresolve(ea), (_wresolve(eb) if eb is not None else None)
        if wa is None or (eb is not None and wb is None):
            continue
        n, k1 = (wa.size(1), wa.size(0)) if t else (wa.size(0), wa.size(1))
        k2 = 0 if wb is None else (wb.size(0) if t else wb.size(1))
        if _split_image_floats(n, k1, k2) > 0:
            jobs.append((wa, wb, t, n, k1, k2))
            keys.append((_wkey(wa), _wkey(wb), t))
    if jobs:
        for key, img in zip(keys, _pack_split(jobs)):
            scope.images[key] = (img, None)        # packed before anything else of the evaluation: valid on every stream
    prev, _image_scope = _image_scope, scope
    try:
        yield
    finally:
        _image_scope = prev
        object.__setattr__(owner, attr, list(scope.requests.values()))


def _images(*specs) -> list:
    """Weight images for ``x_a @ op(wa)^T (+ x_b @ op(wb)^T)``: ``specs`` are ``(wa, wb, transposed)``
    triples (None entries are skipped and returned as None) with op(w) = w or, with ``transposed``,
    w^T (the grad-input product).  All images an operator needs are built by one launch per kind,
    at every use (see :func:`_pack`), unless a :func:`weight_images` context has packed them already."""
    out = [None] * len(specs)
    split, plain = [], []
    scope = _image_scope
    for i, sp in enumerate(specs):
        if sp is None:
            continue
        wa, wb, t = sp
        n, k1 = (wa.size(1), wa.size(0)) if t else (wa.size(0), wa.size(1))
        k2 = 0 if wb is None else (wb.size(0) if t else wb.size(1))
        if _split_kind() is not None and _split_image_floats(n, k1, k2) > 0 and \
                (_projection != "auto" or _split_image_floats(n, k1, k2, "bf16x6") > 0):
            if scope is not None and _plannable(wa) and _plannable(wb):
                key = (_wkey(wa), _wkey(wb), t)
                if key not in scope.requests:
                    scope.requests[key] = (_wref(wa), _wref(wb) if wb is not None else None, t)
                hit = scope.images.get(key)
                if hit is not None and (hit[1] is None or hit[1] == _lib.stream()):
                    out[i] = hit[0]
                    continue
            split.append((i, wa, wb, t, n, k1, k2))
        else:
            plain.append((i, wa, wb, t))
    if split:
        imgs = _pack_split([(wa, wb, t, n, k1, k2) for (_, wa, wb, t, n, k1, k2) in split])
        for (i, wa, wb, t, _n, _k1, _k2), img in zip(split, imgs):
            out[i] = img
            if scope is not None and _plannable(wa) and _plannable(wb):      # the second encode pass reuses it
                scope.images[(_wkey(wa), _wkey(wb), t)] = (img, _lib.stream())
    # fp32 images: two weights per image at most, four weights per launch
    flat = [(i, which, w, t) for (i, wa, wb, t) in plain for (which, w) in (("a", wa), ("b", wb)) if w is not None]
    packed = {}
    for j0 in range(0, len(flat), 4):
        group = flat[j0:j0 + 4]
        for (i, which, _, _), pk in zip(group, _pack(*[(w, t) for (_, _, w, t) in group])):
            packed[(i, which)] = pk
    for (i, wa, wb, t) in plain:
        out[i] = _Image("fp32", packed[(i, "a")], packed.get((i, "b")))
    return out


def _image(wa: torch.Tensor, wb: Optional[torch.Tensor], transposed: bool) -> _Image:
    return _images((wa, wb, transposed))[0]


# HGVAE_CHECK_PADS=1: every projection checks (with a host synchronisation) that the pad columns [K, ld) of its operands are
# zero.  The f16x3 kernels take the largest entry of a row over the whole 16-byte chunks they load, so a stray value in a pad
# column would set the row's scale; every buffer this package produces keeps its pads zero, and this switch proves it on a
# given workload.
_CHECK_PADS = os.environ.get("HGVAE_CHECK_PADS", "0") == "1"


def _assert_zero_pads(*operands) -> None:
    for t in operands:
        if t is None or ld_of(t) <= t.size(1):
            continue
        pads = t.as_strided((t.size(0), ld_of(t) - t.size(1)), (t.stride(0), 1), t.storage_offset() + t.size(1))
        if t.size(0) and float(pads.abs().max()) != 0.0:
            raise _lib.HgvError(f"projection operand of shape {tuple(t.shape)} (leading dimension {ld_of(t)}) has non-zero pad columns")


# A level that wants its producing projection to store d = dropout(relu(y)) itself leaves (relu, p) here; the projection
# that forms the level's pre-activation (``level_out``) takes it when its shape allows and reports in _act_done.
_act_request = None
_act_done = None
_nt_act = os.environ.get("HGVAE_NT_ACT", "1") != "0"            # 0: the statistics pass applies relu / dropout (A/B)
_nt_stats = os.environ.get("HGVAE_NT_STATS", "1") != "0"        # 0: a pass of its own reads the stored level for its statistics


def _take_act(m: int, n_out: int, k1: int, k2: int, kind: str, device):
    """(relu, p, mask, seed, offset, counter, second draw) for a projection that stores the activated level, or None.  A
    request for two draws (``bn_level_pair``) is met with both, taken in order: second draw = (mask, offset)."""
    global _act_request
    req, _act_request = _act_request, None
    if req is None or kind != "f16x3" or m < max(2, DEFER_MIN_ROWS):
        return None
    lib = _lib.load()
    if len(req) == 3:
        if not lib.hgv_gemm_nt16h_act2_supported(m, n_out, k1, k2):
            return None
        relu, p, _ = req
        mask, seed, offset, ctr = current_noise().dropout((m, n_out), device)
        mask2, _, offset2, _ = current_noise().dropout((m, n_out), device)
        return relu, p, mask, seed, offset, ctr, (mask2, offset2)
    if not lib.hgv_gemm_nt16h_act_supported(n_out, k1, k2):
        return None
    relu, p = req
    mask, seed, offset, ctr = current_noise().dropout((m, n_out), device)
    return relu, p, mask, seed, offset, ctr, None


_act_ready = None            # keep bits of a pending activated store, already being generated on the side stream
_bits_early = os.environ.get("HGVAE_BITS_EARLY", "1") != "0"    # 0: the keep bits are generated right in front of the product
_side_streams: dict = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = (device.index if device.index is not None else torch.cuda.current_device())
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def _start_level_bits(m: int, n_out: int, k1: int, k2: int, device) -> None:
    """Called by an operator that has other launches in front of its level product (the aggregation of a SAGEConv): the
    keep bits depend on nothing but the noise stream, so their kernel - bound by the generator's integer multiplies - goes
    to a side stream now and runs beside those launches; the product waits for it."""
    global _act_ready
    if _act_request is None or not _bits_early or torch.cuda.is_current_stream_capturing() or on_branch_stream():
        return
    act = _take_act(m, n_out, k1, k2, "f16x3", device)
    if act is None:
        return
    relu, p, mask, seed, offset, ctr, second = act
    words = ((n_out + 3) // 4 * 4 + 31) // 32
    bits = torch.empty((m, words), dtype=torch.int32, device=device)
    bits2 = torch.empty((m, words), dtype=torch.int32, device=device) if second is not None else None
    main, side = torch.cuda.current_stream(device), _side_stream(device)
    side.wait_stream(main)                 # the buffer may be a recycled block whose last use is still queued on the main stream
    for t in (mask, second[0] if second is not None else None):
        if t is not None:
            t.record_stream(side)          # an explicit mask lives until the side stream has read it
    with torch.cuda.stream(side):
        _lib.call("hgv_dropout_bits", m, n_out, float(p), _lib.ptr(mask), seed, offset, ctr, bits.data_ptr(), words, _lib.stream())
        if second is not None:
            _lib.call("hgv_dropout_bits", m, n_out, float(p), _lib.ptr(second[0]), seed, second[1], ctr, bits2.data_ptr(), words,
                      _lib.stream())
        done = side.record_event()
    bits.record_stream(side)
    if bits2 is not None:
        bits2.record_stream(side)
    _act_ready = (m, n_out, relu, p, bits, words, done, (bits2, seed, offset, second[1], ctr) if second is not None else None)


_act_queue: list = []        # keep bits of the NEXT levels, in the order their products will ask for them (start_bits_ahead)
_bits_ahead = os.environ.get("HGVAE_BITS_AHEAD", "1") != "0"    # 0: no keep bits are generated levels ahead (A/B)


def start_bits_ahead(levels, training: bool, device) -> None:
    """For a run of levels that follow each other with nothing stochastic in between (the decoder's three, gvae/gvae_skipcon.py:
    121-137): ``levels`` = [(rows, n_out, k1, k2, relu, p), ...] in call order.  Their dropout draws are reserved now, in
    that order - exactly the draws the levels would make one after the other - and the keep-bit kernels go to the side stream
    at once, so all but the first run beside the products in front of them.  Stops at the first level whose product would not
    store the activated level (it draws for itself, later, and nothing behind it may draw earlier)."""
    drop_bits_ahead()
    if (not _bits_ahead or not _bits_early or not _nt_act or not _defer or not training or not torch.is_grad_enabled()
            or torch.cuda.is_current_stream_capturing() or on_branch_stream()):
        return
    lib = _lib.load()
    main, side = torch.cuda.current_stream(device), _side_stream(device)
    for (m, n_out, k1, k2, relu, p) in levels:
        if p < DEFER_MIN_P or m < max(2, DEFER_MIN_ROWS) or not lib.hgv_gemm_nt16h_act_supported(n_out, k1, k2):
            break
        mask, seed, offset, ctr = current_noise().dropout((m, n_out), device)
        words = ((n_out + 3) // 4 * 4 + 31) // 32
        bits = torch.empty((m, words), dtype=torch.int32, device=device)
        # behind the draw: an explicit mask may still be on its way to the device on the main stream, and the bit buffer may
        # be a recycled block whose last use is still queued there
        side.wait_stream(main)
        if mask is not None:
            mask.record_stream(side)
        with torch.cuda.stream(side):
            _lib.call("hgv_dropout_bits", m, n_out, float(p), _lib.ptr(mask), seed, offset, ctr, bits.data_ptr(), words, _lib.stream())
            done = side.record_event()
        bits.record_stream(side)
        _act_queue.append((m, n_out, bool(relu), float(p), bits, words, done, None))


def drop_bits_ahead() -> None:
    """Forget keep bits prepared ahead that no product took (their draws stay reserved)."""
    _act_queue.clear()


def _gemm_nt(xa, img: _Image, xb, bias, n_out: int, level_out: bool = False) -> torch.Tensor:
    """out[M, n_out] = xa @ Wa^T (+ xb @ Wb^T) (+ bias) with the weight operand ``img`` (see :func:`_image`).
    xa / xb must already be in the padded layout.  ``level_out``: this product is the pre-activation a level asked for
    (see ``_act_request``)."""
    global _act_done, _act_ready, _act_request
    m = xa.size(0)
    if _CHECK_PADS:
        _assert_zero_pads(xa, xb)
    out = new_padded(m, n_out, xa.device, zero_pad=False)
    ev = _probe.add("nt", (m, xa.size(1) + (xb.size(1) if xb is not None else 0), n_out)) if _probe is not None else None
    ready, _act_ready = (_act_ready, None) if level_out else (None, _act_ready)
    if level_out and ready is None and _act_request is not None and _act_queue:
        ready = _act_queue.pop(0)
        if (ready[2], ready[3]) != (bool(_act_request[0]), float(_act_request[1])) or img.kind != "f16x3":
            raise _lib.HgvError("keep bits prepared ahead do not belong to this level")
        _act_request = None
    act = (_take_act(m, n_out, xa.size(1), xb.size(1) if xb is not None else 0, img.kind, xa.device)
           if (level_out and ready is None and _act_request is not None) else None)
    if ready is not None and (ready[0], ready[1]) != (m, n_out):
        raise _lib.HgvError("keep bits were prepared for another product")
    if ready is not None or act is not None:
        if ready is not None:
            _, _, relu, p, bits, words, done, two = ready
            torch.cuda.current_stream(xa.device).wait_event(done)
        else:
            relu, p, mask, seed, offset, ctr, second = act
            words = (ld_of(out) + 31) // 32
            bits = torch.empty((m, words), dtype=torch.int32, device=xa.device)
            _lib.call("hgv_dropout_bits", m, n_out, float(p), _lib.ptr(mask), seed, offset, ctr, bits.data_ptr(), words, _lib.stream())
            two = None
            if second is not None:
                bits2 = torch.empty((m, words), dtype=torch.int32, device=xa.device)
                _lib.call("hgv_dropout_bits", m, n_out, float(p), _lib.ptr(second[0]), seed, second[1], ctr, bits2.data_ptr(), words,
                          _lib.stream())
                two = (bits2, seed, offset, second[1], ctr)
        if ev:
            ev[0].record()
        if two is not None:
            # two dropout draws of this product: both activated matrices and both sets of column sums in one pass
            bits2, seed, offset, offset2, ctr = two
            out2 = new_padded(m, n_out, xa.device, zero_pad=False)
            nparts = _lib.load().hgv_gemm_nt16h_act2_parts(m)
            partial = torch.empty((2, nparts, 2, ld_of(out)), dtype=torch.float32, device=xa.device)
            _lib.call("hgv_gemm_nt16h_act2_stats", xa.data_ptr(), ld_of(xa), xa.size(1), _lib.ptr(xb), ld_of(xb) if xb is not None else 0,
                      xb.size(1) if xb is not None else 0, img.a.data_ptr(), _lib.ptr(bias), m, n_out, out.data_ptr(), out2.data_ptr(),
                      ld_of(out), bits.data_ptr(), bits2.data_ptr(), words, int(relu), 1.0 / (1.0 - float(p)), partial[0].data_ptr(),
                      partial[1].data_ptr(), _lib.stream())
            if ev:
                ev[1].record()
            _act_done = (out.data_ptr(), m, n_out, relu, p, partial, (out2, seed, offset, offset2, ctr))
            return out
        common = (xa.data_ptr(), ld_of(xa), xa.size(1), _lib.ptr(xb), ld_of(xb) if xb is not None else 0,
                  xb.size(1) if xb is not None else 0, img.a.data_ptr(), _lib.ptr(bias), m, n_out, out.data_ptr(), ld_of(out),
                  bits.data_ptr(), words, int(relu), 1.0 / (1.0 - float(p)))
        partial = None
        if _nt_stats:                  # the product also leaves the column statistics of what it stores
            nparts = _lib.load().hgv_gemm_nt16h_act_parts(m, n_out, xa.size(1), xb.size(1) if xb is not None else 0)
            partial = torch.empty((nparts, 2, ld_of(out)), dtype=torch.float32, device=xa.device)
            _lib.call("hgv_gemm_nt16h_act_stats", *common, partial.data_ptr(), _lib.stream())
        else:
            _lib.call("hgv_gemm_nt16h_act", *common, _lib.stream())
        if ev:
            ev[1].record()
        _act_done = (out.data_ptr(), m, n_out, relu, p, partial)
        return out
    if ev:
        ev[0].record()
    if img.kind == "auto":
        # one pass over the operands decides on the device; both products are enqueued and exactly one of them runs
        flag = _workspace(256, xa.device, "range_flag")
        a2 = (_lib.ptr(xb), ld_of(xb) if xb is not None else 0, xb.size(1) if xb is not None else 0)
        _lib.call("hgv_row_range_flag", xa.data_ptr(), ld_of(xa), xa.size(1), *a2, m, 16, flag.data_ptr(), _lib.stream())
        for name, image, run_if_set in (("hgv_gemm_nt16h_gated", img.a, 0), ("hgv_gemm_nt16_gated", img.b, 1)):
            _lib.call(name, xa.data_ptr(), ld_of(xa), xa.size(1), *a2, image.data_ptr(), _lib.ptr(bias), m, n_out,
                      out.data_ptr(), ld_of(out), flag.data_ptr(), run_if_set, _lib.stream())
    elif img.kind in _SPLIT_KINDS:
        _lib.call(_SPLIT_KINDS[img.kind][3], xa.data_ptr(), ld_of(xa), xa.size(1), _lib.ptr(xb),
                  ld_of(xb) if xb is not None else 0, xb.size(1) if xb is not None else 0, img.a.data_ptr(),
                  _lib.ptr(bias), m, n_out, out.data_ptr(), ld_of(out), _lib.stream())
    else:
        _lib.call("hgv_gemm_nt_packed", xa.data_ptr(), ld_of(xa), xa.size(1), img.a.data_ptr(),
                  _lib.ptr(xb), ld_of(xb) if xb is not None else 0, xb.size(1) if xb is not None else 0,
                  _lib.ptr(img.b), _lib.ptr(bias), m, n_out, out.data_ptr(), ld_of(out), _lib.stream())
    if ev:
        ev[1].record()
    return out


_direct_grads = False


@contextlib.contextmanager
def direct_weight_grads(enabled: bool = True):
    """Inside this context the projection operators ADD weight / bias gradients straight into existing
    ``param.grad`` buffers (e.g. the views of a :class:`FlatGradBucket`) from the reduction kernel and
    return ``None`` for them to autograd, instead of returning a fresh tensor that autograd then adds
    with one small kernel per parameter use.  Only leaf parameters with a preallocated contiguous
    fp32 ``.grad`` of the right shape take the shortcut; everything else keeps the normal path.  Do
    not combine with ``torch.autograd.grad`` or gradient hooks on those parameters."""
    global _direct_grads
    prev, _direct_grads = _direct_grads, bool(enabled)
    try:
        yield
    finally:
        _direct_grads = prev


_trunk_stream: Optional[int] = None      # raw handle of the stream the step runs on while part of it runs on a branch stream
_in_branch = False                       # True only while the operators of a branch are being created (forward side)


@contextlib.contextmanager
def branch_of(trunk: "torch.cuda.Stream"):
    """Marks the operators created inside (with another stream current) as a branch next to ``trunk``: in their
    backward pass - which autograd runs on the branch stream, possibly at the same time as the trunk's - weight and
    bias gradients are returned to autograd instead of being added into ``.grad`` by the kernels (see
    :func:`direct_weight_grads`), and weight images packed on the spot are not shared across the two streams.
    The trunk handle outlives the context - the backward pass comes later - until :func:`end_branches` (called by the
    step that owns the branch after its backward pass) or the next :func:`no_branches`; the forward-side decisions
    (BatchNorm running statistics through stand-ins) are taken only inside the context."""
    global _trunk_stream, _in_branch
    _trunk_stream = trunk.cuda_stream
    prev, _in_branch = _in_branch, True
    try:
        yield
    finally:
        _in_branch = prev


def end_branches() -> None:
    """The backward pass of a branched evaluation is over: later operators on any stream are ordinary ones."""
    global _trunk_stream
    _trunk_stream = None


def no_branches() -> None:
    """The evaluation that starts now runs on one stream (forgets the trunk of an earlier branched evaluation)."""
    end_branches()
    _branch_updates.clear()


def on_branch_stream() -> bool:
    return _in_branch and _trunk_stream is not None and _lib.stream() != _trunk_stream


_branch_updates: list = []      # (BatchNorm module, [2, H] stand-in holding momentum * (mean, unbiased variance), momentum)


def apply_branch_updates() -> None:
    """Running-statistics updates that the branch deferred; call on the trunk stream after the branch has joined."""
    import numpy as np
    bufs, keeps, adds, counters = [], [], [], []
    for bn, stand_in, momentum in _branch_updates:
        keep = float(np.float32(1.0) - np.float32(momentum))          # the kernel's (1.f - momentum)
        if bn.running_mean is not None:
            bufs.append(bn.running_mean); keeps.append(keep); adds.append(stand_in[0])
        if bn.running_var is not None:
            bufs.append(bn.running_var); keeps.append(keep); adds.append(stand_in[1])
        if bn.num_batches_tracked is not None:
            counters.append(bn.num_batches_tracked)
    with torch.no_grad():
        # all levels in three launches (the same multiply, then add, per buffer as one level at a time would do)
        if bufs:
            torch._foreach_mul_(bufs, keeps)
            torch._foreach_add_(bufs, adds)
        if counters:
            torch._foreach_add_(counters, 1)
    _branch_updates.clear()


def _grad_buffer(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """``p.grad`` when gradients may be accumulated into it in place, else None."""
    if not _direct_grads or p is None or not p.is_leaf or not p.requires_grad:
        return None
    if _trunk_stream is not None and _lib.stream() != _trunk_stream:
        # a branch stream: adds into .grad would race with the trunk's.  The bucket's branch buffer if there is one
        # (FlatGradBucket.enable_branch), else back to autograd, which orders the two
        tw = getattr(p, "_hgv_branch_grad", None)
        return tw if tw is not None and tw.shape == p.shape and tw.device == p.device else None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or not g.is_contiguous() or g.device != p.device:
        return None
    return g


def _tn16_name() -> str:
    return "hgv_gemm_tn16h" if _projection == "f16x3" and _tn_f16 else "hgv_gemm_tn16"


# Weight gradients of "f16x3" stay on the three-piece bf16 kernel by default: the fp16 kernel has ONE scale per workgroup
# and operand (both operands are reduced over their rows), so a column more than 2^16 below the largest entry of its
# 144-column group loses bits, and inside the train step it is only 2 % faster.  HGVAE_TN_F16=1 selects it.
_tn_f16 = os.environ.get("HGVAE_TN_F16", "0") == "1"


def _tn16_ok(m, n, k1, k2, *mats) -> bool:
    """Weight gradients run on the 16-bit matrix cores (split operands, fp32-accurate) when the projection
    arithmetic is a split kind, the shape is one the kernel covers and the operands are 16-byte aligned."""
    if (_projection not in _SPLIT_KINDS and _projection != "auto") or not _lib.load().hgv_gemm_tn16_supported(m, n, k1, k2):
        return False
    return all(t.data_ptr() % 16 == 0 and ld_of(t) % 4 == 0 for t in mats)


def _gemm_tn(g, x, want_bias: bool, out: Optional[torch.Tensor] = None, bias_out: Optional[torch.Tensor] = None,
             accumulate: bool = False):
    """(g^T @ x  [N, K],  column sums of g or None); g, x in the padded layout.  ``out`` may be a
    column block (unit column stride) of a wider gradient; by default a new contiguous matrix.
    With ``accumulate`` the results are added to ``out`` / ``bias_out`` instead of overwriting them."""
    m, n, k = g.size(0), g.size(1), x.size(1)
    gw = torch.empty(n, k, dtype=torch.float32, device=g.device) if out is None else out
    if tuple(gw.shape) != (n, k) or gw.stride(1) != 1 or gw.stride(0) < k:
        raise _lib.HgvError(f"weight-gradient output of shape {tuple(gw.shape)} / strides {gw.stride()} for a {n}x{k} product")
    if accumulate and (out is None or (want_bias and bias_out is None)):
        raise _lib.HgvError("accumulating weight gradients needs existing output buffers")
    gb = (bias_out if bias_out is not None else torch.empty(n, dtype=torch.float32, device=g.device)) if want_bias else None
    ev = _probe.add("tn", (m, n, k)) if _probe is not None else None
    if ev:
        ev[0].record()
    if _tn16_ok(m, n, k, 0, g, x):
        ws = _workspace(_lib.load().hgv_gemm_tn16_ws(m, n, k, 0), g.device, "gemm_tn")
        _lib.call(_tn16_name(), g.data_ptr(), ld_of(g), n, x.data_ptr(), ld_of(x), k, gw.data_ptr(), gw.stride(0),
                  None, 0, 0, None, 0, m, _lib.ptr(gb), 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _lib.stream())
    else:
        ws = _workspace(_lib.load().hgv_gemm_tn_ws(m, n, k), g.device, "gemm_tn")
        _lib.call("hgv_gemm_tn", g.data_ptr(), ld_of(g), n, x.data_ptr(), ld_of(x), k, m, gw.data_ptr(), gw.stride(0),
                  _lib.ptr(gb), 1 if accumulate else 0, ws.data_ptr(), ws.numel(), _lib.stream())
    if ev:
        ev[1].record()
    return gw, gb


_tn_gg = os.environ.get("HGVAE_TN_GG", "1") != "0"          # 0: two separate products (A/B measurements)


def _check_gw(gw, n, k):
    if tuple(gw.shape) != (n, k) or gw.stride(1) != 1 or gw.stride(0) < k:
        raise _lib.HgvError(f"weight-gradient output of shape {tuple(gw.shape)} / strides {gw.stride()} for a {n}x{k} product")


def _gemm_tn2(g, xa, xb, want_bias: bool, out_a=None, out_b=None, bias_out=None, accumulate: bool = False):
    """(g^T @ xa [N, Ka], g^T @ xb [N, Kb], column sums of g or None) in one pass over g; see :func:`_gemm_tn`."""
    m, n, ka, kb = g.size(0), g.size(1), xa.size(1), xb.size(1)
    if accumulate and (out_a is None or out_b is None or (want_bias and bias_out is None)):
        raise _lib.HgvError("accumulating weight gradients needs existing output buffers")
    ga = torch.empty(n, ka, dtype=torch.float32, device=g.device) if out_a is None else out_a
    gb_ = torch.empty(n, kb, dtype=torch.float32, device=g.device) if out_b is None else out_b
    _check_gw(ga, n, ka)
    _check_gw(gb_, n, kb)
    gbias = (bias_out if bias_out is not None else torch.empty(n, dtype=torch.float32, device=g.device)) if want_bias else None
    tn16 = _tn16_ok(m, n, ka, kb, g, xa, xb)
    ws = _workspace((_lib.load().hgv_gemm_tn16_ws if tn16 else _lib.load().hgv_gemm_tn2_ws)(m, n, ka, kb), g.device, "gemm_tn")
    ev = _probe.add("tn", (m, n, ka + kb)) if _probe is not None else None
    if ev:
        ev[0].record()
    _lib.call(_tn16_name() if tn16 else "hgv_gemm_tn2", g.data_ptr(), ld_of(g), n, xa.data_ptr(), ld_of(xa), ka, ga.data_ptr(), ga.stride(0),
              xb.data_ptr(), ld_of(xb), kb, gb_.data_ptr(), gb_.stride(0), m, _lib.ptr(gbias), 1 if accumulate else 0,
              ws.data_ptr(), ws.numel(), _lib.stream())
    if ev:
        ev[1].record()
    return ga, gb_, gbias


def _gemm_tn_gg(ga, gb, x, out_a=None, out_b=None, cs_a=None, cs_b=None, accumulate: bool = False):
    """(ga^T @ x, gb^T @ x, column sums of ga, column sums of gb) in one pass over x for two gradient matrices of the same
    width; see :func:`_gemm_tn`."""
    m, n, k = ga.size(0), ga.size(1), x.size(1)
    if gb.size(1) != n or gb.size(0) != m or x.size(0) != m:
        raise _lib.HgvError(f"two-gradient product of {tuple(ga.shape)} and {tuple(gb.shape)} with {tuple(x.shape)}")
    if accumulate and (out_a is None or out_b is None or cs_a is None or cs_b is None):
        raise _lib.HgvError("accumulating weight gradients needs existing output buffers")
    if out_a is None and out_b is None and cs_a is None and cs_b is None and n % 4 == 0 and _tn16_ok(m, 2 * n, k, 0, ga, gb, x):
        # fresh outputs: one product on the 16-bit matrix cores, the two results as row blocks of one matrix
        both = torch.empty(2 * n, k, dtype=torch.float32, device=x.device)
        sums = torch.empty(2 * n, dtype=torch.float32, device=x.device)
        ws = _workspace(_lib.load().hgv_gemm_tn16_ws(m, 2 * n, k, 0), x.device, "gemm_tn")
        ev = _probe.add("tn", (m, 2 * n, k)) if _probe is not None else None
        if ev:
            ev[0].record()
        _lib.call("hgv_gemm_tn16_gg", ga.data_ptr(), ld_of(ga), gb.data_ptr(), ld_of(gb), n, x.data_ptr(), ld_of(x), k, m,
                  both.data_ptr(), both.stride(0), sums.data_ptr(), 0, ws.data_ptr(), ws.numel(), _lib.stream())
        if ev:
            ev[1].record()
        return both[:n], both[n:], sums[:n], sums[n:]
    wa = torch.empty(n, k, dtype=torch.float32, device=x.device) if out_a is None else out_a
    wb = torch.empty(n, k, dtype=torch.float32, device=x.device) if out_b is None else out_b
    _check_gw(wa, n, k)
    _check_gw(wb, n, k)
    ca = torch.empty(n, dtype=torch.float32, device=x.device) if cs_a is None else cs_a
    cb = torch.empty(n, dtype=torch.float32, device=x.device) if cs_b is None else cs_b
    ws = _workspace(_lib.load().hgv_gemm_tn_gg_ws(m, n, k), x.device, "gemm_tn")
    ev = _probe.add("tn", (m, 2 * n, k)) if _probe is not None else None
    if ev:
        ev[0].record()
    _lib.call("hgv_gemm_tn_gg", ga.data_ptr(), ld_of(ga), gb.data_ptr(), ld_of(gb), n, x.data_ptr(), ld_of(x), k, m, wa.data_ptr(),
              wa.stride(0), wb.data_ptr(), wb.stride(0), ca.data_ptr(), cb.data_ptr(), 1 if accumulate else 0, ws.data_ptr(),
              ws.numel(), _lib.stream())
    if ev:
        ev[1].record()
    return wa, wb, ca, cb


def _wgrad_gg(ga, gb, x, wa, wb, aff, bias, want_bias: bool):
    """Weight gradients of  y = aggregate(x @ wa^T) + x @ wb^T + bias  from ga = the transposed aggregate of the output
    gradient and gb = the output gradient: (gwa, gwb, gbias) for autograd, None where a gradient went straight into
    ``.grad``.  ``aff`` as in :func:`_wgrad`."""
    bufs = [_grad_buffer(wa), _grad_buffer(wb)]
    bbuf = _grad_buffer(bias) if want_bias else None
    direct = bufs[0] is not None and bufs[1] is not None and (not want_bias or bbuf is not None)
    if aff is None:
        if direct:
            # the column sums of ga are not a gradient of anything: they land in a scratch vector
            scratch = torch.zeros(ga.size(1), dtype=torch.float32, device=x.device)
            cb = bbuf if want_bias else torch.zeros(ga.size(1), dtype=torch.float32, device=x.device)
            _gemm_tn_gg(ga, gb, x, bufs[0], bufs[1], scratch, cb, accumulate=True)
            return None, None, None
        gwa, gwb, _, cb = _gemm_tn_gg(ga, gb, x)
        return gwa, gwb, (cb if want_bias else None)
    ra, rb, ca, cb = _gemm_tn_gg(ga, gb, x)
    _contribute(aff, wa, ra, ca)
    _contribute(aff, wb, rb, cb)
    dsts = bufs if direct else [torch.empty_like(wa), torch.empty_like(wb)]
    jobs = [(dsts[0], ra, aff, ca), (dsts[1], rb, aff, cb)]
    gbias = None
    if want_bias:
        gbias = bbuf if direct else torch.empty_like(cb)
        jobs.append((gbias.view(-1, 1), cb.view(-1, 1), None))
    _wgrad_finish(jobs, cb, accumulate=direct)
    if direct:
        return None, None, None
    return dsts[0], dsts[1], gbias


def _weight_grad(g, x, w, bias=None, want_bias: bool = False):
    """Weight (and bias) gradient of y = x @ w^T + bias: (gw, gb) for autograd, with None in place of a
    gradient that was added straight into the parameter's ``.grad`` buffer."""
    wbuf = _grad_buffer(w)
    bbuf = _grad_buffer(bias) if want_bias else None
    if wbuf is not None and (not want_bias or bbuf is not None):
        _gemm_tn(g, x, want_bias, wbuf, bbuf, accumulate=True)
        return None, None
    gw, gb = _gemm_tn(g, x, want_bias)
    return gw, gb


def _weight_grad2(g, xa, wa, xb, wb, bias=None, want_bias: bool = False):
    """Weight gradients of y = xa @ wa^T + xb @ wb^T + bias in one pass over g: (gwa, gwb, gb) for autograd,
    None where the gradient was added straight into the parameter's ``.grad`` buffer."""
    abuf, bbuf_w = _grad_buffer(wa), _grad_buffer(wb)
    bbuf = _grad_buffer(bias) if want_bias else None
    if abuf is not None and bbuf_w is not None and (not want_bias or bbuf is not None):
        _gemm_tn2(g, xa, xb, want_bias, abuf, bbuf_w, bbuf, accumulate=True)
        return None, None, None
    return _gemm_tn2(g, xa, xb, want_bias)


_bwd_fuse = os.environ.get("HGVAE_BWD_FUSE", "1") != "0"        # 0: every level runs its backward pass itself (A/B)
_bwd_fuse_add = os.environ.get("HGVAE_BWD_FUSE_ADD", "1") != "0"  # 0: ... every level with two consumers does (A/B)


def _grad_input(g, img: _Image, g2, n_in: int, aff) -> torch.Tensor:
    """The grad-input product  g W (+ g2 W2)  of an operator with respect to an operand of ``n_in`` columns.  When that
    operand was a deferred level (``aff``) with this operator as its ONLY consumer, and the operator has already left its
    weight-gradient parts (so the column sums of the level's backward are known), the product stores the level's dy
    instead of its incoming gradient - relu / dropout / batch-norm backward applied on the way out, with d read once by
    the product - and the level's backward pass over g and d disappears (``_GradParts.fused``)."""
    parts = aff[2] if (aff is not None and len(aff) > 2) else None
    lvl = parts.level if parts is not None else None
    if (not _bwd_fuse or not _alg_sums or lvl is None or parts.handles not in (1, 2) or parts.tainted or parts.fused is not None
            or img.kind != "f16x3" or img.b is not None):
        return _gemm_nt(g, img, g2, None, n_in)
    if parts.handles == 2 and (parts.shared or not _bwd_fuse_add):
        return _gemm_nt(g, img, g2, None, n_in)
    if parts.handles == 2 and not parts.complete():
        # the first of two consumers (a level that also feeds a skip connection): its share is written as it is and waits
        out = _gemm_nt(g, img, g2, None, n_in)
        parts.pending.append(out)
        return out
    d, gamma, beta, mean, rstd, relu, p, need_g, need_b = lvl
    m, h = d.shape
    lib = _lib.load()
    k1, k2 = g.size(1), (g2.size(1) if g2 is not None else 0)
    add = parts.pending[0] if (parts.handles == 2 and len(parts.pending) == 1) else None
    if (not parts.complete() or (parts.handles == 2 and add is None) or h != n_in or g.size(0) != m or ld_of(d) > 144 or not is_vec4(d)
            or any(w.size(1) != h for (w, _, _) in parts.parts)
            or not lib.hgv_gemm_nt16h_bwd_supported(m, h, k1, k2, 1 if add is not None else 0)
            or (add is not None and (tuple(add.shape) != (m, h) or not is_vec4(add)))):
        return _gemm_nt(g, img, g2, None, n_in)
    if _CHECK_PADS:
        _assert_zero_pads(g, g2)
    nparts = len(parts.parts)
    vp_t, i64_t, i32_t = _ctypes_arrays(nparts)
    W, R, CS = vp_t(), vp_t(), vp_t()
    ldw, ldr, css = i64_t(), i64_t(), i64_t()
    nn_ = i32_t()
    for q, (w, raw, cs) in enumerate(parts.parts):
        W[q], R[q], CS[q] = w.data_ptr(), raw.data_ptr(), cs.data_ptr()
        ldw[q], ldr[q], css[q], nn_[q] = w.stride(0), raw.stride(0), (cs.stride(0) if cs.numel() > 1 else 1), w.size(0)
    gbuf, bbuf = (_grad_buffer(gamma) if need_g else None), (_grad_buffer(beta) if need_b else None)
    direct = gbuf is not None and bbuf is not None
    dgamma = gbuf if direct else torch.empty(h, dtype=torch.float32, device=d.device)
    dbeta = bbuf if direct else torch.empty(h, dtype=torch.float32, device=d.device)
    sums = torch.empty(2, (h + 3) // 4 * 4, dtype=torch.float32, device=d.device)
    _lib.call("hgv_bn_bwd_sums_alg", h, mean.data_ptr(), rstd.data_ptr(), nparts, W, ldw, R, ldr, CS, css, nn_, sums.data_ptr(),
              dgamma.data_ptr(), dbeta.data_ptr(), 1 if direct else 0, _lib.stream())
    dy = new_padded(m, h, d.device, zero_pad=False)
    ev = _probe.add("nt", (m, k1 + k2, h)) if _probe is not None else None
    if ev:
        ev[0].record()
    _lib.call("hgv_gemm_nt16h_bwd", g.data_ptr(), ld_of(g), k1, _lib.ptr(g2), ld_of(g2) if g2 is not None else 0, k2, img.a.data_ptr(),
              m, h, dy.data_ptr(), ld_of(dy), d.data_ptr(), ld_of(d), int(relu), float(p), _lib.ptr(gamma), mean.data_ptr(),
              rstd.data_ptr(), sums.data_ptr(), _lib.ptr(add), ld_of(add) if add is not None else 0, _lib.stream())
    if ev:
        ev[1].record()
    parts.fused = (dy, None if direct else dgamma, None if direct else dbeta)
    return dy


class _DualLinear(torch.autograd.Function):
    """y = xa @ Wa^T + xb @ Wb^T + bias on the matrix-core kernels (xb / Wb / bias optional).  ``aff_a`` / ``aff_b``: the
    (s, t) vectors of an operand that arrives un-normalised (see :class:`Deferred`)."""

    @staticmethod
    def forward(ctx, xa, wa, xb, wb, bias, aff_a=None, aff_b=None):
        _lib.require_cuda(xa, wa)
        if wa.size(0) > MAX_GEMM_N or wa.size(1) > MAX_GEMM_N or (wb is not None and wb.size(1) > MAX_GEMM_N):
            raise _lib.HgvError(f"projection widths above {MAX_GEMM_N} are not supported "
                                f"(got {tuple(wa.shape)}{'' if wb is None else ' and ' + str(tuple(wb.shape))})")
        xa = as_padded(xa)
        xb = as_padded(xb) if xb is not None else None
        need = ctx.needs_input_grad
        folded = aff_a is not None or aff_b is not None
        fwd, ctx.pat, ctx.pbt = _images(None if folded else (wa, wb if xb is not None else None, False),
                                        (wa, None, True) if need[0] else None,
                                        (wb, None, True) if xb is not None and need[2] else None)
        if folded:
            fwd, bias_f = _image_affine(wa, aff_a, wb if xb is not None else None, aff_b, bias)
        else:
            bias_f = bias
        y = _gemm_nt(xa, fwd, xb, bias_f, wa.size(0), level_out=True)
        ctx.has_b, ctx.has_bias = xb is not None, bias is not None
        ctx.aff_a, ctx.aff_b = aff_a, aff_b
        ctx.save_for_backward(xa, wa, xb, wb, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        xa, wa, xb, wb, bias = ctx.saved_tensors
        g = as_padded(g)
        need = ctx.needs_input_grad
        gxa = gxb = None
        want_bias = ctx.has_bias and need[4]
        ops = [(xa, wa, ctx.aff_a, need[1])]
        if ctx.has_b:
            ops.append((xb, wb, ctx.aff_b, need[3]))
        gws, gb = _wgrad(g, ops, bias, want_bias)           # first: the operands' levels take their column sums from its parts
        gwa, gwb = gws[0], (gws[1] if ctx.has_b else None)
        if need[0]:
            gxa = _grad_input(g, ctx.pat, None, wa.size(1), ctx.aff_a)
        if ctx.has_b and need[2]:
            gxb = _grad_input(g, ctx.pbt, None, wb.size(1), ctx.aff_b)
        return gxa, gwa, gxb, gwb, gb, None, None


class _ConcatLinear(torch.autograd.Function):
    """y = [xa | xb] @ W^T + bias for one weight W [n_out, Ka + Kb] (a decoder level fed by the
    concatenation of the running activation and a skip tensor, gvae/gvae_skipcon.py:92-100).  The
    concatenation is never built, and both halves of the weight gradient are written straight into
    one [n_out, Ka + Kb] matrix.  ``aff_a`` / ``aff_b`` as in :class:`_DualLinear`."""

    @staticmethod
    def forward(ctx, xa, xb, w, bias, aff_a=None, aff_b=None):
        _lib.require_cuda(xa, xb, w)
        ka, kb = xa.size(1), xb.size(1)
        if w.size(1) != ka + kb:
            raise _lib.HgvError(f"weight of shape {tuple(w.shape)} for inputs of widths {ka} + {kb}")
        if max(w.size(0), ka, kb) > MAX_GEMM_N:
            raise _lib.HgvError(f"projection widths above {MAX_GEMM_N} are not supported (got {tuple(w.shape)})")
        xa, xb = as_padded(xa), as_padded(xb)
        need = ctx.needs_input_grad
        wa, wb = w[:, :ka], w[:, ka:]
        folded = aff_a is not None or aff_b is not None
        fwd, ctx.pat, ctx.pbt = _images(None if folded else (wa, wb, False), (wa, None, True) if need[0] else None,
                                        (wb, None, True) if need[1] else None)
        if folded:
            fwd, bias_f = _image_affine(wa, aff_a, wb, aff_b, bias)
        else:
            bias_f = bias
        y = _gemm_nt(xa, fwd, xb, bias_f, w.size(0), level_out=True)
        ctx.has_bias = bias is not None
        ctx.aff_a, ctx.aff_b = aff_a, aff_b
        ctx.save_for_backward(xa, xb, w, bias)
        return y

    @staticmethod
    def backward(ctx, g):
        xa, xb, w, bias = ctx.saved_tensors
        ka = xa.size(1)
        g = as_padded(g)
        need = ctx.needs_input_grad
        gxa = gxb = gw = gb = None
        want_bias = ctx.has_bias and need[3]
        if not need[2]:
            _taint(ctx.aff_a)
            _taint(ctx.aff_b)
        if need[2] or want_bias:
            wbuf = _grad_buffer(w) if need[2] else None
            bbuf = _grad_buffer(bias) if want_bias else None
            direct = wbuf is not None and (not want_bias or bbuf is not None)
            if ctx.aff_a is None and ctx.aff_b is None:
                if direct:                                      # add both halves into w.grad
                    _gemm_tn2(g, xa, xb, want_bias, wbuf[:, :ka], wbuf[:, ka:], bbuf, accumulate=True)
                else:
                    gw = torch.empty_like(w, memory_format=torch.contiguous_format)
                    _, _, gb = _gemm_tn2(g, xa, xb, want_bias, gw[:, :ka], gw[:, ka:])
                    if not need[2]:
                        gw = None
            else:
                # raw products with the stored operands, then  (G^T d) diag(s) + colsum(G) t^T  per half
                ra, rb, colsum = _gemm_tn2(g, xa, xb, True)
                _contribute(ctx.aff_a, w[:, :ka], ra, colsum)
                _contribute(ctx.aff_b, w[:, ka:], rb, colsum)
                if not direct:
                    gw = torch.empty_like(w, memory_format=torch.contiguous_format)
                    gb = torch.empty_like(colsum) if want_bias else None
                dst = wbuf if direct else gw
                jobs = [(dst[:, :ka], ra, ctx.aff_a), (dst[:, ka:], rb, ctx.aff_b)]
                if want_bias:
                    jobs.append(((bbuf if direct else gb).view(-1, 1), colsum.view(-1, 1), None))
                _wgrad_finish(jobs, colsum, accumulate=direct)
                if direct or not need[2]:
                    gw = None
                if direct:
                    gb = None
        # behind the weight gradient: the operands' levels take their column sums from its parts
        if need[0]:
            gxa = _grad_input(g, ctx.pat, None, ka, ctx.aff_a)
        if need[1]:
            gxb = _grad_input(g, ctx.pbt, None, xb.size(1), ctx.aff_b)
        return gxa, gxb, gw, gb, None, None


def _unwrap(x, ok: bool):
    """(stored operand, (s, t) or None) of a projection input; a :class:`Deferred` the projection cannot fold is written out."""
    if isinstance(x, Deferred):
        if not ok:
            return x.tensor(), None
        if x.parts is not None:
            x.parts.uses += 1
            x.parts.handles += 1
            x.taken += 1
            if x.taken > 1:
                x.parts.shared = True
        return x.d, (x.s, x.t, x.parts)
    return x, None


def concat_linear(xa, xb, w, bias=None):
    ok = _affine_ok(w.size(0), xa.size(1), xb.size(1))
    (xa, aff_a), (xb, aff_b) = _unwrap(xa, ok), _unwrap(xb, ok)
    return _ConcatLinear.apply(xa, xb, w, bias, aff_a, aff_b)


class _SageConv(torch.autograd.Function):
    """y = lin_l(aggregate(x)) + lin_r(x) as one operator (SAGEConv / GraphConv).

    When the output is narrower than the input the projection runs first (one product with the
    stacked weight [W_l ; W_r]) and the aggregation works on the narrow result: aggregation is
    linear, so this equals aggregate-then-project up to fp32 rounding and moves fewer bytes.
    The backward pass returns a single padded gradient for x (the add of the aggregation path and
    the root path is fused into the aggregation kernel).
    """

    @staticmethod
    def forward(ctx, x, wl, bl, wr, graph: Graph, mean: bool, aff=None):
        _lib.require_cuda(x, wl, wr)
        n_out, n_in = wl.shape
        if max(n_out, n_in) > MAX_GEMM_N:
            raise _lib.HgvError(f"projection widths above {MAX_GEMM_N} are not supported (got {tuple(wl.shape)})")
        x = as_padded(x)
        n = graph.num_nodes
        first = n_out < n_in and 2 * n_out <= MAX_GEMM_N
        gi = ctx.needs_input_grad[0]
        ctx.pt = ctx.plt = ctx.prt = None
        # aff = (s, t): x holds d of a deferred level and the operand meant is d * s + t (mean aggregation only)
        if first:
            bias2 = torch.cat([torch.zeros_like(bl), bl]) if bl is not None else None
            stacked_w = torch.cat([wl, wr])
            stacked, ctx.pt = _images(None if aff is not None else (stacked_w, None, False), (wl, wr, True) if gi else None)   # gx = gt wl + g wr
            if aff is not None:
                stacked, bias2 = _image_affine(stacked_w, aff, None, None, bias2)
            tr = _gemm_nt(x, stacked, None, bias2, 2 * n_out)
            y = new_padded(n, n_out, x.device)
            _agg_launch(graph.by_dst, n, tr[:, :n_out], n_out, 1 if mean else 0, None, None, None, tr[:, n_out:], y)
            ctx.save_for_backward(x, wl, wr, bl)
        else:
            # a deferred operand with a free pad column: the aggregation runs on d as on any matrix and leaves a
            # has-neighbours indicator in the pad column, which the projection takes as one more input column with the
            # weight wl t (mean_j(d_j s + t) = mean_j(d_j) s + t * indicator); without a pad column the aggregation
            # applies s / t to its finished rows itself
            ind = aff is not None and n_in % 4 != 0
            _start_level_bits(n, n_out, n_in + (1 if ind else 0), n_in, x.device)
            agg = new_padded(n, n_in, x.device, zero_pad=False)
            if ind:
                _agg_ind_launch(graph.by_dst, n, x, n_in, agg, n_in)
                agg = agg.as_strided((n, n_in + 1), agg.stride(), agg.storage_offset())
            elif aff is not None:
                _agg_affine_launch(graph.by_dst, n, x, n_in, aff[0], aff[1], agg)      # the aggregate of d * s + t itself
            else:
                _agg_launch(graph.by_dst, n, x, n_in, 1 if mean else 0, None, None, None, None, agg)
            # Input gradient: aggregate the output gradient first when that is not the wider side
            # (gx = aggT(g) wl + g wr: one K-concatenated product), else project first (two products,
            # the add fused into the aggregation).
            fwd_spec = None if aff is not None else (wl, wr, False)
            if gi and n_out <= n_in:
                fwd, ctx.pt = _images(fwd_spec, (wl, wr, True))
            else:
                fwd, ctx.plt, ctx.prt = _images(fwd_spec, (wl, None, True) if gi else None, (wr, None, True) if gi else None)
            bias_f = bl
            if ind:
                fwd, bias_f = _image_affine(wl, aff, wr, aff, bl, ind_a=True)      # [wl diag(s) | wl t | wr diag(s)], bl + wr t
            elif aff is not None:
                fwd, bias_f = _image_affine(wl, None, wr, aff, bl)         # [wl | wr diag(s)], bl + wr t
            y = _gemm_nt(agg, fwd, x, bias_f, n_out, level_out=True)
            ctx.save_for_backward(x, wl, wr, bl, agg)
        ctx.graph, ctx.mean, ctx.first, ctx.has_bias, ctx.aff = graph, mean, first, bl is not None, aff
        ctx.ind = (not first) and ind
        return y

    @staticmethod
    def backward(ctx, g):
        graph, mean = ctx.graph, ctx.mean
        n = graph.num_nodes
        g = as_padded(g)
        need = ctx.needs_input_grad
        want_bias = ctx.has_bias and need[2]
        inv = graph.inv_deg if mean else None
        gx = None
        if ctx.first:
            x, wl, wr, bl = ctx.saved_tensors
            n_out, n_in = wl.shape
            gt = new_padded(n, n_out, g.device, zero_pad=False)      # gradient of the aggregated half
            _agg_launch(graph.by_src, n, g, n_out, 0, None, None, inv, None, gt)
            if need[1] and need[3] and _tn_gg and 2 * n_out <= MAX_GEMM_N:
                gwl, gwr, gb = _wgrad_gg(gt, g, x, wl, wr, ctx.aff, bl, want_bias)      # one pass over x for both
            else:
                (gwl,), _ = _wgrad(gt, [(x, wl, ctx.aff, need[1])], None, False)
                (gwr,), gb = _wgrad(g, [(x, wr, ctx.aff, need[3])], bl, want_bias)
            if ctx.aff is not None and len(ctx.aff) > 2 and ctx.aff[2] is not None:
                ctx.aff[2].uses += 1                      # one consumer, two parts (g_x = aggT(g) wl + g wr)
            if need[0]:
                gx = _grad_input(gt, ctx.pt, g, n_in, ctx.aff)
        else:
            x, wl, wr, bl, agg = ctx.saved_tensors
            n_out, n_in = wl.shape
            if ctx.ind:
                # operands [mean(d) | indicator] and d: raw products, then  raw * s + c t^T  with c = the indicator column
                # of the first product (the sum of g over the rows that have neighbours) / the column sums of g
                ra, rb, colsum = _gemm_tn2(g, agg, x, True)
                # g_x = aggT(g) wl + g wr, and aggT(g)^T d = g^T mean_agg(d), colsum(aggT(g)) = g^T indicator
                _contribute(ctx.aff, wl, ra[:, :n_in], ra[:, n_in])
                _contribute(ctx.aff, wr, rb, colsum)
                if ctx.aff[2] is not None:
                    ctx.aff[2].uses += 1                  # one consumer, two parts
                bufs = [_grad_buffer(wl), _grad_buffer(wr), _grad_buffer(bl) if want_bias else None]
                direct = bufs[0] is not None and bufs[1] is not None and (not want_bias or bufs[2] is not None)
                gwl, gwr = (bufs[0], bufs[1]) if direct else (torch.empty_like(wl), torch.empty_like(wr))
                jobs = [(gwl, ra[:, :n_in], ctx.aff, ra[:, n_in]), (gwr, rb, ctx.aff, colsum)]
                gb = None
                if want_bias:
                    gb = bufs[2] if direct else torch.empty_like(colsum)
                    jobs.append((gb.view(-1, 1), colsum.view(-1, 1), None))
                _wgrad_finish(jobs, colsum, accumulate=direct)
                if direct:
                    gwl = gwr = gb = None
            else:
                # agg is the aggregate of the normalised operand itself: only the root operand x may be a stored d
                _taint(ctx.aff)                           # ... and g^T agg(out) says nothing about g^T d
                (gwl, gwr), gb = _wgrad(g, [(agg, wl, None, need[1]), (x, wr, ctx.aff, need[3])], bl, want_bias)
            if need[0] and ctx.pt is not None:
                gt = new_padded(n, n_out, g.device, zero_pad=False)
                _agg_launch(graph.by_src, n, g, n_out, 0, None, None, inv, None, gt)
                gx = _grad_input(gt, ctx.pt, g, n_in, ctx.aff if ctx.ind else None)
            elif need[0]:
                t = _gemm_nt(g, ctx.plt, None, None, n_in)
                gr = _gemm_nt(g, ctx.prt, None, None, n_in)
                gx = new_padded(n, n_in, g.device, zero_pad=False)
                _agg_launch(graph.by_src, n, t, n_in, 0, None, None, inv, gr, gx)
        return gx, gwl, gb, gwr, None, None, None


class _Fork(torch.autograd.Function):
    """Two handles on one tensor for two consumers.  Autograd alone would add the two incoming
    gradients into a dense [N, F] tensor, which every kernel downstream then has to re-pad (a copy)
    or read with unaligned rows; here the sum is written in the padded layout directly."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        if ga.dim() != 2 or not ga.is_cuda:
            return ga + gb
        out = new_padded(ga.size(0), ga.size(1), ga.device)
        torch.add(ga, gb, out=out)
        return out


def fork(x: torch.Tensor):
    """(x, x) for a tensor with two consumers; see :class:`_Fork`."""
    x = plain(x)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return _Fork.apply(x)


def sage_conv(x, graph: Graph, wl, bl, wr, mean: bool = True):
    if x.size(0) != graph.num_nodes:
        raise _lib.HgvError(f"x has {x.size(0)} rows but the graph has {graph.num_nodes} nodes")
    n_out, n_in = wl.shape
    first = n_out < n_in and 2 * n_out <= MAX_GEMM_N
    ok = mean and (_affine_ok(2 * n_out, n_in, 0) if first else _affine_ok(n_out, n_in, n_in))
    x, aff = _unwrap(x, ok)
    return _SageConv.apply(x, wl, bl, wr, graph, mean, aff)


def dual_linear(xa, wa, xb=None, wb=None, bias=None):
    ok = _affine_ok(wa.size(0), wa.size(1), wb.size(1) if (xb is not None and wb is not None) else 0)
    (xa, aff_a), (xb, aff_b) = _unwrap(xa, ok), (_unwrap(xb, ok) if xb is not None else (None, None))
    return _DualLinear.apply(xa, wa, xb, wb, bias, aff_a, aff_b)


# --------------------------------------------------------------------------------------
# (relu) -> dropout -> batch norm
# --------------------------------------------------------------------------------------
class _BnAct(torch.autograd.Function):
    """(relu) -> dropout -> batch norm.  With ``fork`` the output comes back twice (two handles on one buffer) for a
    tensor with two consumers, and the backward pass takes both incoming gradients: the kernels add them on the fly,
    so the sum is never written (autograd alone would add them into a dense tensor with one more pass over memory)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, p, mask, seed, offset, ctr, eps, momentum,
                tracked=None, fork=False):
        _lib.require_cuda(y)
        y = rowmajor(y)
        n, h = y.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        # the 16-byte kernels write zeros into the pad columns themselves; the scalar ones (taken when y is
        # not aligned / padded) touch columns < h only, so the pads are cleared here: consumers read them
        out = new_padded(n, h, y.device, zero_pad=not is_vec4(y))
        save_mean = torch.empty(h, dtype=torch.float32, device=y.device)
        save_rstd = torch.empty(h, dtype=torch.float32, device=y.device)
        _lib.call("hgv_bn_act_fwd", y.data_ptr(), ld_of(y), n, h, int(relu), float(p), _lib.ptr(mask), seed, offset, ctr,
                  _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(running_mean), _lib.ptr(running_var),
                  float(momentum), _lib.ptr(tracked), save_mean.data_ptr(), save_rstd.data_ptr(), out.data_ptr(),
                  ld_of(out), ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.cfg = (relu, p, seed, offset, ctr)
        ctx.save_for_backward(y, gamma, save_mean, save_rstd, mask, beta)
        if fork:
            ctx.set_materialize_grads(False)
            return out, out.view_as(out)
        return out

    @staticmethod
    def backward(ctx, g, g2=None):
        y, gamma, save_mean, save_rstd, mask, beta = ctx.saved_tensors
        relu, p, seed, offset, ctr = ctx.cfg
        if g is None:
            g, g2 = g2, None
        n, h = y.shape
        if g is None:
            return (None,) * 15
        g = rowmajor(g)
        g2 = rowmajor(g2) if g2 is not None else None
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        vec = is_vec4(y) and is_vec4(g) and (g2 is None or is_vec4(g2))
        dy = new_padded(n, h, y.device, zero_pad=not vec)
        need = ctx.needs_input_grad
        # affine gradients: straight into the parameters' .grad buffers when that mode is on
        gbuf, bbuf = (_grad_buffer(gamma) if need[1] else None), (_grad_buffer(beta) if need[2] else None)
        direct = gbuf is not None and bbuf is not None
        dgamma = gbuf if direct else torch.empty(h, dtype=torch.float32, device=y.device)
        dbeta = bbuf if direct else torch.empty(h, dtype=torch.float32, device=y.device)
        _lib.call("hgv_bn_act_bwd", g.data_ptr(), ld_of(g), _lib.ptr(g2), ld_of(g2) if g2 is not None else 0, y.data_ptr(),
                  ld_of(y), n, h, int(relu), float(p),
                  _lib.ptr(mask), seed, offset, ctr, _lib.ptr(gamma), save_mean.data_ptr(), save_rstd.data_ptr(),
                  dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(), 1 if direct else 0, ws.data_ptr(),
                  ws.numel(), _lib.stream())
        if direct:
            dgamma = dbeta = None
        return (dy, dgamma if need[1] else None, dbeta if need[2] else None) + (None,) * 12


class _BnActPair(torch.autograd.Function):
    """Two dropout draws of one pre-activation, (relu) -> dropout -> batch norm each, over one read of it: what two
    :class:`_BnAct` calls on the same ``y`` compute (same statistics, same running-buffer updates in order, bit for bit),
    with the two input gradients added inside the backward kernel.  Every output comes back as a pair of handles."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, p, seed, offset, offset2, ctr, eps, momentum, tracked):
        n, h = y.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training")
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, h), y.device, "bn")
        out, out2 = new_padded(n, h, y.device, zero_pad=False), new_padded(n, h, y.device, zero_pad=False)
        stats = torch.empty(2, 2, h, dtype=torch.float32, device=y.device)          # [mean | rstd][draw][h]
        _lib.call("hgv_bn_act_fwd2", y.data_ptr(), ld_of(y), n, h, int(relu), float(p), seed, offset, offset2, ctr,
                  _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(running_mean), _lib.ptr(running_var),
                  float(momentum), _lib.ptr(tracked), stats[0].data_ptr(), stats[1].data_ptr(), out.data_ptr(),
                  out2.data_ptr(), ld_of(out), ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.cfg = (relu, p, seed, offset, offset2, ctr)
        ctx.save_for_backward(y, gamma, stats, beta)
        ctx.set_materialize_grads(False)
        return out, out.view_as(out), out2, out2.view_as(out2)

    @staticmethod
    def backward(ctx, g, g2, h, h2):
        y, gamma, stats, beta = ctx.saved_tensors
        relu, p, seed, offset, offset2, ctr = ctx.cfg
        if g is None:
            g, g2 = g2, None
        if h is None:
            h, h2 = h2, None
        if g is None and h is None:
            return (None,) * 14
        n, hh = y.shape
        lib = _lib.load()
        ws = _workspace(lib.hgv_bn_act_ws(n, hh), y.device, "bn")
        need = ctx.needs_input_grad
        gbuf, bbuf = (_grad_buffer(gamma) if need[1] else None), (_grad_buffer(beta) if need[2] else None)
        direct = gbuf is not None and bbuf is not None
        dgamma = gbuf if direct else torch.empty(hh, dtype=torch.float32, device=y.device)
        dbeta = bbuf if direct else torch.empty(hh, dtype=torch.float32, device=y.device)
        dy = new_padded(n, hh, y.device, zero_pad=False)
        grads = [as_padded(rowmajor(t)) if t is not None else None for t in (g, g2, h, h2)]
        if grads[0] is None or grads[2] is None:
            # only one of the two draws reached the loss: the single-draw kernels on that draw
            first = grads[0] is not None
            a, a2 = (grads[0], grads[1]) if first else (grads[2], grads[3])
            k = 0 if first else 1
            _lib.call("hgv_bn_act_bwd", a.data_ptr(), ld_of(a), _lib.ptr(a2), ld_of(a2) if a2 is not None else 0, y.data_ptr(),
                      ld_of(y), n, hh, int(relu), float(p), None, seed, offset if first else offset2, ctr, _lib.ptr(gamma),
                      stats[0, k].data_ptr(), stats[1, k].data_ptr(), dy.data_ptr(), ld_of(dy), dgamma.data_ptr(),
                      dbeta.data_ptr(), 1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
        else:
            a, a2, b, b2 = grads
            _lib.call("hgv_bn_act_bwd2", a.data_ptr(), ld_of(a), _lib.ptr(a2), ld_of(a2) if a2 is not None else 0,
                      b.data_ptr(), ld_of(b), _lib.ptr(b2), ld_of(b2) if b2 is not None else 0, y.data_ptr(), ld_of(y), n, hh,
                      int(relu), float(p), seed, offset, offset2, ctr, _lib.ptr(gamma), stats[0].data_ptr(),
                      stats[1].data_ptr(), dy.data_ptr(), ld_of(dy), dgamma.data_ptr(), dbeta.data_ptr(),
                      1 if direct else 0, ws.data_ptr(), ws.numel(), _lib.stream())
        if direct:
            dgamma = dbeta = None
        return (dy, dgamma if need[1] else None, dbeta if need[2] else None) + (None,) * 11


_pair_epilogues = os.environ.get("HGVAE_PAIR_EPILOGUES", "1") != "0"      # 0: always two separate calls (A/B measurements)


def bn_act_pair(y: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool, p: float):
    """Training-mode ``bn_act(y, ..., fork=True)`` for two successive dropout draws of the same ``y`` (the two encode
    passes of a train step over the shared first convolution): ``((out_a, out_a'), (out_b, out_b'))``, or None when
    the fused kernels do not apply (explicit noise tapes, p = 0, unaligned rows, no gradient) and the caller makes
    two ordinary calls."""
    _lib.require_cuda(y)
    if not (_pair_epilogues and p > 0.0 and torch.is_grad_enabled() and y.requires_grad and y.dim() == 2 and y.size(1) <= 512 and is_vec4(y)):
        return None
    noise = current_noise()
    mask, seed, offset, ctr = noise.dropout(y.shape, y.device)
    if mask is not None:
        raise _lib.HgvError("bn_act_pair: explicit dropout masks are consumed one call at a time; check noise_is_counter_based() first")
    _, _, offset2, _ = noise.dropout(y.shape, y.device)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    a, a2, b, b2 = _BnActPair.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, p, seed, offset, offset2,
                                    ctr, bn.eps, momentum, bn.num_batches_tracked)
    return (a, a2), (b, b2)


def noise_is_counter_based() -> bool:
    """True when the current noise source derives masks from (seed, offset) inside the kernels (no explicit tape)."""
    from .noise import PhiloxNoise
    return isinstance(current_noise(), PhiloxNoise)


class _BnActEval(torch.autograd.Function):
    """Evaluation mode: relu? -> identity -> batch norm with the running statistics, an elementwise affine map.  Its
    backward pass is the rare case (fine-tuning with frozen statistics, input gradients of a trained model) and is
    written with torch's elementwise device operators."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, relu, eps):
        y = rowmajor(y)
        out = new_padded(y.size(0), y.size(1), y.device, zero_pad=not is_vec4(y))
        _lib.call("hgv_bn_act_eval", y.data_ptr(), ld_of(y), y.size(0), y.size(1), int(relu), _lib.ptr(gamma),
                  _lib.ptr(beta), float(eps), running_mean.data_ptr(), running_var.data_ptr(),
                  out.data_ptr(), ld_of(out), _lib.stream())
        ctx.relu, ctx.eps, ctx.has_gamma, ctx.has_beta = bool(relu), float(eps), gamma is not None, beta is not None
        ctx.save_for_backward(y, gamma, running_mean, running_var)
        return out

    @staticmethod
    def backward(ctx, g):
        y, gamma, rm, rv = ctx.saved_tensors
        need = ctx.needs_input_grad
        rstd = torch.rsqrt(rv + ctx.eps)
        a = torch.relu(y) if ctx.relu else y
        gy = dgamma = dbeta = None
        if need[0]:
            gy = g * (rstd * gamma if ctx.has_gamma else rstd)
            if ctx.relu:
                gy = gy * (y > 0)
        if ctx.has_gamma and need[1]:
            dgamma = (g * ((a - rm) * rstd)).sum(0)
        if ctx.has_beta and need[2]:
            dbeta = g.sum(0)
        return gy, dgamma, dbeta, None, None, None, None


def bn_act(y: torch.Tensor, bn: torch.nn.BatchNorm1d, relu: bool, p: float, training: bool, fork: bool = False, noise=None):
    """relu? -> Dropout(p) -> BatchNorm1d, fused.  ``bn`` supplies affine parameters and running buffers.  With ``fork``
    the result is returned as a pair of handles for its two consumers (see :class:`_BnAct`).  ``noise``: a dropout draw
    ``(mask, seed, offset, counter)`` the caller has already taken for this call."""
    y = plain(y)
    _lib.require_cuda(y)
    if fork and not (training and torch.is_grad_enabled() and y.requires_grad):
        out = bn_act(y, bn, relu, p, training, noise=noise)
        return out, out
    if not training:
        return _BnActEval.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, bn.eps)
    mask, seed, offset, ctr = None, 0, 0, None
    if p > 0.0:
        mask, seed, offset, ctr = noise if noise is not None else current_noise().dropout(y.shape, y.device)
    momentum = bn.momentum if bn.momentum is not None else 0.1
    if on_branch_stream():
        # The trunk may be updating this module's running statistics at this moment (the other encode pass).  The
        # kernel updates a zeroed stand-in instead - which leaves momentum * statistic there - and the real buffers
        # take (1 - momentum) * buffer + stand-in after the branch has joined: the bits of two updates in program order.
        stand_in = torch.zeros(2, y.size(1), dtype=torch.float32, device=y.device)
        _branch_updates.append((bn, stand_in, momentum))
        return _BnAct.apply(y, bn.weight, bn.bias, stand_in[0], stand_in[1], relu, p, mask, seed, offset, ctr,
                            bn.eps, momentum, None, fork)
    # num_batches_tracked is incremented by the statistics kernel
    return _BnAct.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, relu, p, mask, seed, offset, ctr,
                        bn.eps, momentum, bn.num_batches_tracked, fork)


# --------------------------------------------------------------------------------------
# reparameterisation and losses
# --------------------------------------------------------------------------------------
class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ml, z_dim, n_samples, eps, seed, offset, ctr):
        ml = rowmajor(ml)
        n = ml.size(0)
        z = new_padded(n, z_dim, ml.device)
        eps_bar = torch.empty(n, z_dim, dtype=torch.float32, device=ml.device)
        _lib.call("hgv_reparam_fwd", ml.data_ptr(), ld_of(ml), n, z_dim, n_samples, _lib.ptr(eps), seed, offset, ctr,
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
        return gml, None, None, None, None, None, None


def reparameterize(ml: torch.Tensor, z_dim: int, n_samples: int) -> torch.Tensor:
    """z = mean over n_samples of (mean + exp(log_std) * eps_k); ml = [mean | log_std]."""
    _lib.require_cuda(ml)
    eps, seed, offset, ctr = current_noise().normal((n_samples, ml.size(0), z_dim), ml.device)
    return _Reparam.apply(ml, z_dim, n_samples, eps, seed, offset, ctr)


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


_mse_early = os.environ.get("HGVAE_MSE_EARLY", "1") != "0"      # 0: the gradient of the mean squared error is formed in the backward pass (A/B)


class _MSE(torch.autograd.Function):
    """mean((r - x)^2).  When r needs a gradient the forward pass writes it as well, for an upstream gradient of one (the
    loss enters the total with weight one, gvae/gvae_skipcon.py:176): the pass reads r and x anyway, and the backward pass
    then only has to look at the upstream gradient on the device - a second pass over r and x runs when it is not 1."""

    @staticmethod
    def forward(ctx, r, x):
        r, x = rowmajor(r), rowmajor(x)
        out = torch.empty((), dtype=torch.float32, device=r.device)
        ws = _workspace(_lib.load().hgv_loss_ws(), r.device, "loss")
        ctx.gr, ctx.calls = None, 0
        if _mse_early and ctx.needs_input_grad[0] and r.size(0) >= max(2, DEFER_MIN_ROWS):
            ctx.gr = new_padded(r.size(0), r.size(1), r.device)
            _lib.call("hgv_mse_fwd_grad", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1), out.data_ptr(),
                      ctx.gr.data_ptr(), ld_of(ctx.gr), ws.data_ptr(), ws.numel(), _lib.stream())
        else:
            _lib.call("hgv_mse_fwd", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1),
                      out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream())
        ctx.save_for_backward(r, x)
        return out

    @staticmethod
    def backward(ctx, g):
        r, x = ctx.saved_tensors
        g = g.contiguous().to(torch.float32)
        first = ctx.gr is not None and ctx.calls == 0
        ctx.calls += 1
        if first:
            gr = ctx.gr          # holds the gradient for g == 1; rewritten on the device when g is anything else
            _lib.call("hgv_mse_bwd_unless_unit", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1), g.data_ptr(),
                      gr.data_ptr(), ld_of(gr), _lib.stream())
        else:
            gr = new_padded(r.size(0), r.size(1), r.device)
            _lib.call("hgv_mse_bwd", r.data_ptr(), ld_of(r), x.data_ptr(), ld_of(x), r.size(0), r.size(1), g.data_ptr(),
                      gr.data_ptr(), ld_of(gr), _lib.stream())
        # the target's gradient is the negative of the reconstruction's (only asked for when the input itself needs one)
        return (gr if ctx.needs_input_grad[0] else None), (-gr if ctx.needs_input_grad[1] else None)


def mse_loss(recon: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    recon, x = plain(recon), plain(x)
    _lib.require_cuda(recon, x)
    if recon.shape != x.shape:
        raise _lib.HgvError(f"mse_loss: shapes differ {tuple(recon.shape)} vs {tuple(x.shape)}")
    return _MSE.apply(recon, x)


# --------------------------------------------------------------------------------------
# single-head graph attention (edge softmax + weighted aggregation)
# --------------------------------------------------------------------------------------
_gat_long = os.environ.get("HGVAE_GAT_LONG", "1") != "0"        # 0: hub rows stay with their lane group (A/B measurements)


class _GatAggregate(torch.autograd.Function):
    """out_i = sum_k dropout(softmax_i(leaky_relu(a_src[col_k] + a_dst[i])))_k * xw[col_k] (+ bias)

    over the loop-augmented graph ``lg`` with the logits a_src = xw att[0], a_dst = xw att[1] (``att`` [2, C]: GATConv's
    att_src / att_dst).  One projection (both logit columns at once) and one attention kernel forward; backward: one
    kernel per CSR direction - the attention weights are recomputed, never stored, and the gradient that reaches xw
    through the logits is added inside the second kernel - plus the [2, C] weight gradient of ``att``.
    """

    @staticmethod
    def forward(ctx, xw, att, bias, lg: Graph, slope, p, mask, seed, offset, ctr):
        _lib.require_cuda(xw, att)
        xw = as_padded(xw)
        n, c = lg.num_nodes, xw.size(1)
        if tuple(att.shape) != (2, c):
            raise _lib.HgvError(f"attention vectors of shape {tuple(att.shape)} for features of width {c}")
        csr = lg.by_dst
        dev = xw.device
        ab = _gemm_nt(xw, _image(att, None, False), None, None, 2)            # [N, 2]: a_src | a_dst, leading dimension 4
        row_max = torch.empty(n, dtype=torch.float32, device=dev)
        row_den = torch.empty(n, dtype=torch.float32, device=dev)
        out = new_padded(n, c, dev, zero_pad=False)
        lda = ld_of(ab)
        lr = csr.long_rows(n)          # hub rows go to a workgroup each
        _lib.call("hgv_gat_fwd_long" if _gat_long else "hgv_gat_fwd", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.perm.data_ptr(), n, xw.data_ptr(), ld_of(xw),
                  c, ab.data_ptr(), lda, ab.data_ptr() + 4, lda, float(slope), float(p),
                  _lib.ptr(mask), seed, offset, ctr, _lib.ptr(bias), out.data_ptr(), ld_of(out), row_max.data_ptr(),
                  row_den.data_ptr(), *((lr.threshold, lr.rows.data_ptr(), lr.count.data_ptr(), lr.cap, 1 if lr.maybe_any() else 0)
                                        if _gat_long else ()), _lib.stream())
        ctx.lg, ctx.cfg = lg, (slope, p, seed, offset, ctr)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(xw, att, ab, row_max, row_den, mask)
        return out

    @staticmethod
    def backward(ctx, g):
        xw, att, ab, row_max, row_den, mask = ctx.saved_tensors
        lg = ctx.lg
        slope, p, seed, offset, ctr = ctx.cfg
        n, c = lg.num_nodes, xw.size(1)
        g = as_padded(g)
        csr, csr_t = lg.by_dst, lg.by_src
        dev = xw.device
        scratch = torch.empty(2 * n, dtype=torch.float32, device=dev)
        da = new_padded(n, 2, dev)                                     # d a_src | d a_dst
        att_p = new_padded(2, c, dev)                                  # rows padded to 16 bytes for the kernel's chunk loads
        att_p.copy_(att.detach())
        gxw = new_padded(n, c, dev, zero_pad=False)
        lda, ldd = ld_of(ab), ld_of(da)
        lr, lr_t = csr.long_rows(n), csr_t.long_rows(n)
        _lib.call("hgv_gat_bwd_long" if _gat_long else "hgv_gat_bwd", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.perm.data_ptr(), csr_t.rowptr.data_ptr(),
                  csr_t.col.data_ptr(), csr_t.perm.data_ptr(), n, g.data_ptr(), ld_of(g), xw.data_ptr(), ld_of(xw), c,
                  ab.data_ptr(), lda, ab.data_ptr() + 4, lda, float(slope), float(p),
                  _lib.ptr(mask), seed, offset, ctr, row_max.data_ptr(), row_den.data_ptr(), scratch.data_ptr(),
                  gxw.data_ptr(), ld_of(gxw), da.data_ptr(), da.data_ptr() + 4, ldd, att_p.data_ptr(),
                  att_p.data_ptr() + 4 * ld_of(att_p),
                  *((lr.threshold, lr.rows.data_ptr(), lr.count.data_ptr(), lr.cap, 1 if lr.maybe_any() else 0,
                     lr_t.threshold, lr_t.rows.data_ptr(), lr_t.count.data_ptr(), lr_t.cap, 1 if lr_t.maybe_any() else 0)
                    if _gat_long else ()), _lib.stream())
        gatt = None
        if ctx.needs_input_grad[1]:
            gatt, _ = _gemm_tn(da, xw, False)                          # [2, C] = [da_src | da_dst]^T xw
        gbias = g.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gxw, gatt, gbias, None, None, None, None, None, None, None


def gat_aggregate(xw, att, loop_graph: Graph, slope: float, p: float, training: bool, bias=None):
    """Attention-weighted neighbour sum of GATConv (+ bias) from the projected features ``xw`` [N, C] and the stacked
    attention vectors ``att`` [2, C] (att_src; att_dst)."""
    xw = plain(xw)
    mask, seed, offset, ctr = None, 0, 0, None
    if training and p > 0.0:
        mask, seed, offset, ctr = current_noise().dropout((loop_graph.num_edges, 1), xw.device)
        if mask is not None:
            mask = mask.view(-1)
    else:
        p = 0.0
    return _GatAggregate.apply(xw, att, bias, loop_graph, slope, p, mask, seed, offset, ctr)


# --------------------------------------------------------------------------------------
# pooling helpers
# --------------------------------------------------------------------------------------
class _BatchPlan:
    """Sizes that depend only on the batch vector and the pooling ratio: graph boundaries, kept counts and their
    prefix sums (device arrays) and the host-side totals.  Building one costs a host synchronisation; plans are
    cached per batch tensor object, so a training loop that reuses its batches (or replays a captured step) pays
    it once."""
    __slots__ = ("num_graphs", "ptr", "k", "kptr", "total", "max_nodes")

    def __init__(self, batch: torch.Tensor, ratio: float, dtype, num_graphs: Optional[int]):
        n = batch.numel()
        if num_graphs is None:
            num_graphs = int(batch.max().item()) + 1 if n else 0
        n_g = torch.bincount(batch, minlength=num_graphs)
        k_g = (float(ratio) * n_g.to(dtype)).ceil().to(torch.long)
        self.num_graphs = num_graphs
        self.ptr = torch.zeros(num_graphs + 1, dtype=torch.long, device=batch.device)
        self.ptr[1:] = torch.cumsum(n_g, 0)
        self.k = k_g
        self.kptr = torch.zeros(num_graphs + 1, dtype=torch.long, device=batch.device)
        self.kptr[1:] = torch.cumsum(k_g, 0)
        sizes = torch.stack([self.kptr[-1], n_g.max() if num_graphs else self.kptr[-1]]).tolist()   # one sync
        self.total, self.max_nodes = int(sizes[0]), int(sizes[1])


_plan_cache: "dict[tuple, tuple]" = {}


def batch_plan(batch: torch.Tensor, ratio: float, dtype=torch.float32, num_graphs: Optional[int] = None) -> _BatchPlan:
    import weakref
    key = (id(batch), float(ratio), dtype)
    hit = _plan_cache.get(key)
    if hit is not None and hit[0]() is batch and hit[1] == batch._version:
        return hit[2]
    plan = _BatchPlan(batch, ratio, dtype, num_graphs)
    _plan_cache[key] = (weakref.ref(batch, lambda _r, k=key: _plan_cache.pop(k, None)), batch._version, plan)
    return plan


def topk_perm(score: torch.Tensor, ratio: float, batch: torch.Tensor, num_graphs: Optional[int] = None,
              global_sort: bool = False):
    """Per-graph top ceil(ratio * n_g) nodes by descending score (ties: ascending node id).  Graphs of up to 16384
    nodes are sorted by one workgroup each in LDS; larger ones (or ``global_sort``) by radix passes over all scores."""
    _lib.require_cuda(score, batch)
    n = score.numel()
    plan = batch_plan(batch, ratio, score.dtype, num_graphs)
    perm = torch.empty(plan.total, dtype=torch.long, device=score.device)
    lib = _lib.load()
    segmented = 0 < plan.max_nodes <= 16384 and not global_sort
    ws = None if segmented else _workspace(lib.hgv_topk_ws(n), score.device, "topk")
    score = score.detach().contiguous().to(torch.float32)
    _lib.call("hgv_topk_perm", score.data_ptr(), batch.data_ptr(), plan.ptr.data_ptr(), plan.k.data_ptr(),
              plan.kptr.data_ptr(), n, plan.num_graphs, plan.max_nodes if segmented else 0, perm.data_ptr(), _lib.ptr(ws),
              ws.numel() if ws is not None else 0, _lib.stream())
    return perm


_bounded_pooling = False


@contextlib.contextmanager
def bounded_pooling(enabled: bool = True):
    """Inside this context ``filter_adj`` does not read the number of surviving edges back to the host: it returns
    the coarse ``edge_index`` at its capacity ``[2, E]`` (surviving edges first, in order; the rest zeros) with the
    device-side count attached as ``.hgv_count``.  For callers that discard or only forward the pooled graph (the
    reference's train loops do, autoencoder/vae_with_pool.py:154): no host synchronisation, so the step can be
    captured into a HIP graph."""
    global _bounded_pooling
    prev, _bounded_pooling = _bounded_pooling, bool(enabled)
    try:
        yield
    finally:
        _bounded_pooling = prev


def filter_adj(edge_index: torch.Tensor, edge_attr, perm: torch.Tensor, num_nodes: int):
    """Edges with both ends in ``perm``, relabelled to positions in ``perm``, original order."""
    _lib.require_cuda(edge_index, perm)
    e = edge_index.size(1)
    dev = edge_index.device
    bounded = _bounded_pooling
    out = torch.empty(2, max(e, 1), dtype=torch.long, device=dev)
    kept = torch.empty(max(e, 1), dtype=torch.int32, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws = _workspace(lib.hgv_filter_adj_ws(e, num_nodes), dev, "filter_adj")
    _lib.call("hgv_filter_adj", edge_index.data_ptr(), edge_index.stride(0), edge_index.stride(1), e, num_nodes,
              perm.data_ptr(), perm.numel(), out.data_ptr(), out.stride(0), kept.data_ptr(), count.data_ptr(),
              1 if bounded else 0, ws.data_ptr(), ws.numel(), _lib.stream())
    if bounded:
        ei = out[:, :e]
        ei.hgv_count = count
        ea = edge_attr.index_select(0, kept[:e].long()) if edge_attr is not None else None
        return ei, ea
    m = int(count.item())  # the output size is data-dependent: one host sync, as in PyG
    ei = out[:, :m]
    ea = edge_attr.index_select(0, kept[:m].long()) if edge_attr is not None else None
    return ei, ea


class _GcnAggregate(torch.autograd.Function):
    """out_i = dis_i * sum_{j -> i} dis_j * h_j  (the symmetric normalisation of GCNConv as neighbour / row weights of
    the aggregation kernel; backward = the same reduction over the transposed CSR)."""

    @staticmethod
    def forward(ctx, h, graph: Graph, dis):
        h = as_padded(h)
        out = new_padded(graph.num_nodes, h.size(1), h.device, zero_pad=False)
        _agg_launch(graph.by_dst, graph.num_nodes, h, h.size(1), 2, dis, None, dis, None, out)
        ctx.graph = graph
        ctx.save_for_backward(dis)
        return out

    @staticmethod
    def backward(ctx, g):
        (dis,) = ctx.saved_tensors
        graph = ctx.graph
        g = as_padded(g)
        gh = new_padded(graph.num_nodes, g.size(1), g.device, zero_pad=False)
        _agg_launch(graph.by_src, graph.num_nodes, g, g.size(1), 2, dis, None, dis, None, gh)
        return gh, None, None


def gcn_aggregate(h: torch.Tensor, loop_graph: Graph) -> torch.Tensor:
    """D^-1/2 (A + I) D^-1/2 h over the loop-augmented graph (degrees over targets, self loops included)."""
    h = plain(h)
    _lib.require_cuda(h)
    return _GcnAggregate.apply(h, loop_graph, loop_graph.gcn_dis)


class _GatherScale(torch.autograd.Function):
    """out[q] = x[perm[q]] * scale[perm[q]]  (scale optional)."""

    @staticmethod
    def forward(ctx, x, perm, scale):
        x = rowmajor(x)
        k, f = perm.numel(), x.size(1)
        out = new_padded(k, f, x.device)
        s = scale.contiguous() if scale is not None else None
        _lib.call("hgv_gather_rows", x.data_ptr(), ld_of(x), perm.data_ptr(), _lib.ptr(s), k, f, out.data_ptr(),
                  ld_of(out), _lib.stream())
        ctx.save_for_backward(x, perm, s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, perm, s = ctx.saved_tensors
        g = rowmajor(g)
        k, f = perm.numel(), x.size(1)
        need = ctx.needs_input_grad
        gx = torch.zeros(x.size(0), f, dtype=torch.float32, device=x.device) if need[0] else None
        gs = torch.zeros(x.size(0), dtype=torch.float32, device=x.device) if (s is not None and need[2]) else None
        _lib.call("hgv_gather_rows_bwd", g.data_ptr(), ld_of(g), x.data_ptr(), ld_of(x), perm.data_ptr(), _lib.ptr(s),
                  k, f, _lib.ptr(gx), f, _lib.ptr(gs), _lib.stream())
        return gx, None, gs


def gather_rows(x, perm, scale=None):
    x = plain(x)
    _lib.require_cuda(x, perm)
    return _GatherScale.apply(x, perm, scale)


def gather_i64(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """src[idx] for int64 vectors (batch[perm], perm[edge_index_cg])."""
    _lib.require_cuda(src, idx)
    src, flat = src.contiguous(), idx.contiguous().view(-1)
    out = torch.empty_like(flat)
    _lib.call("hgv_gather_i64", src.data_ptr(), flat.data_ptr(), flat.numel(), out.data_ptr(), _lib.stream())
    return out.view(idx.shape)
