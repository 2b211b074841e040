"""Device-resident graph structure (CSR by target, CSR by source) built once per batch.

PyG operators take a COO ``edge_index`` on every call (reference call sites
gvae/gvae_skipcon.py:105-118).  Here the first operator that sees an
``edge_index`` tensor builds a :class:`Graph` with the radix-sort kernel and every
later layer, both encode passes and the backward pass reuse it.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import _lib


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


def new_padded(rows: int, cols: int, device, zero_pad: bool = True) -> torch.Tensor:
    """[rows, cols] fp32 view of a buffer whose rows are padded to a multiple of 4 floats."""
    ld = pad4(cols)
    buf = torch.empty(rows, ld, dtype=torch.float32, device=device)
    if zero_pad and ld != cols:
        buf[:, cols:].zero_()
    return buf[:, :cols]


def ld_of(t: torch.Tensor) -> int:
    """Leading dimension (elements per row) of a 2-D tensor whose rows are unit-stride."""
    sh, st = t.shape, t.stride()
    if len(sh) != 2 or (sh[1] > 1 and st[1] != 1):
        raise _lib.HgvError(f"expected a row-major 2-D tensor, got shape {tuple(sh)} strides {st}")
    return st[0] if sh[0] > 1 else max(st[0], sh[1])


def rowmajor(t: torch.Tensor) -> torch.Tensor:
    """Return t if its rows are unit-stride with ld >= cols, else a padded copy."""
    if t.dim() == 2 and t.dtype == torch.float32 and (t.size(1) <= 1 or t.stride(1) == 1) \
            and (t.size(0) <= 1 or t.stride(0) >= t.size(1)):
        return t
    out = new_padded(t.size(0), t.size(1), t.device)
    out.copy_(t)
    return out


def is_vec4(t: torch.Tensor) -> bool:
    ld = ld_of(t)
    return t.data_ptr() % 16 == 0 and ld % 4 == 0 and ld >= pad4(t.size(1))


def as_padded(t: torch.Tensor) -> torch.Tensor:
    """Row-major fp32 tensor laid out for the 16-byte kernel paths (copy only if needed)."""
    t = rowmajor(t)
    if is_vec4(t):
        return t
    out = new_padded(t.size(0), t.size(1), t.device)
    out.copy_(t)
    return out


class _LongRows:
    """Rows of a CSR that are far longer than the average (hubs), listed on the device once per CSR.  Whether there
    are any reaches the host through an asynchronous copy; until it has arrived the answer is "maybe"."""
    __slots__ = ("threshold", "rows", "count", "cap", "_host", "_event", "_known")

    def __init__(self, rowptr: torch.Tensor, n_rows: int, n_entries: int):
        dev = rowptr.device
        self.threshold = max(64, 4 * -(-n_entries // max(n_rows, 1)))
        self.cap = n_entries // self.threshold + 1
        self.rows = torch.empty(self.cap, dtype=torch.int32, device=dev)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.call("hgv_csr_long_rows", rowptr.data_ptr(), n_rows, self.threshold, self.rows.data_ptr(), self.cap,
                  self.count.data_ptr(), _lib.stream())
        self._known: Optional[int] = None
        self._host = self._event = None
        if not torch.cuda.is_current_stream_capturing():      # a capture must not hold host copies or foreign events
            self._host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._host.copy_(self.count, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()

    def maybe_any(self) -> bool:
        """False only once the device has reported an empty list (never blocks)."""
        if self._known is None and self._event is not None and self._event.query():
            self._known = int(self._host[0])
        return self._known is None or self._known > 0


class _Csr:
    __slots__ = ("rowptr", "col", "perm", "_long")

    def __init__(self, rowptr, col, perm):
        self.rowptr, self.col, self.perm = rowptr, col, perm
        self._long = None

    def long_rows(self, n_rows: int) -> _LongRows:
        if self._long is None:
            self._long = _LongRows(self.rowptr, n_rows, int(self.col.numel()))
        return self._long


def _build_csr(edge_index: torch.Tensor, n: int, by_src: bool, status: torch.Tensor) -> _Csr:
    e = edge_index.size(1)
    dev = edge_index.device
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(e, dtype=torch.int32, device=dev)
    perm = torch.empty(e, dtype=torch.int32, device=dev)
    lib = _lib.load()
    ws_bytes = lib.hgv_csr_build_ws(e, n)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
    _lib.call("hgv_csr_build", edge_index.data_ptr(), edge_index.stride(0), edge_index.stride(1), e, n,
              1 if by_src else 0, rowptr.data_ptr(), col.data_ptr(), perm.data_ptr(), status.data_ptr(),
              ws.data_ptr(), ws.numel(), _lib.stream())
    return _Csr(rowptr, col, perm)


class Graph:
    """CSR views of one (batched) graph with ``num_nodes`` nodes.

    ``by_dst``: row i lists the sources of the edges into i, in stable edge order
    (forward aggregation).  ``by_src``: row j lists the targets of the edges out of
    j (backward aggregation); built on first use.
    """

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, hold: bool = True):
        _lib.require_cuda(edge_index)
        if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
            raise _lib.HgvError(f"edge_index must be int64 [2, E], got {edge_index.dtype} {tuple(edge_index.shape)}")
        self.num_nodes = int(num_nodes)
        self.num_edges = int(edge_index.size(1))
        # graph_of() keys its cache by the caller's tensor and passes hold=False: a strong
        # reference from the cached Graph would keep that cache entry alive forever.
        self._ei_ref = weakref.ref(edge_index)
        self._ei_hold = edge_index if hold else None
        self.status = torch.zeros(1, dtype=torch.int32, device=edge_index.device)
        self.by_dst = _build_csr(edge_index, self.num_nodes, False, self.status)
        self._by_src: Optional[_Csr] = None
        self._inv_deg = None
        self._loops: Optional["Graph"] = None
        self._t2f = None

    @classmethod
    def from_parts(cls, edge_index: torch.Tensor, num_nodes: int, by_dst: _Csr, by_src: Optional[_Csr]) -> "Graph":
        """Graph around CSR arrays that already exist (e.g. assembled from a packed shard): nothing is sorted."""
        g = cls.__new__(cls)
        g.num_nodes, g.num_edges = int(num_nodes), int(edge_index.size(1))
        g._ei_ref, g._ei_hold = weakref.ref(edge_index), edge_index
        g.status = torch.zeros(1, dtype=torch.int32, device=edge_index.device)
        g.by_dst, g._by_src = by_dst, by_src
        g._inv_deg = g._loops = g._t2f = None
        return g

    @property
    def edge_index(self) -> torch.Tensor:
        ei = self._ei_ref()
        if ei is None:
            raise _lib.HgvError("the edge_index tensor this Graph was built from no longer exists")
        return ei

    @property
    def by_src(self) -> _Csr:
        if self._by_src is None:
            self._by_src = _build_csr(self.edge_index, self.num_nodes, True, self.status)
        return self._by_src

    @property
    def in_degree(self) -> torch.Tensor:
        rp = self.by_dst.rowptr
        return rp[1:] - rp[:-1]

    @property
    def inv_deg(self) -> torch.Tensor:
        """1 / max(in-degree, 1) as fp32 [N]."""
        if self._inv_deg is None:
            self._inv_deg = self.in_degree.clamp(min=1).to(torch.float32).reciprocal()
        return self._inv_deg

    @property
    def gcn_dis(self) -> torch.Tensor:
        """in-degree^-1/2 as fp32 [N] (0 for isolated targets): GCNConv's symmetric normalisation, meant for the
        loop-augmented graph where every degree is >= 1."""
        if getattr(self, "_gcn_dis", None) is None:
            d = self.in_degree.to(torch.float32).pow(-0.5)
            self._gcn_dis = torch.where(torch.isinf(d), torch.zeros_like(d), d)
        return self._gcn_dis

    @property
    def src_pos_to_dst_pos(self) -> torch.Tensor:
        """For each position of the by-source CSR, the position of the same edge in the by-target CSR."""
        if self._t2f is None:
            inv = torch.empty(self.num_edges, dtype=torch.int64, device=self.by_dst.perm.device)
            inv[self.by_dst.perm.long()] = torch.arange(self.num_edges, device=inv.device)
            self._t2f = inv[self.by_src.perm.long()]
        return self._t2f

    def with_self_loops(self) -> "Graph":
        """Graph of (edges minus existing self loops) followed by one loop per node.

        This is the edge list GATConv (remove_self_loops + add_self_loops) and GCNConv
        (add_remaining_self_loops, unweighted) aggregate over.
        """
        if self._loops is None:
            ei = self.edge_index
            keep = ei[0] != ei[1]
            if not bool(keep.all()):
                ei = ei[:, keep]
            loop = torch.arange(self.num_nodes, dtype=torch.int64, device=ei.device)
            self._loops = Graph(torch.cat([ei, loop.unsqueeze(0).expand(2, -1)], dim=1).contiguous(), self.num_nodes,
                                hold=True)
        return self._loops

    def check(self) -> None:
        """Synchronising validity check: raises if an index was outside [0, num_nodes)."""
        if int(self.status.item()) != 0:
            raise _lib.HgvError("edge_index holds node indices outside [0, num_nodes)")


# edge_index tensor object -> Graph, dropped when the tensor dies
_cache: "dict[int, tuple]" = {}


def graph_of(edge_index, num_nodes: Optional[int] = None) -> Graph:
    """Graph for an ``edge_index`` tensor (cached per tensor object and version) or pass-through."""
    if isinstance(edge_index, Graph):
        return edge_index
    key = id(edge_index)
    hit = _cache.get(key)
    if hit is not None:
        ref, version, n, g = hit
        if ref() is edge_index and version == edge_index._version and (num_nodes is None or n == num_nodes):
            if torch.is_grad_enabled():
                g.by_src  # the caller's tensor may be gone by the time backward runs
            return g
    if num_nodes is None:
        raise _lib.HgvError("num_nodes is required to build a Graph")
    g = Graph(edge_index, num_nodes, hold=False)
    g.by_dst.long_rows(g.num_nodes)
    if torch.is_grad_enabled():
        # built now: the caller's tensor may be gone by the time backward runs - and everything the backward pass reads
        # exists before any of it can run on a second stream
        g.by_src.long_rows(g.num_nodes)
        g.inv_deg
    ref = weakref.ref(edge_index, lambda _r, k=key: _cache.pop(k, None))
    _cache[key] = (ref, edge_index._version, num_nodes, g)
    return g
