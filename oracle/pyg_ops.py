"""Pure-torch CPU restatement of the PyTorch Geometric operators the reference calls.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  PARITY UNPINNED: PyG is
absent, so every operator below restates the published PyG (~2.3) algorithm
built from ``index_select`` + ``scatter_add_`` exactly as PyG runs it on CPU
without ``torch_scatter``.

Reference call sites (relative to /root/reference):
  SAGEConv          gvae/gvae_skipcon.py:61,68,75,78,105,110,115,118
  GATConv           gvae/gvae_skipcon_attention.py:68,109
  SAGPooling        autoencoder/vae_with_pool.py:25,36,39  (GraphConv scorer)
  SAGPooling+GCN    autoencoder/vae_custom_pool.py:56-58   (GCNConv scorer)
  Batch             gvae/gvae_skipcon.py:262-263
  Data              gvae/gvae_skipcon.py:27
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


# ----------------------------------------------------------------------------
# scatter primitives (PyG torch_geometric.utils.scatter, pure-torch branch)
# ----------------------------------------------------------------------------
def scatter_sum(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """out[i] = sum of src[e] over e with index[e] == i (edge order)."""
    shape = (dim_size,) + tuple(src.shape[1:])
    out = src.new_zeros(shape)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return out.scatter_add_(0, idx, src)


def scatter_mean(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """sum / clamp(count, min=1) by true division (PyG reduce='mean')."""
    count = src.new_zeros(dim_size).scatter_add_(0, index, src.new_ones(index.numel()))
    count = count.clamp(min=1)
    out = scatter_sum(src, index, dim_size)
    return out / count.view(-1, *([1] * (src.dim() - 1)))


def scatter_max(src: Tensor, index: Tensor, dim_size: int) -> Tensor:
    shape = (dim_size,) + tuple(src.shape[1:])
    out = src.new_full(shape, float("-inf"))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)


def segment_softmax(e: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """PyG utils.softmax: max is detached, denominator gets +1e-16."""
    m = scatter_max(e.detach(), index, dim_size)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = (e - m.index_select(0, index)).exp()
    s = scatter_sum(p, index, dim_size) + 1e-16
    return p / s.index_select(0, index)


# ----------------------------------------------------------------------------
# edge-list utilities
# ----------------------------------------------------------------------------
def remove_self_loops(edge_index: Tensor) -> Tensor:
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep]


def add_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, loop.unsqueeze(0).repeat(2, 1)], dim=1)


def add_remaining_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    """Unweighted case: drop every (i, i) edge then append one loop per node."""
    return add_self_loops(remove_self_loops(edge_index), num_nodes)


def csr_reference(edge_index: Tensor, num_nodes: int, by_src: bool = False):
    """The contract the HIP csr_build kernel is bit-exact against.

    perm = stable argsort of the key row; col = other row permuted;
    rowptr = [0, cumsum(bincount(key))].
    """
    key = edge_index[0 if by_src else 1]
    other = edge_index[1 if by_src else 0]
    perm = torch.sort(key, stable=True).indices
    col = other[perm]
    deg = torch.bincount(key, minlength=num_nodes)
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    return rowptr.to(torch.int32), col.to(torch.int32), perm.to(torch.int32)


# ----------------------------------------------------------------------------
# noise tape: explicit dropout masks / normal draws consumed in call order
# ----------------------------------------------------------------------------
class NoiseTape:
    """Queue of pre-drawn noise tensors; ``None`` entries mean "draw with torch".

    Dropout layers pop a {0,1} float mask of the activation's shape, the
    reparameterisation pops an [n_samples, N, Z] normal tensor, GAT pops an
    [E', heads] mask for its attention dropout.
    """

    def __init__(self, items=None):
        self.items = list(items or [])
        self.pos = 0
        self.record: List[Tensor] = []

    def next(self):
        if self.pos >= len(self.items):
            return None
        t = self.items[self.pos]
        self.pos += 1
        return t


def dropout_with(x: Tensor, p: float, training: bool, tape: Optional[NoiseTape]) -> Tensor:
    """torch.nn.Dropout: x * (bernoulli(1-p) / (1-p)) in training, identity in eval."""
    if not training or p == 0.0:
        return x
    mask = tape.next() if tape is not None else None
    if mask is None:
        mask = torch.bernoulli(torch.full_like(x, 1.0 - p))
        if tape is not None:
            tape.record.append(mask)
    return x * (mask / (1.0 - p))


# ----------------------------------------------------------------------------
# convolutions
# ----------------------------------------------------------------------------
class SAGEConv(nn.Module):
    """PyG SAGEConv defaults: aggr='mean', root_weight=True, bias=True.

    out_i = lin_l(mean_{j->i} x_j) + lin_r(x_i); messages flow
    edge_index[0] (source j) -> edge_index[1] (target i).
    lin_l has the bias, lin_r has none.
    """

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        x_j = x.index_select(0, src)
        agg = scatter_mean(x_j, dst, x.size(0))
        return self.lin_l(agg) + self.lin_r(x)


class GraphConv(nn.Module):
    """PyG GraphConv (aggr='add'): lin_rel(sum_j x_j) + lin_root(x_i)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        src, dst = edge_index[0], edge_index[1]
        agg = scatter_sum(x.index_select(0, src), dst, x.size(0))
        return self.lin_rel(agg) + self.lin_root(x)


class GCNConv(nn.Module):
    """PyG GCNConv: D^-1/2 (A+I) D^-1/2 (x W^T) + b, degree over targets."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x: Tensor, edge_index: Tensor) -> Tensor:
        n = x.size(0)
        ei = add_remaining_self_loops(edge_index, n)
        row, col = ei[0], ei[1]
        w = x.new_ones(ei.size(1))
        deg = scatter_sum(w, col, n)
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
        norm = dis[row] * w * dis[col]
        h = self.lin(x)
        out = scatter_sum(norm.view(-1, 1) * h.index_select(0, row), col, n)
        return out + self.bias


class GATConv(nn.Module):
    """PyG GATConv with concat=True, negative_slope=0.2, add_self_loops=True.

    One shared linear map for sources and targets (no bias), attention
    vectors att_src / att_dst of shape [1, heads, C], bias of size heads*C.
    The attention dropout mask has shape [E', heads].
    """

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, dropout: float = 0.0):
        super().__init__()
        self.heads, self.out_channels, self.dropout = heads, out_channels, dropout
        self.negative_slope = 0.2
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        nn.init.xavier_uniform_(self.lin_src.weight)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def forward(self, x: Tensor, edge_index: Tensor, tape: Optional[NoiseTape] = None) -> Tensor:
        n, h, c = x.size(0), self.heads, self.out_channels
        xw = self.lin_src(x).view(n, h, c)
        a_s = (xw * self.att_src).sum(-1)
        a_d = (xw * self.att_dst).sum(-1)
        ei = add_self_loops(remove_self_loops(edge_index), n)
        src, dst = ei[0], ei[1]
        e = a_s.index_select(0, src) + a_d.index_select(0, dst)
        e = F.leaky_relu(e, self.negative_slope)
        alpha = segment_softmax(e, dst, n)
        alpha = dropout_with(alpha, self.dropout, self.training, tape)
        msg = xw.index_select(0, src) * alpha.unsqueeze(-1)
        out = scatter_sum(msg, dst, n).reshape(n, h * c)
        return out + self.bias


# ----------------------------------------------------------------------------
# pooling
# ----------------------------------------------------------------------------
def topk(score: Tensor, ratio: float, batch: Tensor) -> Tensor:
    """Per-graph top ceil(ratio*n_g) by descending score, graphs in order.

    PyG sorts a dense [B, n_max] table (fill -60000); ties there are broken by
    whatever ``sort`` does.  The restatement fixes ties to ascending node
    index (stable sort), which is what the HIP kernel implements.
    """
    num_graphs = int(batch.max()) + 1 if batch.numel() else 0
    n_g = torch.bincount(batch, minlength=num_graphs)
    k_g = (float(ratio) * n_g.to(score.dtype)).ceil().to(torch.long)
    cum = torch.zeros(num_graphs + 1, dtype=torch.long)
    cum[1:] = torch.cumsum(n_g, 0)
    out = []
    for g in range(num_graphs):
        seg = score[cum[g]:cum[g + 1]]
        order = torch.sort(seg, descending=True, stable=True).indices
        out.append(order[: int(k_g[g])] + cum[g])
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.long)


def filter_adj(edge_index: Tensor, edge_attr: Optional[Tensor], perm: Tensor, num_nodes: int):
    mask = perm.new_full((num_nodes,), -1)
    mask[perm] = torch.arange(perm.numel(), dtype=perm.dtype)
    r, c = mask[edge_index[0]], mask[edge_index[1]]
    keep = (r >= 0) & (c >= 0)
    ea = edge_attr[keep] if edge_attr is not None else None
    return torch.stack([r[keep], c[keep]], dim=0), ea


class SelectTopK(nn.Module):
    """Scalar selector that PyG >= 2.4 places inside SAGPooling (``select.weight`` [1, 1], U(-1, 1) at init):
    score = tanh(s * w / ||w||), i.e. the sign of ``w`` decides whether high or low GNN outputs are kept."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, 1).uniform_(-1.0, 1.0))

    def forward(self, s: Tensor) -> Tensor:
        score = (s.view(-1, 1) * self.weight).sum(dim=-1)
        return torch.tanh(score / self.weight.norm(p=2, dim=-1))


class SAGPooling(nn.Module):
    """PyG SAGPooling(in_channels, ratio=0.5, GNN=GraphConv, nonlinearity=tanh).

    ``select_weight=False`` restates PyG <= 2.3 (score = tanh(gnn(x))); ``True`` restates PyG >= 2.4, whose
    SelectTopK multiplies the GNN output by a learnable scalar before the nonlinearity."""

    def __init__(self, in_channels: int, ratio: float = 0.5, GNN=GraphConv, select_weight: bool = False):
        super().__init__()
        self.ratio = ratio
        self.gnn = GNN(in_channels, 1)
        self.select = SelectTopK() if select_weight else None

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        if batch is None:
            batch = edge_index.new_zeros(x.size(0))
        s = self.gnn(x, edge_index).view(-1)
        score = torch.tanh(s) if self.select is None else self.select(s)
        perm = topk(score, self.ratio, batch)
        x_out = x[perm] * score[perm].view(-1, 1)
        b_out = batch[perm]
        ei, ea = filter_adj(edge_index, edge_attr, perm, x.size(0))
        return x_out, ei, ea, b_out, perm, score[perm]


# ----------------------------------------------------------------------------
# data containers
# ----------------------------------------------------------------------------
class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, batch=None):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch

    @property
    def num_nodes(self):
        return self.x.size(0)

    @property
    def num_node_features(self):
        return self.x.size(1)


class Batch(Data):
    @staticmethod
    def from_data_list(items):
        xs, eis, eas, bs, ptr, off = [], [], [], [], [0], 0
        for g, d in enumerate(items):
            n = d.x.size(0)
            xs.append(d.x)
            eis.append(d.edge_index + off)
            if d.edge_attr is not None:
                eas.append(d.edge_attr)
            bs.append(torch.full((n,), g, dtype=torch.long))
            off += n
            ptr.append(off)
        out = Batch(torch.cat(xs), torch.cat(eis, dim=1),
                    torch.cat(eas) if eas else None, torch.cat(bs))
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.num_graphs = len(items)
        return out


# ----------------------------------------------------------------------------
# independent dense formulations (second implementation used to pin the above)
# ----------------------------------------------------------------------------
def dense_adjacency(edge_index: Tensor, n: int, dtype=torch.float64) -> Tensor:
    """A[i, j] = number of edges j -> i."""
    a = torch.zeros(n, n, dtype=dtype)
    a.index_put_((edge_index[1], edge_index[0]), torch.ones(edge_index.size(1), dtype=dtype), accumulate=True)
    return a


def dense_sage_mean(x: Tensor, edge_index: Tensor) -> Tensor:
    a = dense_adjacency(edge_index, x.size(0), x.dtype)
    deg = a.sum(1).clamp(min=1)
    return (a @ x) / deg.view(-1, 1)


def dense_gat(x, edge_index, w, att_src, att_dst, bias, slope=0.2):
    """Single-head GAT by masked dense softmax (multi-edges counted)."""
    n = x.size(0)
    xw = x @ w.t()
    a_s, a_d = xw @ att_src.view(-1), xw @ att_dst.view(-1)
    ei = add_self_loops(remove_self_loops(edge_index), n)
    cnt = dense_adjacency(ei, n, x.dtype)
    e = F.leaky_relu(a_d.view(-1, 1) + a_s.view(1, -1), slope)
    e = e.masked_fill(cnt == 0, float("-inf"))
    p = torch.exp(e - e.max(1, keepdim=True).values) * cnt
    alpha = p / (p.sum(1, keepdim=True) + 1e-16)
    return alpha @ xw + bias


def glorot_bound(fan_in: int, fan_out: int) -> float:
    return math.sqrt(6.0 / (fan_in + fan_out))
