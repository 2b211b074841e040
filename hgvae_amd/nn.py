"""Graph operators with PyTorch Geometric's constructor/forward signatures and parameter names.

These are the drop-in replacements for the PyG classes the reference imports
(``from torch_geometric.nn import SAGEConv, SAGPooling, GATConv`` at
gvae/gvae_skipcon_attention.py:16, ``GCNConv`` at autoencoder/vae_custom_pool.py:13).
``forward`` takes the same ``(x, edge_index)`` arguments; ``edge_index`` may also be a
prebuilt :class:`hgvae_amd.graph.Graph`.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as Fn
from .graph import Graph, graph_of


def _kaiming_linear_(lin: nn.Linear) -> None:
    nn.init.kaiming_uniform_(lin.weight, a=math.sqrt(5))
    if lin.bias is not None:
        bound = 1.0 / math.sqrt(lin.weight.size(1))
        nn.init.uniform_(lin.bias, -bound, bound)


class SAGEConv(nn.Module):
    """out_i = lin_l(mean_{j -> i} x_j) + lin_r(x_i)   (PyG defaults: mean, root weight, bias in lin_l)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        _kaiming_linear_(self.lin_l)
        _kaiming_linear_(self.lin_r)

    def forward(self, x: torch.Tensor, edge_index) -> torch.Tensor:
        g = graph_of(edge_index, x.size(0))
        return Fn.sage_conv(x, g, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, mean=True)

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, aggr=mean"


class GraphConv(nn.Module):
    """out_i = lin_rel(sum_{j -> i} x_j) + lin_root(x_i)   (PyG GraphConv, aggr='add')."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)
        _kaiming_linear_(self.lin_rel)
        _kaiming_linear_(self.lin_root)

    def forward(self, x: torch.Tensor, edge_index) -> torch.Tensor:
        g = graph_of(edge_index, x.size(0))
        return Fn.sage_conv(x, g, self.lin_rel.weight, self.lin_rel.bias, self.lin_root.weight, mean=False)


class GCNConv(nn.Module):
    """out = D^-1/2 (A + I) D^-1/2 (x W^T) + b with the degree taken over targets incl. self loops
    (PyG GCNConv defaults: add_self_loops, normalize, unweighted edges)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        nn.init.xavier_uniform_(self.lin.weight)

    def forward(self, x: torch.Tensor, edge_index) -> torch.Tensor:
        g = graph_of(edge_index, x.size(0)).with_self_loops()
        h = Fn.dual_linear(x, self.lin.weight)
        return Fn.gcn_aggregate(h, g) + self.bias      # both degree scalings run inside the aggregation kernel


class GATConv(nn.Module):
    """Graph attention (PyG GATConv: concat=True, negative_slope=0.2, add_self_loops=True, bias).

    Parameters follow PyG <= 2.3 naming (``lin_src`` and ``lin_dst`` share one weight);
    a PyG >= 2.5 checkpoint with a single ``lin.weight`` is accepted on load.
    """

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, dropout: float = 0.0,
                 negative_slope: float = 0.2):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.dropout, self.negative_slope = dropout, negative_slope
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        nn.init.xavier_uniform_(self.lin_src.weight)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        single = prefix + "lin.weight"
        if single in state_dict:
            w = state_dict.pop(single)
            state_dict[prefix + "lin_src.weight"] = w
            state_dict[prefix + "lin_dst.weight"] = w
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x: torch.Tensor, edge_index) -> torch.Tensor:
        n, h, c = x.size(0), self.heads, self.out_channels
        lg = graph_of(edge_index, n).with_self_loops()
        xw = Fn.dual_linear(x, self.lin_src.weight)
        if h == 1:
            # both attention logits come from one pass over xw inside the operator: [a_src | a_dst] = xw @ att^T
            att = torch.cat([self.att_src.view(1, -1), self.att_dst.view(1, -1)])
            return Fn.gat_aggregate(xw, att, lg, self.negative_slope, self.dropout, self.training, bias=self.bias)
        else:
            raise NotImplementedError(
                "heads > 1 is not supported: the reference's encoder only type-checks with heads=1 "
                "(gvae/gvae_skipcon_attention.py:74,96)")


class SelectTopK(nn.Module):
    """The scalar selector PyG >= 2.4 keeps inside SAGPooling (checkpoint key ``select.weight``, shape [1, 1],
    U(-1, 1) at init): score = tanh(s * w / ||w||)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(1, 1).uniform_(-1.0, 1.0))

    def forward(self, s: torch.Tensor) -> torch.Tensor:
        w = self.weight.view(-1)
        return torch.tanh(s * (w / w.norm(p=2)))


class SAGPooling(nn.Module):
    """Self-attention graph pooling (PyG SAGPooling: ratio=0.5, GNN scorer -> tanh -> per-graph top-k).

    Returns ``(x[perm] * score[perm], edge_index_cg, edge_attr_cg, batch[perm], perm, score[perm])``.

    PyG <= 2.3 scores nodes with ``tanh(gnn(x))``.  PyG >= 2.4 routes the GNN output through a ``SelectTopK``
    module with one learnable scalar (``select.weight``): ``tanh(gnn(x) * w / ||w||)``.  ``select_weight=True``
    builds that variant; loading a checkpoint that holds ``select.weight`` switches to it as well, so
    ``load_state_dict`` accepts files written by either PyG generation.
    """

    def __init__(self, in_channels: int, ratio: float = 0.5, GNN=GraphConv, select_weight: bool = False, **kwargs):
        super().__init__()
        self.in_channels, self.ratio = in_channels, ratio
        self.gnn = GNN(in_channels, 1, **kwargs)
        self.select = SelectTopK() if select_weight else None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        if prefix + "select.weight" in state_dict and self.select is None:
            ref = self.gnn.parameters().__next__()
            self.select = SelectTopK().to(ref.device)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x, edge_index, edge_attr=None, batch=None):
        n = x.size(0)
        if batch is None:
            batch = torch.zeros(n, dtype=torch.long, device=x.device)
        raw_ei = edge_index.edge_index if isinstance(edge_index, Graph) else edge_index
        s = self.gnn(x, edge_index).reshape(-1)
        score = torch.tanh(s) if self.select is None else self.select(s)
        perm = Fn.topk_perm(score, self.ratio, batch)
        x_out = Fn.gather_rows(x, perm, score)
        b_out = Fn.gather_i64(batch, perm)
        ei_cg, ea_cg = Fn.filter_adj(raw_ei, edge_attr, perm, n)
        return x_out, ei_cg, ea_cg, b_out, perm, score.index_select(0, perm)
