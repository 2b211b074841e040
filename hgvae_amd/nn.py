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
        agg = Fn.aggregate(x, g, mean=True)
        return Fn.dual_linear(agg, self.lin_l.weight, x, self.lin_r.weight, self.lin_l.bias)

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
        agg = Fn.aggregate(x, g, mean=False)
        return Fn.dual_linear(agg, self.lin_rel.weight, x, self.lin_root.weight, self.lin_rel.bias)
