"""CPU restatement of the reference's dense-Laplacian spectral terms.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  PARITY UNPINNED beyond hand-computed answers: the reference
holds no recorded outputs of these functions, and ``torch_geometric.utils.to_dense_adj`` (absent here) is restated as
its published behaviour for a single graph: ``adj[r, c]`` = number of edges r -> c.

  normalized_laplacian / compute_degree   autoencoder/vae_laplacian.py:24-52 (same text: autoencoder/vae_custom_pool.py:19-47)
  compute_fiedler_value                    autoencoder/vae_custom_pool.py:50-53
  l_frequency                              autoencoder/vae_laplacian.py:170-191
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def compute_degree(edge_index: Tensor, num_nodes: int) -> Tensor:
    row = edge_index[0]
    out = torch.zeros(num_nodes, dtype=torch.float)
    return out.scatter_add_(dim=0, index=row, src=torch.ones_like(row).float())


def normalized_laplacian(edge_index: Tensor, num_nodes: int) -> Tensor:
    if edge_index.shape[0] != 2:
        raise ValueError(f"Expected edge_index to have shape (2, E), but got {edge_index.shape}.")
    if num_nodes is None:
        raise ValueError("num_nodes must be provided.")
    if edge_index.numel() and edge_index.max() >= num_nodes:
        raise ValueError(f"Invalid edge_index, max index is {edge_index.max()}, but num_nodes is {num_nodes}.")
    deg = compute_degree(edge_index, num_nodes)
    deg_inv_sqrt = deg.pow(-0.5)
    deg_inv_sqrt[deg_inv_sqrt == float("inf")] = 0
    adj = torch.zeros(num_nodes, num_nodes)
    adj.index_put_((edge_index[0], edge_index[1]), torch.ones(edge_index.size(1)), accumulate=True)
    identity = torch.eye(num_nodes)
    return identity - torch.mm(torch.mm(deg_inv_sqrt.diag(), adj), deg_inv_sqrt.diag())


def compute_fiedler_value(edge_index: Tensor, num_nodes: int) -> Tensor:
    return torch.linalg.eigvalsh(normalized_laplacian(edge_index, num_nodes))[1]


def l_frequency(edge_index_orig, edge_index_pooled, num_nodes_orig, num_nodes_pooled, num_eigs: int = 1) -> Tensor:
    """``num_eigs=1`` is the reference text (one eigenvalue, min-max normalised: 0 / 0)."""
    if num_nodes_orig is None or num_nodes_pooled is None:
        raise ValueError("The number of nodes for both original and pooled graphs must be provided.")
    a, _ = torch.linalg.eigh(normalized_laplacian(edge_index_orig, num_nodes_orig))
    b, _ = torch.linalg.eigh(normalized_laplacian(edge_index_pooled, num_nodes_pooled))
    a, b = a[:num_eigs], b[:num_eigs]
    a = (a - torch.min(a)) / (torch.max(a) - torch.min(a))
    b = (b - torch.min(b)) / (torch.max(b) - torch.min(b))
    return F.mse_loss(a, b)
