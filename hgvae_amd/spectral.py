"""Spectral terms of the reference's Laplacian-regularised variants, on the device.

* :func:`normalized_laplacian` - ``L = I - D^-1/2 A D^-1/2`` as a dense float32 matrix
  (autoencoder/vae_laplacian.py:24-46; the same function at autoencoder/vae_custom_pool.py:19-41): one HIP entry point
  counts the edges into the output buffer and finishes every element in place, instead of ``to_dense_adj`` and two
  n x n x n products with diagonal matrices.
* :func:`fiedler_value` - second-smallest eigenvalue (vae_custom_pool.py:50-53), the quantity
  ``SpectralRegularizedSAGPooling`` compares before and after pooling (:63-76).
* :func:`l_frequency` - the loss term of ``VAE.l_frequency`` (vae_laplacian.py:170-191).  As written in the reference
  it keeps ONE eigenvalue per graph (``eigvals[:1]``) and min-max normalises that single number, i.e. it evaluates
  0 / 0: the term is NaN for every input, and so is the training loss it is added to (:214-217).  ``num_eigs=1``
  reproduces exactly that; a larger ``num_eigs`` compares that many of the smallest eigenvalues and is finite whenever
  they are not all equal.

Eigenvalues come from ``torch.linalg.eigvalsh`` on the device (the library's symmetric eigensolver, as in the
reference); nothing here runs on the host.
"""
from __future__ import annotations

import torch

from . import _lib


def _raw_edges(edge_index) -> torch.Tensor:
    return edge_index if isinstance(edge_index, torch.Tensor) else edge_index.edge_index


def compute_degree(edge_index, num_nodes: int) -> torch.Tensor:
    """Out-degree as float32: one count per edge at its SOURCE (vae_laplacian.py:49-52)."""
    ei = _raw_edges(edge_index)
    return torch.bincount(ei[0], minlength=num_nodes).to(torch.float32)


def normalized_laplacian(edge_index, num_nodes: int) -> torch.Tensor:
    """[n, n] float32 normalised Laplacian; duplicate edges count as often as they occur (``to_dense_adj`` sums them)."""
    ei = _raw_edges(edge_index)
    _lib.require_cuda(ei)
    if ei.dim() != 2 or ei.size(0) != 2:
        raise ValueError(f"Expected edge_index to have shape (2, E), but got {tuple(ei.shape)}.")
    if num_nodes is None:
        raise ValueError("num_nodes must be provided.")
    n, e = int(num_nodes), ei.size(1)
    ei = ei.to(torch.int64)
    status = torch.zeros(1, dtype=torch.int32, device=ei.device)
    # degree over valid sources only, so that a bad index is reported by the kernel's status word, not by bincount
    src = ei[0]
    deg = torch.zeros(n, dtype=torch.float32, device=ei.device)
    if e:
        ok = (src >= 0) & (src < n)
        deg.scatter_add_(0, src.clamp(0, max(n - 1, 0)), ok.to(torch.float32))
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    out = torch.empty(n, n, dtype=torch.float32, device=ei.device)
    _lib.call("hgv_normalized_laplacian", _lib.ptr(ei) if e else None, ei.stride(0), ei.stride(1), e, n, dis.data_ptr(),
              out.data_ptr(), n, status.data_ptr(), _lib.stream())
    if e and int(status.item()) != 0:
        raise ValueError(f"Invalid edge_index, max index is {int(ei.max())}, but num_nodes is {n}.")
    return out


def laplacian_eigenvalues(edge_index, num_nodes: int) -> torch.Tensor:
    """Ascending eigenvalues of the normalised Laplacian (symmetric part is what ``eigvalsh`` reads: the lower triangle)."""
    return torch.linalg.eigvalsh(normalized_laplacian(edge_index, num_nodes))


def fiedler_value(edge_index, num_nodes: int) -> torch.Tensor:
    """``eigvalsh(L)[1]`` (vae_custom_pool.py:50-53)."""
    return laplacian_eigenvalues(edge_index, num_nodes)[1]


def _minmax(v: torch.Tensor) -> torch.Tensor:
    return (v - torch.min(v)) / (torch.max(v) - torch.min(v))


def l_frequency(edge_index_orig, edge_index_pooled, num_nodes_orig: int, num_nodes_pooled: int, num_eigs: int = 1) -> torch.Tensor:
    """MSE between the min-max normalised ``num_eigs`` smallest Laplacian eigenvalues of the original and the pooled
    graph (vae_laplacian.py:170-191; ``num_eigs=1`` is the reference and yields NaN, see the module text)."""
    if num_nodes_orig is None or num_nodes_pooled is None:
        raise ValueError("The number of nodes for both original and pooled graphs must be provided.")
    a = laplacian_eigenvalues(edge_index_orig, num_nodes_orig)[:num_eigs]
    b = laplacian_eigenvalues(edge_index_pooled, num_nodes_pooled)[:num_eigs]
    return torch.nn.functional.mse_loss(_minmax(a), _minmax(b))
