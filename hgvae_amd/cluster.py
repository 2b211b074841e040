"""Device half of the reference's clustering export (gvae/gvae_skipcon.py:333-383).

The reference encodes one graph into per-node latent vectors (:333-339), builds the full atom-atom distance
matrix with ``scipy.spatial.distance_matrix`` (:352-354), reduces it to 10 columns with ``sklearn.decomposition.PCA``
(:356-358), stacks the two (:361) and hands the rows to Ward linkage / agglomerative clustering (:364-383).
Everything up to the stacked matrix runs here on the GPU in float64 (the dtype scipy / sklearn work in):

* :func:`distance_matrix` - one HIP kernel writes the n x n matrix once and its row sums with it;
* :func:`pca_fit_transform` - top components of the column-centred matrix by blocked subspace iteration, the
  centring applied implicitly through the row sums (the matrix is symmetric, so column means = row means), i.e.
  the matrix is only ever read by the float64 library GEMM;
* :func:`combined_embeddings` - ``hstack(embeddings, pca_scores)`` as one device tensor.

The tree building itself (:func:`ward_linkage`, :func:`cut_tree`) stays with scipy on the host, as in the reference.
Component signs follow sklearn >= 1.5 (``svd_flip`` on the components: the entry of largest magnitude in every
component is positive); older sklearn decides the sign on the scores instead, which can flip columns - cluster
assignments do not depend on it.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def distance_matrix(coords: torch.Tensor, return_rowsum: bool = False):
    """[n, n] float64 Euclidean distances of the rows of ``coords`` [n, d <= 4] (scipy.spatial.distance_matrix(c, c))."""
    _lib.require_cuda(coords)
    c = coords.detach().to(torch.float64).contiguous()
    if c.dim() != 2 or not 1 <= c.size(1) <= 4:
        raise _lib.HgvError(f"coords must be [n, d] with 1 <= d <= 4, got {tuple(c.shape)}")
    n = c.size(0)
    out = torch.empty(n, n, dtype=torch.float64, device=c.device)
    rowsum = torch.empty(n, dtype=torch.float64, device=c.device)
    _lib.call("hgv_distance_matrix", c.data_ptr(), n, c.size(1), out.data_ptr(), n, rowsum.data_ptr(), _lib.stream())
    return (out, rowsum) if return_rowsum else out


def pca_fit_transform(x: torch.Tensor, n_components: int = 10, col_sum: Optional[torch.Tensor] = None,
                      oversample: int = 10, iters: int = 40, tol: float = 1e-13, seed: int = 0) -> torch.Tensor:
    """Scores ``(x - column means) @ components^T`` of the leading ``n_components`` principal components: what
    ``PCA(n_components).fit_transform(x)`` returns.  ``x`` [m, n] float64 on the device is never modified or copied:
    with c = column means, every product with the centred matrix is formed as ``x @ q - 1 (c^T q)`` /
    ``x^T y - c (1^T y)``.  Blocked subspace iteration with a fixed random start (deterministic), stopped when the
    Ritz values have settled to ``tol`` (relative) or after ``iters`` rounds, then one small SVD."""
    _lib.require_cuda(x)
    if x.dtype != torch.float64 or x.dim() != 2:
        raise _lib.HgvError("pca_fit_transform expects a float64 matrix")
    m, n = x.shape
    k = min(n_components, m, n)
    b = min(k + oversample, m, n)
    c = (col_sum if col_sum is not None else x.sum(dim=0)) / m
    gen = torch.Generator(device=x.device).manual_seed(seed)
    q = torch.linalg.qr(torch.randn(n, b, dtype=torch.float64, device=x.device, generator=gen)).Q
    prev, settled, drift = None, False, float("inf")
    for _ in range(max(iters, 1)):
        y = x @ q - (c @ q).unsqueeze(0)                       # centred matrix times q   [m, b]
        z = x.t() @ y - torch.outer(c, y.sum(dim=0))           # its transpose times y    [n, b]
        q, r = torch.linalg.qr(z)
        ritz = r.diagonal().abs()[:k]
        if prev is not None:
            drift = float(((ritz - prev).abs().max() / ritz.max()).item())      # one read-back per round, as the stopping test needs
            if drift <= tol:
                settled = True
                break
        prev = ritz
    if not settled and iters > 1 and drift > 1e3 * tol:
        import warnings
        warnings.warn(f"pca_fit_transform: the subspace iteration stopped after {iters} rounds with the Ritz values still "
                      f"moving by {drift:.1e} (relative; tol {tol:.1e}); raise iters or oversample for tighter scores",
                      RuntimeWarning, stacklevel=2)
    y = x @ q - (c @ q).unsqueeze(0)
    u, s, vt = torch.linalg.svd(y, full_matrices=False)        # y = u s vt, components = vt @ q^T
    comps = vt[:k] @ q.t()                                     # [k, n]
    idx = comps.abs().argmax(dim=1, keepdim=True)
    sign = torch.sign(comps.gather(1, idx)).squeeze(1)
    sign = torch.where(sign == 0, torch.ones_like(sign), sign)
    return (u[:, :k] * s[:k]) * sign


def combined_embeddings(embeddings: torch.Tensor, coords: torch.Tensor, n_components: int = 10) -> torch.Tensor:
    """``np.hstack((new_embeddings, PCA(n_components).fit_transform(distance_matrix(coords, coords))))`` (:352-361)
    as a float64 device tensor [n, Z + n_components]."""
    d, rowsum = distance_matrix(coords, return_rowsum=True)
    scores = pca_fit_transform(d, n_components, col_sum=rowsum)
    return torch.cat([embeddings.detach().to(torch.float64), scores], dim=1)


def ward_linkage(combined: torch.Tensor):
    """scipy's Ward linkage of the stacked rows (:368), on the host as in the reference."""
    from scipy.cluster.hierarchy import linkage
    return linkage(combined.detach().cpu().numpy(), "ward")


def cut_tree(linked, height: float):
    """Cluster labels from cutting the dendrogram at ``height`` (the reference asks for the height interactively and
    re-clusters with AgglomerativeClustering(distance_threshold=height, linkage='ward'), :378-383: the same partition)."""
    from scipy.cluster.hierarchy import fcluster
    return fcluster(linked, t=height, criterion="distance") - 1
