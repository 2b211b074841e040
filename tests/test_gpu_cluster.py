"""Clustering export on the device (gvae/gvae_skipcon.py:352-361): distance matrix against scipy, PCA scores against
sklearn's full-SVD PCA, on the chignolin coordinates of the reference's PDB file and on a synthetic 2000-atom coil."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def coords_of(kind):
    if kind == "chignolin":
        return np.load(os.path.join(GOLDEN, "chig_pdb.npz"), allow_pickle=False)["coords"]
    rng = np.random.default_rng(5)
    v = rng.normal(size=(2000, 3))
    return np.cumsum(v / np.linalg.norm(v, axis=1, keepdims=True) * 3.8, axis=0)


@pytest.mark.parametrize("kind", ["chignolin", "coil2000"])
def test_distance_matrix_and_pca_match_scipy_and_sklearn(kind):
    from scipy.spatial import distance_matrix as sp_dm
    from sklearn.decomposition import PCA
    from hgvae_amd import cluster
    c = coords_of(kind)
    dev = torch.device("cuda:0")
    d, rowsum = cluster.distance_matrix(torch.from_numpy(c).to(dev), return_rowsum=True)
    want = sp_dm(c, c)
    assert d.dtype == torch.float64
    assert np.abs(d.cpu().numpy() - want).max() <= 1e-12 * want.max()
    assert np.allclose(rowsum.cpu().numpy(), want.sum(1), rtol=1e-12)
    scores = cluster.pca_fit_transform(d, 10, col_sum=rowsum).cpu().numpy()
    ref = PCA(n_components=10, svd_solver="full").fit_transform(want)
    assert scores.shape == ref.shape
    scale = np.abs(ref).max()
    assert np.abs(scores - ref).max() <= 1e-7 * scale, np.abs(scores - ref).max() / scale
    # without the precomputed column sums, and the stacked matrix of the export
    assert np.allclose(cluster.pca_fit_transform(d, 10).cpu().numpy(), scores, rtol=0, atol=1e-9 * scale)
    emb = torch.randn(len(c), 28, device=dev)
    comb = cluster.combined_embeddings(emb, torch.from_numpy(c).to(dev))
    assert comb.shape == (len(c), 38)
    assert torch.equal(comb[:, :28], emb.double()) and np.allclose(comb[:, 28:].cpu().numpy(), scores, atol=1e-9 * scale)
    linked = cluster.ward_linkage(comb)
    labels = cluster.cut_tree(linked, float(linked[-3, 2]))
    assert linked.shape == (len(c) - 1, 4) and labels.min() == 0 and 1 <= labels.max() <= 3
