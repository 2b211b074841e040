"""Names exported by the reference's ``gvae/gvae_skipcon.py`` (:20-263), backed by the HIP path.

``gvae/hypertune.py:2`` imports ``VAE, train_vae, validate_vae`` from this module name.
"""
from .data import CustomGraphDataset, collate_fn  # noqa: F401
from .vae import SkipconVAE as VAE  # noqa: F401
from .vae import PoolGraph, PoolUnpoolGraph  # noqa: F401
from .vae import compute_rmsd, node_embeddings, test_vae_rmsd, train_vae, validate_vae  # noqa: F401
