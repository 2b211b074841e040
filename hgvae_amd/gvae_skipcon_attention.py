"""Names exported by the reference's ``gvae/gvae_skipcon_attention.py`` (:20-261), backed by the HIP path."""
from .data import CustomGraphDataset, collate_fn  # noqa: F401
from .vae import PoolGraph, PoolUnpoolGraph  # noqa: F401
from .vae import SkipconAttentionVAE as VAE  # noqa: F401
from .vae import compute_rmsd, node_embeddings, test_vae_rmsd, train_vae, validate_vae  # noqa: F401
