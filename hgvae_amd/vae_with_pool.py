"""Names exported by the reference's ``autoencoder/vae_with_pool.py`` (:22-248), backed by the HIP path."""
from .data import CustomGraphDataset, collate_fn  # noqa: F401
from .vae import PoolGraph, PoolUnpoolGraph  # noqa: F401
from .vae import PoolVAE as VAE  # noqa: F401
from .vae import compute_rmsd, test_vae_rmsd, train_vae, validate_vae  # noqa: F401
