"""Names exported by the reference's ``autoencoder/vae_custom_pool.py`` (:56-263), backed by the HIP path."""
from .data import CustomGraphDataset, collate_fn  # noqa: F401
from .nn import SAGPooling as _SAGPooling, GCNConv as _GCNConv
from .vae import CoarsenGraph as PoolUnpoolGraph  # noqa: F401
from .vae import CustomPoolVAE as VAE  # noqa: F401
from .vae import train_vae_custom_pool as train_vae  # noqa: F401


class SpectralRegularizedSAGPooling(_SAGPooling):
    """SAGPooling with a GCNConv scorer (the reference's subclass, vae_custom_pool.py:56-81)."""

    def __init__(self, in_channels, alpha=0.5, ratio=0.5, GNN=_GCNConv, **kwargs):
        super().__init__(in_channels, ratio=ratio, GNN=GNN, **kwargs)
        self.alpha = alpha
