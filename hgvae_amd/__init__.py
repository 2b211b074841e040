"""MI355X-native graph-VAE message-passing train step (drop-in for the reference's PyG path).

GPU only: every operator runs hand-written gfx950 kernels from the in-tree
``hgvae_amd/csrc/libhgvae_hip.so`` through the C ABI of ``include/hgvae_hip.h``.
"""
from . import _lib  # noqa: F401
from .data import Batch, CustomGraphDataset, Data, collate_fn  # noqa: F401
from .graph import Graph, graph_of  # noqa: F401
from .noise import NoiseTape, manual_seed, use_noise  # noqa: F401
from .shard import GraphShard  # noqa: F401

__all__ = ["GraphShard", "Batch", "CustomGraphDataset", "Data", "collate_fn", "Graph", "graph_of", "NoiseTape", "manual_seed",
           "use_noise"]
