"""CPU restatement of the reference's four graph-VAE models and their train steps.

TEST INFRASTRUCTURE ONLY - see ``oracle/__init__.py``.  PARITY UNPINNED (no
PyG, no reference tests).  Module trees reproduce the reference's
``state_dict`` keys; every stochastic draw goes through a ``NoiseTape`` so a
test can feed the same dropout masks and normal draws to the HIP path.

Followed files (relative to /root/reference):
  SkipconVAE           gvae/gvae_skipcon.py:53-156, train step :159-187
  SkipconAttentionVAE  gvae/gvae_skipcon_attention.py:53-154, :157-185
  PoolVAE              autoencoder/vae_with_pool.py:33-143, :146-173
  CustomPoolVAE        autoencoder/vae_custom_pool.py:94-205, :208-232
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .pyg_ops import GATConv, GCNConv, GraphConv, NoiseTape, SAGEConv, SAGPooling, dropout_with


def _bn(m: nn.BatchNorm1d, x):
    return m(x)


def reparameterize(mean, log_std, n_samples: int, tape: Optional[NoiseTape]):
    """gvae_skipcon.py:138-142 - mean over n_samples of (mean + std * eps_k)."""
    std = log_std.exp()
    eps = tape.next() if tape is not None else None
    if eps is None:
        eps = torch.stack([torch.randn_like(std) for _ in range(n_samples)])
        if tape is not None:
            tape.record.append(eps)
    samples = [mean + std * eps[k] for k in range(n_samples)]
    return torch.mean(torch.stack(samples), dim=0)


def kl_divergence(mean, log_std):
    """gvae_skipcon.py:153-156."""
    std = log_std.exp()
    kl = -0.5 * torch.sum(1 + 2 * log_std - mean.pow(2) - std.pow(2), dim=-1)
    return kl.mean()


def compute_rmsd(a, b):
    """gvae_skipcon.py:207-215."""
    d = a - b
    return torch.sqrt((d * d).mean(dim=-1)).mean()


class SkipconVAE(nn.Module):
    """Three SAGE levels + mean/log_std head; concat-skip MLP decoder."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, dropout_prob=0.5):
        super().__init__()
        h, z, p = hidden_channels, out_channels, dropout_prob
        self.dropout_prob, self.n_samples = p, n_samples
        self.encoder1 = nn.Sequential(SAGEConv(in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.encoder2 = nn.Sequential(self._level2(h, p), nn.Dropout(p), nn.BatchNorm1d(h))
        self.encoder3 = nn.Sequential(SAGEConv(h, h), nn.Dropout(p), nn.BatchNorm1d(h), SAGEConv(h, 2 * z))
        self.decoder1 = nn.Sequential(nn.Linear(z, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.decoder2 = nn.Sequential(nn.Linear(2 * h, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.decoder3 = nn.Sequential(nn.Linear(2 * h, h), nn.Dropout(p), nn.BatchNorm1d(h), nn.Linear(h, in_channels))

    def _level2(self, h, p):
        return SAGEConv(h, h)

    def _drop(self, x, tape):
        return dropout_with(x, self.dropout_prob, self.training, tape)

    def _encode_level2(self, x1, edge_index, tape):
        x2 = F.relu(self.encoder2[0](x1, edge_index))
        x2 = self._drop(x2, tape)
        return _bn(self.encoder2[2], x2)

    def encode(self, x, edge_index, tape=None):
        x1 = _bn(self.encoder1[2], self._drop(F.relu(self.encoder1[0](x, edge_index)), tape))
        x2 = self._encode_level2(x1, edge_index, tape)
        x3 = _bn(self.encoder3[2], self._drop(F.relu(self.encoder3[0](x2, edge_index)), tape))
        mean, log_std = self.encoder3[3](x3, edge_index).chunk(2, dim=-1)
        return x1, x2, mean, log_std

    def decode(self, z, skip1, skip2, tape=None):
        h1 = _bn(self.decoder1[2], self._drop(self.decoder1[0](z), tape))
        h1 = torch.cat([h1, skip1], dim=1)
        h2 = _bn(self.decoder2[2], self._drop(self.decoder2[0](h1), tape))
        h2 = torch.cat([h2, skip2], dim=1)
        h3 = _bn(self.decoder3[2], self._drop(self.decoder3[0](h2), tape))
        return self.decoder3[3](h3)

    def forward(self, x, edge_index, batch=None, tape=None):
        s1, s2, mean, log_std = self.encode(x, edge_index, tape)
        z = reparameterize(mean, log_std, self.n_samples, tape)
        return self.decode(z, s1, s2, tape), mean, log_std


class SkipconAttentionVAE(SkipconVAE):
    """Level 2 is relu(GATConv) -> BatchNorm (no feature dropout there)."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, heads=1):
        self._heads = heads
        super().__init__(in_channels, hidden_channels, out_channels, n_samples, dropout_prob=0.49)
        h = hidden_channels
        self.encoder2 = nn.Sequential(GATConv(h, h, heads=heads, dropout=0.49), nn.BatchNorm1d(h * heads))

    def _encode_level2(self, x1, edge_index, tape):
        x2 = F.relu(self.encoder2[0](x1, edge_index, tape))
        return _bn(self.encoder2[1], x2)


def skipcon_loss(model, x, edge_index, batch=None, lambda_kl=1.0, tape=None):
    """Loss of one reference train step: forward, SECOND encode, mse + lambda*kl."""
    recon, _, _ = model(x, edge_index, batch, tape)
    _, _, mean, log_std = model.encode(x, edge_index, tape)
    rec = F.mse_loss(recon, x, reduction="mean")
    kl = kl_divergence(mean, log_std)
    return rec + lambda_kl * kl, rec, kl


class PoolUnpoolGraph(nn.Module):
    """vae_with_pool.py:33-52."""

    def __init__(self, num_node_features, GNN=GraphConv):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, GNN=GNN)

    def forward(self, x, edge_index, batch):
        x_cg, ei_cg, _, _, perm, _ = self.pool1(x, edge_index, batch=batch)
        new_batch = batch[perm]
        x_unpooled = x[perm]
        ei_unpooled = torch.stack([perm[ei_cg[0]], perm[ei_cg[1]]], dim=0)
        return x_unpooled, x_cg, ei_unpooled, new_batch


class PoolGraph(nn.Module):
    def __init__(self, num_node_features, GNN=GraphConv):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, GNN=GNN)


class PoolVAE(nn.Module):
    """Two SAGE levels, residual-add decoder, pool/unpool evaluated in forward."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10):
        super().__init__()
        h, z, p = hidden_channels, out_channels, 0.5
        self.dropout_prob, self.n_samples, self.out_channels = p, n_samples, z
        self.encoder1 = nn.ModuleList([SAGEConv(in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.encoder2 = nn.ModuleList([SAGEConv(h, 2 * h), nn.Dropout(p), nn.BatchNorm1d(2 * h), SAGEConv(2 * h, 2 * z)])
        self.decoder1 = nn.ModuleList([nn.Linear(z, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.decoder2 = nn.ModuleList([nn.Linear(h, in_channels)])
        self.pool_graph = PoolGraph(h)
        self.pool_unpool_graph = PoolUnpoolGraph(h)

    def _drop(self, x, tape):
        return dropout_with(x, self.dropout_prob, self.training, tape)

    def encode(self, x, edge_index, tape=None):
        x1 = _bn(self.encoder1[2], self._drop(F.relu(self.encoder1[0](x, edge_index)), tape))
        x2 = _bn(self.encoder2[2], self._drop(F.relu(self.encoder2[0](x1, edge_index)), tape))
        mean, log_std = self.encoder2[3](x2, edge_index).chunk(2, dim=-1)
        return x1, mean, log_std

    def decode(self, z, x1, tape=None):
        h = _bn(self.decoder1[2], self._drop(F.relu(self.decoder1[0](z)), tape))
        return self.decoder2[0](h + x1)

    def forward(self, x, edge_index, batch, tape=None):
        x1, mean, log_std = self.encode(x, edge_index, tape)
        z = reparameterize(mean, log_std, self.n_samples, tape)
        recon = self.decode(z, x1, tape)
        x_unpooled, _, ei_unpooled, _ = self.pool_unpool_graph(x1, edge_index, batch)
        return recon, x_unpooled, ei_unpooled


def pool_loss(model, x, edge_index, batch, lambda_kl=1.0, tape=None):
    """vae_with_pool.py:146-173 (pool outputs discarded, second encode for KL)."""
    recon, _, _ = model(x, edge_index, batch, tape)
    _, mean, log_std = model.encode(x, edge_index, tape)
    rec = F.mse_loss(recon, x, reduction="mean")
    kl = kl_divergence(mean, log_std)
    return rec + lambda_kl * kl, rec, kl


class CustomPoolUnpoolGraph(nn.Module):
    """vae_custom_pool.py:94-106: GCNConv-scored pooling, returns (x_cg, ei_cg, n_cg).

    The dense-Laplacian Fiedler values the reference also computes
    (vae_custom_pool.py:63-76) land in a tuple slot that this wrapper drops, so
    they never reach an output; they are not restated.
    """

    def __init__(self, num_node_features):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, ratio=0.5, GNN=GCNConv)

    def forward(self, x, edge_index, batch):
        x_cg, ei_cg, _, _, _, _ = self.pool1(x, edge_index, batch=batch)
        return x_cg, ei_cg, x_cg.size(0)


class CustomPoolVAE(nn.Module):
    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, beta=0.5):
        super().__init__()
        h, z, p = hidden_channels, out_channels, 0.5
        self.dropout_prob, self.n_samples, self.beta, self.out_channels = p, n_samples, beta, z
        self.encoder1 = nn.ModuleList([SAGEConv(in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.encoder2 = nn.ModuleList([SAGEConv(h, 2 * h), nn.Dropout(p), nn.BatchNorm1d(2 * h), SAGEConv(2 * h, 2 * z)])
        self.decoder1 = nn.ModuleList([nn.Linear(z + in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.decoder2 = nn.ModuleList([nn.Linear(h, in_channels)])
        self.pool_graph = CustomPoolUnpoolGraph(h)  # same parameter names: pool1.gnn.*
        self.pool_unpool_graph = CustomPoolUnpoolGraph(h)
        self.x1_transform = nn.Linear(h, in_channels)

    def _drop(self, x, tape):
        return dropout_with(x, self.dropout_prob, self.training, tape)

    def encode(self, x, edge_index, tape=None):
        x1 = _bn(self.encoder1[2], self._drop(F.relu(self.encoder1[0](x, edge_index)), tape))
        x2 = _bn(self.encoder2[2], self._drop(F.relu(self.encoder2[0](x1, edge_index)), tape))
        mean, log_std = self.encoder2[3](x2, edge_index).chunk(2, dim=-1)
        return x1, mean, log_std

    def decode(self, z, x1, tape=None):
        comb = torch.cat([z, self.x1_transform(x1)], dim=-1)
        h = _bn(self.decoder1[2], self._drop(F.relu(self.decoder1[0](comb)), tape))
        return self.decoder2[0](h)

    def forward(self, x, edge_index, batch, tape=None):
        x1, mean, log_std = self.encode(x, edge_index, tape)
        z = reparameterize(mean, log_std, self.n_samples, tape)
        recon = self.decode(z, x1, tape)
        x_cg, ei_cg, _ = self.pool_unpool_graph(x1, edge_index, batch)
        return recon, x_cg, ei_cg, mean, log_std, x1

    def recon_loss(self, x_original, mean, log_std, x1, tape=None):
        z = reparameterize(mean, log_std, self.n_samples, tape)
        return F.mse_loss(self.decode(z, x1, tape), x_original, reduction="mean")


def custom_pool_loss(model, x, edge_index, batch, tape=None):
    """vae_custom_pool.py:219-225: loss re-samples z and re-decodes; recon + beta*kl."""
    _, _, _, mean, log_std, x1 = model(x, edge_index, batch, tape)
    rec = model.recon_loss(x, mean, log_std, x1, tape)
    kl = kl_divergence(mean, log_std)
    return rec + model.beta * kl, rec, kl
