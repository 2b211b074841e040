"""Graph-VAE models and train/eval loops with the reference's Python interface.

Module trees reproduce the reference's ``state_dict`` keys
(``encoder1.0.lin_l.weight`` ... ``decoder3.3.bias``, gvae/gvae_skipcon.py:60-101), so
checkpoints written by either side load in the other.  The ``Dropout`` /
``BatchNorm1d`` children only hold hyper-parameters, affine weights and running
statistics: the arithmetic of each level runs in fused HIP kernels
(aggregation -> projection pair -> relu/dropout/batch-norm).
"""
from __future__ import annotations

import contextlib
import os
import weakref
from typing import Optional

import torch
from torch import nn

from . import functional as Fn
from .graph import Graph, as_padded, graph_of, ld_of, new_padded, pad4
from .data import Data
from .nn import GATConv, GCNConv, GraphConv, SAGEConv, SAGPooling

_pad_cache: "dict[int, tuple]" = {}


def padded_input(x: torch.Tensor) -> torch.Tensor:
    """Feature matrix laid out for 16-byte loads; the copy is cached per input tensor object."""
    if x.requires_grad:
        return x
    key = id(x)
    hit = _pad_cache.get(key)
    if hit is not None and hit[0]() is x and hit[1] == x._version:
        return hit[2]
    p = as_padded(x)
    if p is x and x.dim() == 2 and ld_of(x) > pad4(x.size(1)):
        # a column slice of a wider matrix: the kernels load whole 16-byte chunks, and what lies behind the last column
        # here is foreign data (it would meet zero weights, but it takes part in the row scale of the fp16 projections)
        p = new_padded(x.size(0), x.size(1), x.device)
        p.copy_(x)
    if p is not x:
        _pad_cache[key] = (weakref.ref(x, lambda _r, k=key: _pad_cache.pop(k, None)), x._version, p)
    return p


def split_head(ml: torch.Tensor, z_dim: int):
    """chunk(2, dim=-1) of the [N, 2Z] head output; the halves remember their parent buffer."""
    mean, log_std = ml[:, :z_dim], ml[:, z_dim:]
    mean._hgv_parent = ml
    log_std._hgv_parent = ml
    return mean, log_std


def join_head(mean: torch.Tensor, log_std: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`split_head`; falls back to a concatenation for unrelated tensors."""
    parent = getattr(mean, "_hgv_parent", None)
    if parent is not None and parent is getattr(log_std, "_hgv_parent", None) \
            and parent.size(1) == mean.size(1) + log_std.size(1):
        return parent
    return torch.cat([mean, log_std], dim=1)


def compute_rmsd(coords1: torch.Tensor, coords2: torch.Tensor) -> torch.Tensor:
    """Mean over rows of sqrt(mean over columns of squared difference) (gvae_skipcon.py:207-215)."""
    assert coords1.size() == coords2.size(), "Coordinate tensors must have the same shape"
    return torch.sqrt(((coords1 - coords2) ** 2).mean(dim=-1)).mean()


class _VAEBase(nn.Module):
    dropout_prob: float
    n_samples: int

    # one encoder/decoder level: conv-or-linear output -> (relu) -> dropout -> batch norm
    def _finish(self, y, dropout: Optional[nn.Dropout], bn: nn.BatchNorm1d, relu: bool, fork: bool = False):
        p = dropout.p if dropout is not None else 0.0
        return Fn.bn_act(y, bn, relu, p, self.training, fork=fork)

    def _sage_level(self, seq, x, g, fork: bool = False):
        conv, drop, bn = seq[0], seq[1], seq[2]
        # a SAGEConv is one projection operator (its product may store relu -> dropout itself); attention layers are not
        return Fn.bn_level(lambda: conv(x, g), bn, True, drop.p, self.training, fork=fork, act_ok=type(conv) is SAGEConv,
                           rows=x.size(0))

    def _linear_level(self, xa, xb, lin, dropout, bn, relu: bool):
        """Decoder level: Linear over xa (or over [xa | xb]) -> (relu) -> dropout -> batch norm."""
        if xb is None:
            make = lambda: Fn.dual_linear(xa, lin.weight, None, None, lin.bias)       # noqa: E731
        else:
            make = lambda: Fn.concat_linear(xa, xb, lin.weight, lin.bias)             # noqa: E731
        return Fn.bn_level(make, bn, relu, dropout.p if dropout is not None else 0.0, self.training, act_ok=True, rows=xa.size(0))

    def reparameterize(self, mean, log_std):
        ml = join_head(mean, log_std)
        return Fn.reparameterize(ml, mean.size(1), self.n_samples)

    def kl_divergence(self, mean, log_std):
        return Fn.kl_divergence(join_head(mean, log_std), mean.size(1))

    def _device(self):
        return next(self.parameters()).device


class SkipconVAE(_VAEBase):
    """gvae/gvae_skipcon.py:53-156: three SAGE levels, mean/log_std head, concat-skip decoder."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, dropout_prob=0.5):
        super().__init__()
        h, z, p = hidden_channels, out_channels, dropout_prob
        self.dropout_prob, self.n_samples = p, n_samples
        self.in_channels, self.hidden_channels, self.out_channels = in_channels, h, z
        self.encoder1 = nn.Sequential(SAGEConv(in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.encoder2 = self._make_level2(h, p)
        self.encoder3 = nn.Sequential(SAGEConv(h, h), nn.Dropout(p), nn.BatchNorm1d(h), SAGEConv(h, 2 * z))
        self.decoder1 = nn.Sequential(nn.Linear(z, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.decoder2 = nn.Sequential(nn.Linear(2 * h, h), nn.Dropout(p), nn.BatchNorm1d(h))
        self.decoder3 = nn.Sequential(nn.Linear(2 * h, h), nn.Dropout(p), nn.BatchNorm1d(h), nn.Linear(h, in_channels))
        self._conv1_memo = None

    @contextlib.contextmanager
    def share_first_conv(self):
        """Inside this context the first convolution is evaluated once for a pair of encode passes.

        The reference encodes every batch twice per step (forward, then ``model.encode`` again for the KL
        term, gvae/gvae_skipcon.py:166-171).  Nothing stochastic precedes the first SAGEConv, so its
        output is the same tensor in both passes: the second pass reuses it (and the autograd node,
        so the convolution is also differentiated once, on the summed gradient).  The sharing is
        scoped to this context (one loss evaluation, no parameter update inside) and whatever is
        still pending is dropped on exit; outside it every call computes the convolution.
        In training the first pass already runs the second pass's dropout and batch norm of that
        level (one read of the convolution output for both, running statistics updated twice), so
        the context is for exactly the reference's pair of passes over one batch.
        """
        prev, self._conv1_memo = self._conv1_memo, []
        try:
            yield
        finally:
            self._conv1_memo = prev

    def _level1(self, x, g):
        """First encoder level -> (x1, skip1).  Inside :meth:`share_first_conv` the first call also prepares the second
        pass: in training with counter-based noise both dropout draws and both batch norms are computed over one read
        of the convolution output (``Fn.bn_act_pair``) and the second pass only collects its pair; otherwise the
        convolution output itself is kept and the second pass runs its own relu -> dropout -> batch norm on it."""
        conv, drop, bn = self.encoder1[0], self.encoder1[1], self.encoder1[2]
        memo = self._conv1_memo
        if memo is None:
            return self._sage_level(self.encoder1, x, g, fork=True)
        if memo and memo[0] is x and memo[1] is g and memo[2] == torch.is_grad_enabled():
            kind, kept = memo[3]
            memo.clear()
            return kept if kind == "levels" else self._finish(kept, drop, bn, relu=True, fork=True)
        if self.training and Fn.noise_is_counter_based() and type(conv) is SAGEConv:
            # a SAGEConv is one projection operator (see _sage_level): handed over as a callable, its product may store both
            # draws of the level itself
            pair, y = Fn.bn_level_pair(lambda: conv(x, g), bn, True, drop.p, rows=x.size(0))
        else:
            y = conv(x, g)
            pair = Fn.bn_level_pair(y, bn, True, drop.p) if self.training and Fn.noise_is_counter_based() else None
        if pair is not None:
            done = None
            if _two_streams(y.size(0)):  # the second pass may start here, next to the rest of the first
                done = torch.cuda.Event()
                done.record()
            memo[:] = [x, g, torch.is_grad_enabled(), ("levels", pair[1]), done]
            return pair[0]
        y, y_again = Fn.fork(y)
        memo[:] = [x, g, torch.is_grad_enabled(), ("preact", y_again)]
        return self._finish(y, drop, bn, relu=True, fork=True)

    def _make_level2(self, h, p):
        return nn.Sequential(SAGEConv(h, h), nn.Dropout(p), nn.BatchNorm1d(h))

    def _level2(self, x1, g):
        return self._sage_level(self.encoder2, x1, g, fork=True)

    def encode(self, x, edge_index):
        g = graph_of(edge_index, x.size(0))
        x = padded_input(x)
        # x1 / x2 feed the next level and, as skips, the decoder: the level's epilogue hands out two handles and its
        # backward pass adds the two incoming gradients inside the kernels
        x1, skip1 = self._level1(x, g)
        x2, skip2 = self._level2(x1, g)
        x3 = self._sage_level(self.encoder3, x2, g)
        mean, log_std = split_head(self.encoder3[3](x3, g), self.out_channels)
        return skip1, skip2, mean, log_std

    def decode(self, z, skip1, skip2):
        d1, d2, d3 = self.decoder1, self.decoder2, self.decoder3
        rows, h = z.size(0), d1[0].out_features
        Fn.start_bits_ahead([(rows, h, z.size(1), 0, False, d1[1].p), (rows, h, h, skip1.size(1), False, d2[1].p),
                             (rows, h, h, skip2.size(1), False, d3[1].p)], self.training, z.device)
        try:
            h1 = self._linear_level(z, None, d1[0], d1[1], d1[2], relu=False)
            h2 = self._linear_level(h1, skip1, d2[0], d2[1], d2[2], relu=False)
            h3 = self._linear_level(h2, skip2, d3[0], d3[1], d3[2], relu=False)
        finally:
            Fn.drop_bits_ahead()
        return Fn.dual_linear(h3, d3[3].weight, None, None, d3[3].bias)

    def forward(self, x, edge_index, batch=None):
        skip1, skip2, mean, log_std = self.encode(x, edge_index)
        z = self.reparameterize(mean, log_std)
        return self.decode(z, skip1, skip2), mean, log_std

    def recon_loss(self, x, reconstruction):
        return Fn.mse_loss(reconstruction, padded_input(x))


class SkipconAttentionVAE(SkipconVAE):
    """gvae/gvae_skipcon_attention.py:53-154: level 2 is relu(GATConv) -> BatchNorm1d; dropout 0.49."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, heads=1):
        self.heads = heads
        super().__init__(in_channels, hidden_channels, out_channels, n_samples, dropout_prob=0.49)

    def _make_level2(self, h, p):
        return nn.Sequential(GATConv(h, h, heads=self.heads, dropout=p), nn.BatchNorm1d(h * self.heads))

    def _level2(self, x1, g):
        return self._finish(self.encoder2[0](x1, g), None, self.encoder2[1], relu=True, fork=True)


class PoolGraph(nn.Module):
    """SAGPooling wrapper returning the coarse graph (gvae/gvae_skipcon.py:20-27)."""

    def __init__(self, num_node_features, GNN=GraphConv):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, GNN=GNN)

    def forward(self, x, edge_index, batch):
        x, edge_index, _, batch, _, _ = self.pool1(x, edge_index, batch=batch)
        return Data(x=x, edge_index=edge_index, batch=batch)


class PoolUnpoolGraph(nn.Module):
    """Pool, then map the coarse graph back onto original node ids (autoencoder/vae_with_pool.py:33-52)."""

    def __init__(self, num_node_features, GNN=GraphConv):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, GNN=GNN)

    def forward(self, x, edge_index, batch):
        x_cg, ei_cg, _, _, perm, _ = self.pool1(x, edge_index, batch=batch)
        new_batch = Fn.gather_i64(batch, perm)
        x_unpooled = Fn.gather_rows(x, perm)
        ei_unpooled = Fn.gather_i64(perm, ei_cg)
        return x_unpooled, x_cg, ei_unpooled, new_batch


class PoolVAE(_VAEBase):
    """autoencoder/vae_with_pool.py:55-143: two SAGE levels, residual-add decoder, pool/unpool in forward."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10):
        super().__init__()
        h, z, p = hidden_channels, out_channels, 0.5
        self.dropout_prob, self.n_samples, self.out_channels = p, n_samples, z
        self.in_channels, self.hidden_channels = in_channels, h
        self.encoder1 = nn.ModuleList([SAGEConv(in_channels, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.encoder2 = nn.ModuleList([SAGEConv(h, 2 * h), nn.Dropout(p), nn.BatchNorm1d(2 * h), SAGEConv(2 * h, 2 * z)])
        self.decoder1 = nn.ModuleList([nn.Linear(z, h), nn.Dropout(p), nn.BatchNorm1d(h)])
        self.decoder2 = nn.ModuleList([nn.Linear(h, in_channels)])
        self._make_pools(h)

    def _make_pools(self, h):
        self.pool_graph = PoolGraph(h)
        self.pool_unpool_graph = PoolUnpoolGraph(h)

    def encode(self, x, edge_index):
        g = graph_of(edge_index, x.size(0))
        x = padded_input(x)
        x1 = self._sage_level(self.encoder1, x, g)
        x2 = self._sage_level(self.encoder2, x1, g)
        mean, log_std = split_head(self.encoder2[3](x2, g), self.out_channels)
        return x1, mean, log_std

    def decode(self, z, x1):
        d1 = self.decoder1
        h = self._linear_level(z, None, d1[0], d1[1], d1[2], relu=True)
        # (h + x1) W^T as one K-concatenated product h W^T + x1 W^T: the sum is never written
        w = self.decoder2[0].weight
        return Fn.dual_linear(h, w, x1, w, self.decoder2[0].bias)

    def forward(self, x, edge_index, batch):
        x1, mean, log_std = self.encode(x, edge_index)
        z = self.reparameterize(mean, log_std)
        reconstruction = self.decode(z, x1)
        x_unpooled, _, ei_unpooled, _ = self.pool_unpool_graph(x1, edge_index, batch)
        return reconstruction, x_unpooled, ei_unpooled

    def recon_loss(self, x, mean, log_std, edge_index, x1):
        return Fn.mse_loss(self.decode(self.reparameterize(mean, log_std), x1), padded_input(x))


class CoarsenGraph(nn.Module):
    """GCNConv-scored SAGPooling returning (x_cg, edge_index_cg, n_cg) (autoencoder/vae_custom_pool.py:94-106).

    The reference's pooling class also evaluates two dense-Laplacian Fiedler values per call
    (vae_custom_pool.py:63-76); their difference lands in a tuple slot this wrapper drops, so it
    never reaches an output.  ``spectral_probe=True`` evaluates it anyway (:mod:`hgvae_amd.spectral`: the dense
    Laplacian from one device pass, then the library's symmetric eigensolver, O(N^3); off by default).
    """

    def __init__(self, num_node_features, alpha=0.5, spectral_probe=False):
        super().__init__()
        self.pool1 = SAGPooling(num_node_features, ratio=0.5, GNN=GCNConv)
        self.alpha, self.spectral_probe = alpha, spectral_probe
        self.last_spectral_loss = None

    @staticmethod
    def fiedler_value(edge_index, num_nodes):
        """Second-smallest eigenvalue of I - D^-1/2 A D^-1/2, degree over SOURCES (vae_custom_pool.py:19-53)."""
        from .spectral import fiedler_value
        return fiedler_value(edge_index, num_nodes)

    def forward(self, x, edge_index, batch):
        x_cg, ei_cg, _, _, _, _ = self.pool1(x, edge_index, batch=batch)
        if self.spectral_probe:
            raw = edge_index.edge_index if not isinstance(edge_index, torch.Tensor) else edge_index
            self.last_spectral_loss = self.alpha * (self.fiedler_value(raw, x.size(0))
                                                    - self.fiedler_value(ei_cg, x_cg.size(0))).abs()
        return x_cg, ei_cg, x_cg.size(0)


class CustomPoolVAE(PoolVAE):
    """autoencoder/vae_custom_pool.py:109-205: concat decoder with x1_transform, beta-weighted KL."""

    def __init__(self, in_channels, hidden_channels, out_channels, n_samples=10, beta=0.5):
        super().__init__(in_channels, hidden_channels, out_channels, n_samples)
        self.beta = beta
        self.decoder1[0] = nn.Linear(out_channels + in_channels, hidden_channels)
        self.x1_transform = nn.Linear(hidden_channels, in_channels)

    def _make_pools(self, h):
        self.pool_graph = CoarsenGraph(h)
        self.pool_unpool_graph = CoarsenGraph(h)

    def decode(self, z, x1):
        zt = self.x1_transform
        x1_t = Fn.dual_linear(x1, zt.weight, None, None, zt.bias)
        d1 = self.decoder1
        h = self._linear_level(z, x1_t, d1[0], d1[1], d1[2], relu=True)
        return Fn.dual_linear(h, self.decoder2[0].weight, None, None, self.decoder2[0].bias)

    def forward(self, x, edge_index, batch):
        x1, mean, log_std = self.encode(x, edge_index)
        z = self.reparameterize(mean, log_std)
        reconstruction = self.decode(z, x1)
        x_cg, ei_cg, _ = self.pool_unpool_graph(x1, edge_index, batch)
        return reconstruction, x_cg, ei_cg, mean, log_std, x1

    def recon_loss(self, x_recon, x_original, mean, log_std, edge_index, x1):
        """Re-samples z and re-decodes; ``x_recon`` is ignored, as in the reference (:196-200)."""
        return Fn.mse_loss(self.decode(self.reparameterize(mean, log_std), x1), padded_input(x_original))


def train_vae_custom_pool(model, loader, optimizer, clip_value=None, device=None):
    """autoencoder/vae_custom_pool.py:208-232: recon (re-sampled) + beta * kl, optional grad-norm clipping."""
    if device is not None:
        model.to(device)
    model.train()
    dev = model._device()
    total = torch.zeros((), dtype=torch.float64, device=dev)
    for data in loader:
        data = _to_device(data, dev)
        optimizer.zero_grad()
        loss = custom_pool_step_loss(model, data.x, data.edge_index, data.batch)
        loss.backward()
        if clip_value is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip_value)
        optimizer.step()
        total += loss.detach()
    return float(total.item()) / len(loader)


def custom_pool_step_loss(model, x, edge_index, batch):
    Fn.no_branches()
    with Fn.weight_images(model), Fn.deferred_levels():
        with Fn.bounded_pooling():      # the coarse graph is discarded here: no need to read its size back
            reconstruction, _, _, mean, log_std, x1 = model(x, edge_index, batch)
        recon_loss = model.recon_loss(reconstruction, x, mean, log_std, edge_index, x1)
    return torch.add(recon_loss, model.kl_divergence(mean, log_std), alpha=float(model.beta))


def _to_device(data, device):
    for name in ("x", "edge_index", "batch", "edge_attr"):
        t = getattr(data, name, None)
        if isinstance(t, torch.Tensor) and t.device != device:
            setattr(data, name, t.to(device))
    return data


def _edges(model, data):
    """The batch's ready-made Graph when the loader supplies one (hgvae_amd.shard.GraphShard) and the model
    takes it in place of edge_index; else the edge_index tensor."""
    g = getattr(data, "graph", None)
    return g if g is not None and isinstance(model, SkipconVAE) else data.edge_index


def train_vae(model, loader, optimizer, lambda_kl=1.0):
    """One epoch; per batch: forward, SECOND encode, mse + lambda*kl, backward, step
    (gvae/gvae_skipcon.py:159-187).  The running loss stays on the device; one host sync per epoch."""
    model.train()
    dev = model._device()
    total = torch.zeros((), dtype=torch.float64, device=dev)
    for data in loader:
        data = _to_device(data, dev)
        optimizer.zero_grad()
        loss = train_step_loss(model, data.x, _edges(model, data), data.batch, lambda_kl)
        loss.backward()
        optimizer.step()
        total += loss.detach()
    return float(total.item()) / len(loader)


# The second encode pass on a stream of its own: "capture" (default) while a HIP graph is being captured - the two
# passes become parallel branches of the graph, which a batch of a few thousand nodes needs to fill the chip (its
# kernels occupy a fraction of the CUs each: captured chignolin step 0.85 -> 0.71 ms) - "always", or "never".  Eager
# steps of that size are bound by the host, and large batches fill the chip with every kernel (measured: no gain).
_branch_streams = os.environ.get("HGVAE_BRANCH_STREAMS", "capture")
_side_streams: dict = {}


def set_branch_streams(mode: str) -> None:
    global _branch_streams
    if mode not in ("capture", "always", "never"):
        raise ValueError("branch streams: 'capture', 'always' or 'never'")
    _branch_streams = mode


BRANCH_MAX_ROWS = 65536      # beyond this every kernel fills the chip by itself and the branch only adds bookkeeping


def _two_streams(n_rows: int) -> bool:
    if _branch_streams == "never":
        return False
    if _branch_streams == "always":
        return True
    return n_rows <= BRANCH_MAX_ROWS and torch.cuda.is_current_stream_capturing()


def _side_stream(device):
    s = _side_streams.get(device.index)
    if s is None:
        s = _side_streams[device.index] = torch.cuda.Stream(device)
    return s


def train_step_loss(model, x, edge_index, batch=None, lambda_kl=1.0):
    """Loss of one reference train step (no backward)."""
    share = getattr(model, "share_first_conv", None)
    Fn.no_branches()
    # the loss looks at the reconstruction only: pooled graphs (pool variants) may stay at their capacity
    with share() if share is not None else contextlib.nullcontext(), Fn.bounded_pooling(), Fn.weight_images(model), \
            Fn.deferred_levels():
        out = model(x, edge_index, batch)
        memo = getattr(model, "_conv1_memo", None)
        if memo and len(memo) > 4 and memo[4] is not None:
            # second pass as a branch of its own: it needs the paired first level only.  Its operators allocate from
            # the side stream's pool and its backward nodes run there (autograd keeps a node on its forward stream)
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_event(memo[4])
            for t in memo[3][1]:
                t.record_stream(side)
            with torch.cuda.stream(side), Fn.branch_of(main):
                enc = model.encode(x, edge_index)
            for t in enc:
                t.record_stream(main)
            main.wait_stream(side)
            Fn.apply_branch_updates()
        else:
            enc = model.encode(x, edge_index)
    mean, log_std = enc[-2], enc[-1]
    recon_loss = Fn.mse_loss(out[0], padded_input(x))
    kl_loss = model.kl_divergence(mean, log_std)
    # one launch forward, and none backward for lambda = 1 (a separate product would cost one each way)
    return torch.add(recon_loss, kl_loss, alpha=float(lambda_kl))


def validate_vae(model, loader):
    """gvae/gvae_skipcon.py:190-204: eval mode, same double pass, recon + kl."""
    model.eval()
    dev = model._device()
    total = torch.zeros((), dtype=torch.float64, device=dev)
    with torch.no_grad():
        for data in loader:
            data = _to_device(data, dev)
            total += train_step_loss(model, data.x, _edges(model, data), data.batch, 1.0)
    return float(total.item()) / len(loader)


def test_vae_rmsd(model, loader):
    """gvae/gvae_skipcon.py:218-233: RMSD between the first three columns of input and reconstruction."""
    model.eval()
    dev = model._device()
    total = torch.zeros((), dtype=torch.float64, device=dev)
    with torch.no_grad():
        for data in loader:
            data = _to_device(data, dev)
            recon = model(data.x, data.edge_index, data.batch)[0]
            total += compute_rmsd(data.x[:, :3], recon[:, :3])
    return float(total.item()) / len(loader)


test_vae_rmsd.__test__ = False  # not a pytest test

def node_embeddings(model, x, edge_index):
    """Per-node latent embeddings of one graph for the clustering step of the reference
    (gvae/gvae_skipcon.py:333-339: eval mode, no grad, ``encode`` then ``reparameterize``); returns the
    [N, Z] tensor on the model's device (the reference moves it to numpy for scipy / sklearn)."""
    was_training = model.training
    model.eval()
    dev = model._device()
    try:
        with torch.no_grad():
            enc = model.encode(x.to(dev), edge_index if isinstance(edge_index, Graph) else edge_index.to(dev))
            return model.reparameterize(enc[-2], enc[-1]).contiguous()
    finally:
        model.train(was_training)
