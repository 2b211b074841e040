"""Whole-train-step HIP graph: capture once, replay per step.

One reference train step (gvae/gvae_skipcon.py:163-185) on a fixed batch is ~450 kernel
launches; at the reference's own batch sizes (a few thousand nodes) the GPU work per launch is
a few microseconds and the host launch path dominates.  ``GraphedTrainStep`` records
``zero grads -> forward -> second encode -> loss -> backward (-> optimizer step)`` into a HIP
graph on a side stream and replays it; dropout masks and normal draws stay fresh on every replay
because the kernels add a device-side step counter to their Philox offsets.

The pooling variants are capturable as well: inside their loss functions the coarse graph stays at
its capacity (``functional.bounded_pooling``: no read-back of the surviving edge count), and the sizes
that depend only on the batch vector are cached per batch tensor during the warm-up.
"""
from __future__ import annotations

import warnings
from typing import Callable, Optional

import torch

from .distributed import FlatGradBucket
from .functional import direct_weight_grads, end_branches
from .graph import graph_of
from .noise import PhiloxNoise, use_noise
from .vae import train_step_loss


class GraphedTrainStep:
    """Captured train step on a fixed batch.

    ``optimizer`` is stepped inside the graph when it was built with ``capturable=True`` and the
    job is single-process; otherwise (and always for world size > 1, where the gradient
    all-reduce sits between backward and the update) it is stepped eagerly after the replay.
    """

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, x: torch.Tensor,
                 edge_index: torch.Tensor, batch: Optional[torch.Tensor] = None, lambda_kl: float = 1.0,
                 loss_fn: Callable = train_step_loss, bucket: Optional[FlatGradBucket] = None, seed: int = 0,
                 warmup: int = 3):
        self.model, self.optimizer = model, optimizer
        self.x, self.edge_index, self.batch, self.lambda_kl, self.loss_fn = x, edge_index, batch, lambda_kl, loss_fn
        if bucket is None:                # an optimizer that owns a flat bucket (FlatAdam) reads its gradients there
            bucket = getattr(optimizer, "bucket", None)
        self.bucket = bucket if bucket is not None else FlatGradBucket(model.parameters())
        self.bucket.enable_branch()
        dev = x.device
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.noise = PhiloxNoise(seed, device_counter=self.counter)
        graph_of(edge_index, x.size(0))  # CSR built outside the capture
        capturable = all(g.get("capturable", False) for g in optimizer.param_groups)
        self.optimizer_in_graph = capturable and self.bucket.world == 1
        self.loss = None
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # Warm-up allocates workspaces and weight images outside the capture.  It must not train:
            # parameters, optimizer state and BatchNorm buffers are as before it when the capture starts.
            from .optim import FlatAdam
            lazy_state = not isinstance(optimizer, FlatAdam)    # torch optimizers create their state in step()
            keep = [(t, t.clone()) for t in list(model.buffers()) + (list(model.parameters()) if lazy_state else [])]
            # the optimizer may already have trained (an earlier capture, eager steps, load_state_dict): what it holds
            # now is put back after the throw-away steps; only state that those steps created starts from zero
            state_before = {p: {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                            for p, st in optimizer.state.items()} if lazy_state else {}
            for _ in range(warmup):
                self._body(step_optimizer=lazy_state, reduce=True)
            with torch.no_grad():
                for t, saved in keep:
                    t.copy_(saved)
                if lazy_state:
                    for p, st in optimizer.state.items():
                        old = state_before.get(p, {})
                        for k, v in list(st.items()):
                            if k in old:
                                if isinstance(v, torch.Tensor):
                                    v.copy_(old[k])
                                else:
                                    st[k] = old[k]
                            elif isinstance(v, torch.Tensor):      # created by the warm-up: "no step taken"
                                v.zero_()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self._body(step_optimizer=self.optimizer_in_graph, reduce=False)
        torch.cuda.synchronize(dev)

    def _body(self, step_optimizer: bool, reduce: bool):
        self.noise.advance()
        self.bucket.enable_branch()       # another bucket over the same parameters may have claimed the branch views since
        self.bucket.zero()
        with use_noise(self.noise):
            loss = self.loss_fn(self.model, self.x, self.edge_index, self.batch, self.lambda_kl)
        with direct_weight_grads(), warnings.catch_warnings():
            # the second encode pass is a branch on a stream of its own under capture; its weight gradients reach the
            # accumulation nodes of the capture stream from there, on purpose
            warnings.filterwarnings("ignore", message="The AccumulateGrad node's stream does not match")
            loss.backward()
        end_branches()                    # branch membership ends with the step that created it
        self.bucket.merge_branch()
        if reduce:
            self.bucket.all_reduce_mean()
        if step_optimizer:
            self.optimizer.step()
        return loss.detach()

    def __call__(self) -> torch.Tensor:
        """Run one step; returns the (device-resident, overwritten on the next call) loss."""
        self.graph.replay()
        if not self.optimizer_in_graph:
            self.bucket.all_reduce_mean()
            self.optimizer.step()
        return self.loss


class GraphedEvalStep:
    """Captured evaluation of the loss on a fixed batch structure (``validate_vae``'s body, gvae/gvae_skipcon.py:190-204:
    eval mode, no gradients, forward + second encode + recon + kl); fresh eps per replay through the device-side counter."""

    def __init__(self, model: torch.nn.Module, x: torch.Tensor, edge_index: torch.Tensor, batch: Optional[torch.Tensor] = None,
                 lambda_kl: float = 1.0, loss_fn: Callable = train_step_loss, seed: int = 0, warmup: int = 2):
        self.model, self.x, self.edge_index, self.batch, self.lambda_kl, self.loss_fn = model, x, edge_index, batch, lambda_kl, loss_fn
        dev = x.device
        self.counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self.noise = PhiloxNoise(seed, device_counter=self.counter)
        graph_of(edge_index, x.size(0))
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.cuda.graph(self.graph, stream=side):
            self.loss = self._body()
        torch.cuda.synchronize(dev)

    def _body(self):
        self.noise.advance()
        with torch.no_grad(), use_noise(self.noise):
            return self.loss_fn(self.model, self.x, self.edge_index, self.batch, self.lambda_kl).detach()

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss


class CapturedTraining:
    """``train_vae`` for loaders whose batches repeat a structure, replayed as captured steps.

    The reference trains on MD frames of one molecule (gvae/gvae_skipcon.py:262-263 feeds
    gvae/create_graphs_for_gvae.py graphs: the same atoms and consecutive-ATOM edges in every frame, new
    coordinates and forces), 32 graphs of 138 nodes per batch: ~140 launches of a few microseconds each, so an eager
    step is bound by the host (2.2 ms against 0.75 ms of GPU work).  Here the step is captured once per batch
    STRUCTURE - (edge_index, batch vector) - and every later batch of that structure only copies its features into the
    captured input and replays.  Structures are recognised by ``data.structure_id`` when the loader provides one
    (:class:`hgvae_amd.shard.GraphShard` does for data sets whose graphs all share one topology), otherwise by comparing
    ``edge_index`` / ``batch`` with the captured ones on the device (one small read-back per batch).  At most
    ``max_structures`` structures are captured; batches beyond that run eagerly, as do all batches of a data set of
    unequal graphs - results are the same either way, only the launch path differs.

    Keep the object across epochs.  ``optimizer`` steps inside the graph when it is capturable
    (``FlatAdam(..., capturable=True)``, ``torch.optim.Adam(..., capturable=True)``), otherwise right after each replay.
    """

    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, lambda_kl: float = 1.0,
                 loss_fn: Callable = train_step_loss, max_structures: int = 4, seed: int = 0):
        self.model, self.optimizer, self.lambda_kl, self.loss_fn = model, optimizer, lambda_kl, loss_fn
        self.max_structures, self.seed = max_structures, seed
        self._entries: list = []            # (structure_id or None, edge_index, batch, GraphedTrainStep)
        self._eval_entries: list = []       # the same for evaluation steps (GraphedEvalStep)
        self._bucket = getattr(optimizer, "bucket", None)
        self.captured_steps = self.eager_steps = 0
        # ranks draw different dropout / eps streams (the Philox counters are (row, chunk, offset): one seed on every
        # rank would repeat the same noise at the same positions of every shard)
        import torch.distributed as dist
        self._rank_term = 1_000_003 * (dist.get_rank() if dist.is_available() and dist.is_initialized() else 0)

    def _find(self, data, entries=None):
        sid = getattr(data, "structure_id", None)
        shape = (tuple(data.x.shape), tuple(data.edge_index.shape))
        for ent in (self._entries if entries is None else entries):
            esid, ei, bt, step = ent
            if (tuple(step.x.shape), tuple(ei.shape)) != shape:
                continue
            if sid is not None and esid is not None:
                if sid == esid:
                    return step
                continue
            same = torch.equal(ei, data.edge_index) and (bt is None) == (data.batch is None) and \
                (bt is None or torch.equal(bt, data.batch))
            if same:
                return step
        return None

    def _capture(self, data):
        from .graph import new_padded
        x = new_padded(data.x.size(0), data.x.size(1), data.x.device)      # 16-byte layout: used as it is, never re-padded
        x.copy_(data.x)
        ei = data.edge_index.clone()
        bt = data.batch.clone() if data.batch is not None else None
        step = GraphedTrainStep(self.model, self.optimizer, x, ei, bt, self.lambda_kl, loss_fn=self.loss_fn,
                                bucket=self._bucket, seed=self.seed + 7919 * len(self._entries) + self._rank_term)
        self._bucket = step.bucket
        self._entries.append((getattr(data, "structure_id", None), ei, bt, step))
        return step

    def step(self, data) -> torch.Tensor:
        """One train step on ``data`` (already on the model's device); returns the loss (device tensor, valid until the
        next step)."""
        step = self._find(data)
        if step is None and len(self._entries) < self.max_structures:
            step = self._capture(data)
        if step is None:
            self.eager_steps += 1
            # gradients must stay in the bucket's views: a captured step of this object (replayed before or after this
            # one) writes them there, and zero_grad() of a torch optimizer would detach p.grad from the views
            if self._bucket is not None:
                self._bucket.zero()
            else:
                self.optimizer.zero_grad()
            loss = self.loss_fn(self.model, data.x, data.edge_index, data.batch, self.lambda_kl)
            loss.backward()
            if self._bucket is not None:
                self._bucket.all_reduce_mean()
            self.optimizer.step()
            return loss.detach()
        self.captured_steps += 1
        step.x.copy_(data.x)
        return step()

    def train_epoch(self, loader) -> float:
        """Same contract as :func:`hgvae_amd.vae.train_vae`: mean loss of the epoch, one host synchronisation."""
        from .vae import _to_device
        self.model.train()
        dev = next(self.model.parameters()).device
        total = torch.zeros((), dtype=torch.float64, device=dev)
        n = 0
        for data in loader:
            data = _to_device(data, dev)
            total += self.step(data)
            n += 1
        return float(total.item()) / max(n, 1)

    def validate_epoch(self, loader, lambda_kl: float = 1.0) -> float:
        """Same contract as :func:`hgvae_amd.vae.validate_vae` (eval mode, no gradients, mean of recon + kl over the
        batches), with batches of a known structure replayed as captured evaluation steps.  The model is left in eval mode,
        as the reference's function leaves it."""
        from .graph import new_padded
        from .vae import _to_device
        self.model.eval()
        dev = next(self.model.parameters()).device
        total = torch.zeros((), dtype=torch.float64, device=dev)
        n = 0
        for data in loader:
            data = _to_device(data, dev)
            # a captured evaluation has its KL weight baked in: entries are looked up per weight
            entries = [e for e in self._eval_entries if e[3].lambda_kl == lambda_kl]
            step = self._find(data, entries)
            if step is None and len(self._eval_entries) < self.max_structures:
                x = new_padded(data.x.size(0), data.x.size(1), dev)
                x.copy_(data.x)
                ei = data.edge_index.clone()
                bt = data.batch.clone() if data.batch is not None else None
                step = GraphedEvalStep(self.model, x, ei, bt, lambda_kl, loss_fn=self.loss_fn,
                                       seed=self.seed + 104729 * (len(self._eval_entries) + 1) + self._rank_term)
                self._eval_entries.append((getattr(data, "structure_id", None), ei, bt, step))
            if step is None:
                with torch.no_grad():
                    total += self.loss_fn(self.model, data.x, data.edge_index, data.batch, lambda_kl)
            else:
                step.x.copy_(data.x)
                total += step()
            n += 1
        return float(total.item()) / max(n, 1)
