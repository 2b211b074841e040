"""Data-parallel training: one process per GPU, one flat fp32 gradient bucket, one all-reduce.

The reference is single-process (no distributed code anywhere); graphs of a mini-batch are
disjoint components, so every kernel is rank-local and the only exchange is the gradient
sum after backward.  The whole model is ~0.2-0.55 M parameters (0.8-2.2 MB): one contiguous
``all_reduce`` (RCCL over xGMI with the ``nccl`` backend, ``gloo`` in CPU tests) is
latency-bound and needs no overlap with backward.

Parameters that received no gradient (e.g. the SAGPooling scorer, whose output the
reference's train loop discards, autoencoder/vae_with_pool.py:154) contribute zeros.
BatchNorm statistics stay rank-local (statistics over the nodes of the local shard).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from torchrun's environment. Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        n_dev = torch.cuda.device_count()
        if backend is None:
            # RCCL needs one GPU per rank.  With fewer GPUs than ranks (rehearsal on a 1-GPU box) the ranks
            # share the devices and the collectives run over gloo on host copies (HGVAE_DIST_BACKEND overrides).
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            backend = os.environ.get("HGVAE_DIST_BACKEND") or ("nccl" if n_dev >= local_world else "gloo")
        if n_dev > 0:
            local_rank = local_rank % n_dev
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        # per-rank dropout / eps streams: the Philox counters are (row, chunk, offset), so ranks sharing one
        # seed would draw identical noise for the same positions of their shards
        from . import noise
        noise.manual_seed(noise.DEFAULT_SEED + rank)
    return rank, world, local_rank


def barrier(group=None) -> None:
    """dist.barrier pinned to this rank's GPU under the RCCL backend (no-op for one rank)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if dist.get_backend(group) == "nccl":
        dist.barrier(group=group, device_ids=[torch.cuda.current_device()])
    else:
        dist.barrier(group=group)


class FlatGradBucket:
    """Owns one contiguous buffer holding every parameter's gradient."""

    def __init__(self, params: Iterable[torch.nn.Parameter], process_group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
        self.views, off = [], 0
        for p, n in zip(self.params, sizes):
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.branch_flat: Optional[torch.Tensor] = None
        self.attach()

    def enable_branch(self) -> None:
        """A second flat buffer for the part of a step that runs on a branch stream next to the rest (the second
        encode pass of a captured step): its kernels add weight gradients there instead of racing with the trunk's
        adds into ``p.grad``; :meth:`merge_branch` folds it into the bucket after backward."""
        if self.branch_flat is None:
            self.branch_flat = torch.zeros_like(self.flat)
            self.branch_views, off = [], 0
            for p in self.params:
                n = p.numel()
                self.branch_views.append(self.branch_flat[off:off + n].view_as(p))
                off += n
        # (re-)claim the parameters: a second bucket over the same parameters points them at its own buffer, so the
        # bucket whose step runs next calls this again (GraphedTrainStep does at the start of every evaluation)
        for p, v in zip(self.params, self.branch_views):
            if getattr(p, "_hgv_branch_grad", None) is not v:
                p._hgv_branch_grad = v

    def merge_branch(self) -> None:
        if self.branch_flat is not None:
            self.flat.add_(self.branch_flat)

    def attach(self) -> None:
        """Point every ``p.grad`` at its slice of the flat buffer (autograd then accumulates in place)."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self) -> None:
        """Replacement for ``optimizer.zero_grad()`` that keeps the views alive."""
        self.flat.zero_()
        if self.branch_flat is not None:
            self.branch_flat.zero_()
        self.attach()

    def gather(self) -> None:
        """Make the flat buffer hold the current gradients (zeros for parameters without one)."""
        src, dst = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                src.append(g)
                dst.append(v)
            p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def all_reduce_mean(self) -> None:
        """Sum the bucket over ranks and divide by the world size (no-op for one rank)."""
        self.gather()
        if self.world > 1:
            all_reduce_any(self.flat, dist.ReduceOp.SUM, self.group)
            self.flat.div_(self.world)


def all_reduce_any(t: torch.Tensor, op=dist.ReduceOp.SUM, group=None) -> torch.Tensor:
    """In-place all-reduce of ``t``.  Device tensors go straight to RCCL; under the gloo backend (CPU tests,
    several ranks sharing one GPU) they are reduced through a host copy, because gloo's own handling of device
    tensors is not available in every build and never returns when it is missing."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) != "nccl":
        host = t.detach().cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def all_reduce_scalar_mean(t: torch.Tensor, group=None) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = all_reduce_any(t.clone(), group=group)
        t /= dist.get_world_size(group)
    return t


def save_checkpoint(model: torch.nn.Module, path: str, bn_buffers: str = "rank0", group=None) -> None:
    """``torch.save(model.state_dict(), path)`` of the reference (gvae/gvae_skipcon_attention.py:302) for a
    data-parallel job: every rank calls it, rank 0 writes the file, all ranks leave together.

    Parameters are identical on every rank (same initial weights, averaged gradients).  BatchNorm running statistics
    are not: each rank tracks the nodes of its own shards.  ``bn_buffers`` states which ones the file holds:
    ``"rank0"`` (default) rank 0's buffers as they are - what a single-process run over rank 0's shards would have
    saved; ``"mean"`` the average over the ranks of every floating-point buffer (running_mean, running_var), which for
    equal shard sizes is the running estimate over all shards up to the between-rank spread of the means;
    ``num_batches_tracked`` is the same number on every rank.  The model itself is not modified.  The file loads with
    ``model.load_state_dict(torch.load(path, weights_only=True))`` on any number of ranks, and in the reference."""
    if bn_buffers not in ("rank0", "mean"):
        raise ValueError("bn_buffers: 'rank0' or 'mean'")
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if multi and bn_buffers == "mean":
        names = {n for n, b in model.named_buffers() if b.is_floating_point()}
        for k in sorted(names):                                  # same order on every rank
            state[k] = all_reduce_scalar_mean(state[k], group)
    if not multi or dist.get_rank(group) == 0:
        tmp = f"{path}.tmp{os.getpid()}"
        torch.save({k: v.cpu() for k, v in state.items()}, tmp)
        os.replace(tmp, path)                                    # a reader never sees a half-written file
    barrier(group)


def load_checkpoint(model: torch.nn.Module, path: str) -> None:
    """``model.load_state_dict(torch.load(path))`` (gvae/gvae_skipcon.py:327) with a loader that executes nothing
    from the file; every rank reads the same file."""
    dev = next(model.parameters()).device
    model.load_state_dict(torch.load(path, map_location=dev, weights_only=True))


def shard_range(num_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced split of ``num_items`` graphs over ``world`` ranks."""
    base, rem = divmod(num_items, world)
    lo = rank * base + min(rank, rem)
    return range(lo, lo + base + (1 if rank < rem else 0))


def balanced_ranges(costs: Sequence[int], world: int) -> list:
    """Contiguous split of graphs with unequal sizes: ``world`` ranges (some possibly empty) over ``costs`` whose
    largest total is as small as a contiguous split allows.  ``costs[i]`` is the work of graph i - its edge count
    for a message-passing step.  Every rank calls this with the same list and takes ``[rank]``, so the partition
    needs no communication; graphs keep their order, which keeps a rank's nodes contiguous in the collated batch.
    """
    costs = [int(c) for c in costs]
    if world < 1:
        raise ValueError("world must be >= 1")
    if any(c < 0 for c in costs):
        raise ValueError("costs must be non-negative")

    def cut(limit):
        """Greedy ranges with totals <= limit, or None when more than ``world`` are needed."""
        out, lo, acc = [], 0, 0
        for i, c in enumerate(costs):
            if acc + c > limit and i > lo:
                out.append(range(lo, i))
                lo, acc = i, 0
            acc += c
        out.append(range(lo, len(costs)))
        return out if len(out) <= world else None

    lo, hi = max(costs, default=0), sum(costs)
    while lo < hi:                       # smallest feasible bound on a rank's total
        mid = (lo + hi) // 2
        if cut(mid) is None:
            lo = mid + 1
        else:
            hi = mid
    parts = cut(lo)
    # the greedy cut may need fewer ranges than there are ranks: split the heaviest splittable range at its best
    # point until every rank has work (a split never raises the largest total)
    total = lambda r: sum(costs[i] for i in r)
    while len(parts) < world:
        wide = [k for k in range(len(parts)) if len(parts[k]) > 1]
        if not wide:
            break
        k = max(wide, key=lambda k: total(parts[k]))
        r, best, acc, whole = parts[k], None, 0, total(parts[k])
        for m in range(r.start + 1, r.stop):
            acc += costs[m - 1]
            worst = max(acc, whole - acc)
            if best is None or worst < best[0]:
                best = (worst, m)
        parts[k:k + 1] = [range(r.start, best[1]), range(best[1], r.stop)]
    n = len(costs)
    parts += [range(n, n)] * (world - len(parts))
    return parts
