"""A dataset of graphs resident on the device as one packed block, and mini-batches gathered from it.

The reference's loader reads one file per graph and per epoch (``torch.load`` in
``CustomGraphDataset.__getitem__``, gvae/gvae_skipcon.py:247-259) and concatenates the batch in
Python (``collate_fn`` -> ``Batch.from_data_list``, :262-263); at its batch sizes the step is
input-bound.  Here the whole dataset is normalised once (same dataset-wide mean / unbiased std of
the selected columns, :240-245), stored as ONE disjoint union with its CSR in both directions, and
a batch is two gather launches (``hgv_collate_nodes`` / ``hgv_collate_edges``): features, ``batch``
vector, ``edge_index`` and the ready-made CSR of the batch, bit-identical to collating on the host
and sorting the batch's edges.
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .data import Batch, Data
from .graph import Graph, _Csr, new_padded, pad4


class GraphShard:
    """All graphs of a dataset on one device.  Build with :meth:`from_data_list`, persist with
    :meth:`save` / :meth:`load`, draw batches with :meth:`batch` or :meth:`loader`."""

    def __init__(self, x: torch.Tensor, edge_index: torch.Tensor, node_ptr: torch.Tensor, edge_ptr: torch.Tensor,
                 edge_attr: Optional[torch.Tensor] = None, mean: Optional[torch.Tensor] = None,
                 std: Optional[torch.Tensor] = None, numerical_indices: Optional[Sequence[int]] = None):
        _lib.require_cuda(x, edge_index)
        dev = x.device
        self.num_features = int(x.size(1))
        self.x = new_padded(x.size(0), self.num_features, dev)          # [Nt, F] view of a 16-byte aligned block
        self.x.copy_(x)
        self.edge_index = edge_index.contiguous()
        self.edge_attr = edge_attr.contiguous().to(torch.float32) if edge_attr is not None else None
        self._node_ptr_host = node_ptr.to("cpu", torch.int64).numpy().copy()
        self._edge_ptr_host = edge_ptr.to("cpu", torch.int64).numpy().copy()
        self.node_ptr = torch.from_numpy(self._node_ptr_host).to(dev)
        self.edge_ptr = torch.from_numpy(self._edge_ptr_host).to(dev)
        self.mean, self.std = mean, std
        self.numerical_indices = list(numerical_indices) if numerical_indices is not None else None
        if int(self._edge_ptr_host[-1]) != self.edge_index.size(1) or int(self._node_ptr_host[-1]) != x.size(0):
            raise _lib.HgvError("node_ptr / edge_ptr do not match the packed arrays")
        if self.edge_index.size(1) >= 2 ** 31 or x.size(0) >= 2 ** 31:
            raise _lib.HgvError("a shard holds at most 2^31 - 1 nodes and edges (int32 CSR); split the dataset into several shards")
        # CSR of the whole union, once: every batch takes its segments from here
        self.graph = Graph(self.edge_index, x.size(0))
        self.graph.by_src
        self.graph.check()

    # ------------------------------------------------------------------ construction / persistence
    @classmethod
    def from_data_list(cls, data_list: Sequence[Data], device, numerical_indices: Optional[Sequence[int]] = None):
        """Pack host graphs.  With ``numerical_indices`` those feature columns are standardised with the mean
        and unbiased std over all nodes of all graphs, as CustomGraphDataset does per item."""
        sizes = np.array([int(d.x.size(0)) for d in data_list], dtype=np.int64)
        ecnt = np.array([int(d.edge_index.size(1)) for d in data_list], dtype=np.int64)
        node_ptr = np.concatenate([[0], np.cumsum(sizes)])
        edge_ptr = np.concatenate([[0], np.cumsum(ecnt)])
        x = torch.cat([d.x.to(torch.float32) for d in data_list], dim=0)
        mean = std = None
        if numerical_indices is not None:
            idx = list(numerical_indices)
            feats = x[:, idx]
            mean, std = torch.mean(feats, dim=0), torch.std(feats, dim=0)
            x = x.clone()
            x[:, idx] = (feats - mean) / std
        ei = torch.cat([d.edge_index.to(torch.int64) + int(off) for d, off in zip(data_list, node_ptr[:-1])], dim=1)
        eas = [d.edge_attr for d in data_list]
        ea = torch.cat([e if e.dim() == 2 else e.unsqueeze(1) for e in eas], dim=0) if all(e is not None for e in eas) else None
        return cls(x.to(device), ei.to(device), torch.from_numpy(node_ptr), torch.from_numpy(edge_ptr),
                   ea.to(device) if ea is not None else None, mean, std, numerical_indices)

    def save(self, path: str) -> None:
        """Plain tensor dictionary (readable with ``torch.load(weights_only=True)``)."""
        obj = {"x": self.x.cpu().contiguous(), "edge_index": self.edge_index.cpu(),
               "node_ptr": torch.from_numpy(self._node_ptr_host), "edge_ptr": torch.from_numpy(self._edge_ptr_host)}
        if self.edge_attr is not None:
            obj["edge_attr"] = self.edge_attr.cpu()
        if self.mean is not None:
            obj["mean"], obj["std"] = self.mean.cpu(), self.std.cpu()
            obj["numerical_indices"] = torch.tensor(self.numerical_indices, dtype=torch.int64)
        torch.save(obj, path)

    @classmethod
    def load(cls, path: str, device) -> "GraphShard":
        obj = torch.load(path, map_location="cpu", weights_only=True)
        ni = obj.get("numerical_indices")
        ea = obj.get("edge_attr")
        return cls(obj["x"].to(device), obj["edge_index"].to(device), obj["node_ptr"], obj["edge_ptr"],
                   ea.to(device) if ea is not None else None, obj.get("mean"), obj.get("std"),
                   ni.tolist() if ni is not None else None)

    # ------------------------------------------------------------------ batches
    def _upload(self, host: np.ndarray) -> torch.Tensor:
        """Asynchronous copy of a small int64 index block through a ring of pinned staging buffers (allocating
        pinned memory per batch would cost more than the two gather launches)."""
        ring = getattr(self, "_ring", None)
        if ring is None:
            ring = self._ring = {"slot": 0, "bufs": [None] * 8, "events": [None] * 8}
        k = ring["slot"]
        ring["slot"] = (k + 1) % 8
        n = int(host.size)
        buf = ring["bufs"][k]
        if buf is None or buf.numel() < n:
            buf = ring["bufs"][k] = torch.empty(max(n, 1024), dtype=torch.int64).pin_memory()
        elif ring["events"][k] is not None:
            ring["events"][k].synchronize()          # the copy that last used this slot has finished
        buf[:n].copy_(torch.from_numpy(host))
        out = buf[:n].to(self.x.device, non_blocking=True)
        ev = ring["events"][k] = torch.cuda.Event()
        ev.record()
        return out

    def __len__(self) -> int:
        return len(self._node_ptr_host) - 1

    _next_token = 0

    def _token(self) -> int:
        """Process-wide serial number of this shard (never reused, unlike ``id()`` after a shard is collected)."""
        t = getattr(self, "_serial", None)
        if t is None:
            t = self._serial = GraphShard._next_token
            GraphShard._next_token += 1
        return t

    @property
    def uniform(self) -> bool:
        """True when every graph of the shard has the same nodes and edges (frames of one molecule): then all batches
        of one size share their ``edge_index`` / ``batch`` vector and carry a ``structure_id`` (see
        :class:`hgvae_amd.graphed.CapturedTraining`).  Decided once, with one read-back."""
        u = getattr(self, "_uniform", None)
        if u is None:
            sizes, ecnt = np.diff(self._node_ptr_host), np.diff(self._edge_ptr_host)
            u = len(sizes) > 0 and bool((sizes == sizes[0]).all()) and bool((ecnt == ecnt[0]).all())
            if u and ecnt[0] > 0:
                g, e = len(sizes), int(ecnt[0])
                local = self.edge_index.view(2, g, e) - self.node_ptr[:-1].view(1, g, 1)
                u = bool((local == local[:, :1, :]).all().item())
            self._uniform = u
        return u

    def batch(self, ids: Sequence[int], with_csr: bool = True) -> Batch:
        """The graphs ``ids`` (any order) as one Batch: ``x`` (padded layout), ``edge_index``, ``batch``,
        ``ptr``, ``edge_attr`` and, with ``with_csr``, ``graph`` - the batch's :class:`Graph` with both CSR
        directions filled in (pass it to the model in place of ``edge_index``)."""
        ids_np = np.asarray(list(ids), dtype=np.int64)
        b = int(ids_np.size)
        if b == 0:
            raise _lib.HgvError("empty batch")
        if ids_np.min() < 0 or ids_np.max() >= len(self):
            raise _lib.HgvError(f"graph ids must be in [0, {len(self)})")
        nptr = np.concatenate([[0], np.cumsum(self._node_ptr_host[ids_np + 1] - self._node_ptr_host[ids_np])])
        eptr = np.concatenate([[0], np.cumsum(self._edge_ptr_host[ids_np + 1] - self._edge_ptr_host[ids_np])])
        n_out, e_out = int(nptr[-1]), int(eptr[-1])
        max_n, max_e = int(np.diff(nptr).max()), int(np.diff(eptr).max())
        dev = self.x.device
        # one small upload: ids | node prefix | edge prefix
        idx = self._upload(np.concatenate([ids_np, nptr, eptr]))
        d_ids, d_nptr, d_eptr = idx[:b], idx[b:2 * b + 1], idx[2 * b + 1:]
        f = self.num_features
        x = new_padded(n_out, f, dev, zero_pad=False)
        bvec = torch.empty(n_out, dtype=torch.int64, device=dev)
        ei = torch.empty(2, e_out, dtype=torch.int64, device=dev)
        ea = torch.empty(e_out, self.edge_attr.size(1), dtype=torch.float32, device=dev) if self.edge_attr is not None else None
        g = self.graph
        i32 = dict(dtype=torch.int32, device=dev)
        if with_csr:
            rp_d, rp_s = torch.empty(n_out + 1, **i32), torch.empty(n_out + 1, **i32)
            col_d, perm_d = torch.empty(e_out, **i32), torch.empty(e_out, **i32)
            col_s, perm_s = torch.empty(e_out, **i32), torch.empty(e_out, **i32)
        else:
            rp_d = rp_s = col_d = perm_d = col_s = perm_s = None
        st = _lib.stream()
        _lib.call("hgv_collate_nodes", self.x.data_ptr(), self.x.stride(0), f, self.node_ptr.data_ptr(),
                  self.edge_ptr.data_ptr(), g.by_dst.rowptr.data_ptr(), g.by_src.rowptr.data_ptr(), d_ids.data_ptr(), b,
                  d_nptr.data_ptr(), d_eptr.data_ptr(), n_out, max_n, x.data_ptr(), x.stride(0), bvec.data_ptr(),
                  _lib.ptr(rp_d), _lib.ptr(rp_s), st)
        _lib.call("hgv_collate_edges", self.edge_index.data_ptr(), self.edge_index.size(1), g.by_dst.col.data_ptr(),
                  g.by_dst.perm.data_ptr(), g.by_src.col.data_ptr(), g.by_src.perm.data_ptr(), _lib.ptr(self.edge_attr),
                  self.edge_attr.size(1) if self.edge_attr is not None else 0, self.node_ptr.data_ptr(),
                  self.edge_ptr.data_ptr(), d_ids.data_ptr(), b, d_nptr.data_ptr(), d_eptr.data_ptr(), e_out, max_e,
                  ei.data_ptr(), _lib.ptr(col_d), _lib.ptr(perm_d), _lib.ptr(col_s), _lib.ptr(perm_s), _lib.ptr(ea), st)
        out = Batch(x=x, edge_index=ei, edge_attr=ea, batch=bvec)
        out.ptr = d_nptr
        out.num_graphs = b
        if self.uniform:
            out.structure_id = ("uniform", self._token(), b)
        if with_csr:
            out.graph = Graph.from_parts(ei, n_out, _Csr(rp_d, col_d, perm_d), _Csr(rp_s, col_s, perm_s))
        return out

    def loader(self, batch_size: int, shuffle: bool = False, seed: int = 0, drop_last: bool = False,
               with_csr: bool = True) -> "ShardLoader":
        """Iterable over mini-batches, usable where the reference passes its DataLoader (``len`` = batches per epoch)."""
        return ShardLoader(self, batch_size, shuffle, seed, drop_last, with_csr)


class ShardLoader:
    """Batches of a :class:`GraphShard`; with ``shuffle`` every epoch (every ``iter``) draws a new permutation."""

    def __init__(self, shard: GraphShard, batch_size: int, shuffle: bool, seed: int, drop_last: bool, with_csr: bool):
        if batch_size < 1:
            raise ValueError("batch_size must be positive")
        self.shard, self.batch_size, self.shuffle, self.drop_last, self.with_csr = shard, batch_size, shuffle, drop_last, with_csr
        self._rng = np.random.default_rng(seed)

    def __len__(self) -> int:
        n = len(self.shard)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self) -> Iterator[Batch]:
        order = np.arange(len(self.shard))
        if self.shuffle:
            self._rng.shuffle(order)
        for k in range(len(self)):
            yield self.shard.batch(order[k * self.batch_size:(k + 1) * self.batch_size], with_csr=self.with_csr)
