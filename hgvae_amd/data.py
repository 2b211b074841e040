"""Graph containers, batching and the dataset class (the reference's L1 data layer).

``Data`` / ``Batch.from_data_list`` mirror the PyG objects used at
gvae/gvae_skipcon.py:27,262-263; ``CustomGraphDataset`` / ``collate_fn`` mirror
gvae/gvae_skipcon.py:236-263 (per-column normalisation with the dataset-wide mean and
unbiased std of the selected feature columns).

Files are plain tensor dictionaries ``{"x", "edge_index", "edge_attr"}`` written by
:func:`save_graph` and read with ``torch.load(weights_only=True)``; pickled PyG ``Data``
objects are not loaded (that would need PyG and arbitrary unpickling) - convert them once
with :func:`read_pyg_pt_tensors`, which reads the raw tensor storages from the zip archive
without unpickling anything.
"""
from __future__ import annotations

import os
import zipfile
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset


class Data:
    def __init__(self, x=None, edge_index=None, edge_attr=None, batch=None, **kw):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def num_nodes(self) -> int:
        return int(self.x.size(0))

    @property
    def num_node_features(self) -> int:
        return int(self.x.size(1))

    @property
    def num_edges(self) -> int:
        return int(self.edge_index.size(1))

    def to(self, device):
        for name in ("x", "edge_index", "edge_attr", "batch", "ptr"):
            t = getattr(self, name, None)
            if isinstance(t, torch.Tensor):
                setattr(self, name, t.to(device))
        return self

    def __repr__(self):
        parts = [f"{k}={list(v.shape)}" for k, v in vars(self).items() if isinstance(v, torch.Tensor)]
        return f"{type(self).__name__}({', '.join(parts)})"


class Batch(Data):
    """Disjoint union of graphs: features concatenated, edge indices offset, ``batch`` = graph id per node."""

    @staticmethod
    def from_data_list(data_list: Sequence[Data]) -> "Batch":
        sizes = [int(d.x.size(0)) for d in data_list]
        dev = data_list[0].x.device
        ptr = torch.zeros(len(sizes) + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(torch.tensor(sizes, dtype=torch.long), 0)
        x = torch.cat([d.x for d in data_list], dim=0)
        ecounts = torch.tensor([int(d.edge_index.size(1)) for d in data_list], dtype=torch.long)
        edge_index = torch.cat([d.edge_index for d in data_list], dim=1)
        edge_index = edge_index + torch.repeat_interleave(ptr[:-1], ecounts).to(dev).unsqueeze(0)
        batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(dev)
        eas = [d.edge_attr for d in data_list]
        edge_attr = torch.cat(eas, dim=0) if all(e is not None for e in eas) else None
        out = Batch(x=x, edge_index=edge_index, edge_attr=edge_attr, batch=batch)
        out.ptr = ptr.to(dev)
        out.num_graphs = len(sizes)
        return out


def collate_fn(batch: List[Data]) -> Batch:
    return Batch.from_data_list(batch)


def save_graph(path: str, data: Data) -> None:
    obj = {"x": data.x.cpu(), "edge_index": data.edge_index.cpu()}
    if data.edge_attr is not None:
        obj["edge_attr"] = data.edge_attr.cpu()
    torch.save(obj, path)


def load_graph(path: str) -> Data:
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(obj, dict) or "x" not in obj or "edge_index" not in obj:
        raise ValueError(f"{path}: expected a dict with 'x' and 'edge_index' tensors (see hgvae_amd.data.save_graph)")
    return Data(x=obj["x"], edge_index=obj["edge_index"], edge_attr=obj.get("edge_attr"))


def read_pyg_pt_tensors(path: str, num_features: int) -> Data:
    """Read x / edge_index / edge_attr of a ``torch.save``d PyG ``Data`` WITHOUT unpickling.

    Such an archive stores raw storages ``<name>/data/0`` (x, fp32), ``data/1`` (edge_index,
    int64, an [E, 2] buffer viewed as [2, E] with strides (1, 2)) and ``data/2`` (edge_attr,
    fp32 [E, 1]) - the layout written by ``torch.tensor(list(G.edges)).long().t()`` at
    autoencoder/graph_with_energy.py:93.  The number of feature columns is not recoverable
    from the raw bytes and must be given.
    """
    with zipfile.ZipFile(path) as zf:
        names = {os.path.basename(os.path.dirname(n)) + "/" + os.path.basename(n): n for n in zf.namelist()}
        raw = {k: zf.read(names[f"data/{k}"]) for k in ("0", "1", "2") if f"data/{k}" in names}
    x = np.frombuffer(raw["0"], dtype="<f4").reshape(-1, num_features).copy()
    ei = np.frombuffer(raw["1"], dtype="<i8").reshape(-1, 2).copy()
    ea = np.frombuffer(raw["2"], dtype="<f4").reshape(-1, 1).copy() if "2" in raw else None
    if ei.size and (ei.min() < 0 or ei.max() >= x.shape[0]):
        raise ValueError(f"{path}: raw storages do not have the expected PyG Data layout")
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei).t(),
                edge_attr=torch.from_numpy(ea) if ea is not None else None)


class CustomGraphDataset(Dataset):
    """Folder of graph files; the selected feature columns are standardised with dataset-wide statistics."""

    def __init__(self, data_folder: str, numerical_indices: Sequence[int], cache: bool = True):
        self.data_file_list = sorted(os.path.join(data_folder, f) for f in os.listdir(data_folder) if f.endswith(".pt"))
        self.numerical_indices = list(numerical_indices)
        self._cache: Optional[dict] = {} if cache else None
        self.mean, self.std = self._compute_mean_std()

    def __len__(self):
        return len(self.data_file_list)

    def _load(self, index: int) -> Data:
        if self._cache is not None and index in self._cache:
            d = self._cache[index]
        else:
            d = load_graph(self.data_file_list[index])
            if self._cache is not None:
                self._cache[index] = d
        return Data(x=d.x.clone(), edge_index=d.edge_index, edge_attr=d.edge_attr)

    def __getitem__(self, index: int) -> Data:
        d = self._load(index)
        idx = self.numerical_indices
        d.x[:, idx] = (d.x[:, idx] - self.mean) / self.std
        return d

    def _compute_mean_std(self):
        feats = torch.cat([self._load(i).x[:, self.numerical_indices] for i in range(len(self))], dim=0)
        return torch.mean(feats, dim=0), torch.std(feats, dim=0)
