"""PDB files -> graphs, built on the device in the packed layout of a :class:`~hgvae_amd.shard.GraphShard`.

The reference turns every ``.pdb`` file into one graph on the host (``pdb_to_pyg_data``,
gvae/create_graphs_for_gvae.py:67-144): nodes = ATOM records with features one-hot(atom name) | one-hot(residue
name) | 3 force columns read from ``<name>_forces.txt`` (:11-20); one edge from every ATOM record to the next one
(:99-111) carrying the distance of the two atoms; one file written per graph, loaded again per item and per epoch
by the dataset class.  Here the text is parsed on the host (fixed PDB columns, as the reference reads them), the
records of ALL files go to the device as flat arrays, and one launch pair builds features, edges and edge attributes
of the whole dataset as one disjoint union, from which mini-batches are gathered on the device.

Two places where the reference's behaviour is not a function of its input and is pinned here:

* one-hot category order: the reference takes ``list(set(...))`` (:36-48), i.e. hash order, different per Python
  process; sorted order is used (``unique_types``).
* repeated atom serial numbers inside a file: the reference maps a serial to the LAST node that carries it and builds
  an undirected networkx graph (duplicates and self loops collapse); such files take a host path that reproduces this
  (``_edges_with_repeated_serials``).  Files with unique serials - every file written by OpenMM / PDBFixer, the tools
  the reference uses - produce the directed path 0 -> 1 -> ... -> n-1 and are built on the device.
"""
from __future__ import annotations

import os
import re
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .graph import pad4
from .shard import GraphShard

_FLOAT = re.compile(r"[-+]?\d*\.\d+|\d+")        # the pattern of load_energies_for_pdb (:13)


class AtomRecords:
    """The ATOM records of one file: serial [n] int64, atom_name / residue_name lists, coords [n, 3] float64."""
    __slots__ = ("serial", "atom_name", "residue_name", "residue_id", "coords", "path")

    def __init__(self, serial, atom_name, residue_name, residue_id, coords, path=None):
        self.serial = np.asarray(serial, dtype=np.int64)
        self.atom_name, self.residue_name = list(atom_name), list(residue_name)
        self.residue_id = np.asarray(residue_id, dtype=np.int64)
        self.coords = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        self.path = path
        if not (len(self.serial) == len(self.atom_name) == len(self.residue_name) == len(self.coords)):
            raise ValueError("ATOM record arrays of different lengths")

    def __len__(self):
        return len(self.serial)


def parse_pdb(path_or_lines) -> AtomRecords:
    """ATOM records of a PDB file (or of an iterable of its lines), read from the columns the reference reads
    (:80-85): serial 6:11, atom name 12:16, residue name 17:20, residue number 22:26, x / y / z 30:38 / 38:46 / 46:54."""
    path = path_or_lines if isinstance(path_or_lines, (str, os.PathLike)) else None
    lines = open(path) if path is not None else path_or_lines
    serial, names, residues, resid, coords = [], [], [], [], []
    try:
        for line in lines:
            if line.startswith("ATOM"):
                serial.append(int(line[6:11].strip()))
                names.append(line[12:16].strip())
                residues.append(line[17:20].strip())
                resid.append(int(line[22:26].strip()))
                coords.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    finally:
        if path is not None:
            lines.close()
    return AtomRecords(serial, names, residues, resid, coords, path)


def load_forces(path: str) -> np.ndarray:
    """[n, 3] forces of ``<name>_forces.txt``: the first three numbers of every line (load_energies_for_pdb, :11-20)."""
    rows = []
    with open(path) as f:
        for line in f:
            rows.append([float(v) for v in _FLOAT.findall(line)[:3]])
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


def unique_types(records: Sequence[AtomRecords]):
    """(atom names, residue names) over all files, sorted (get_unique_types, :35-48, with a defined order)."""
    atoms, residues = set(), set()
    for r in records:
        atoms.update(r.atom_name)
        residues.update(r.residue_name)
    return sorted(atoms), sorted(residues)


def _edges_with_repeated_serials(serial: np.ndarray):
    """Edge list the reference builds when serial numbers repeat: node_idx_map keeps the last node of a serial, the
    networkx graph is undirected (an edge is listed once, from the endpoint that comes first in node order, neighbours
    in insertion order; self loops included)."""
    last = {}
    for i, s in enumerate(serial.tolist()):
        last[s] = i
    adj = {i: {} for i in range(len(serial))}
    prev = None
    for s in serial.tolist():
        if prev is not None:
            u, v = last[prev], last[s]
            adj[u].setdefault(v, None)
            adj[v].setdefault(u, None)
        prev = s
    done, out = set(), []
    for u in range(len(serial)):                      # G.edges order: by node; an edge is reported by its first node
        for v in adj[u]:
            if v not in done:
                out.append((u, v))
        done.add(u)
    return np.array(out, dtype=np.int64).reshape(-1, 2).T


def build_shard(records: Sequence[AtomRecords], device, forces: Optional[Sequence[Optional[np.ndarray]]] = None,
                atom_types: Optional[List[str]] = None, residue_types: Optional[List[str]] = None,
                numerical_indices: Optional[Sequence[int]] = None) -> GraphShard:
    """All files as one device-resident :class:`GraphShard`.  ``forces[i]`` is the [n_i, 3] array of file i (None:
    zeros); ``numerical_indices`` standardises those feature columns with the dataset-wide mean and unbiased std, as
    ``CustomGraphDataset`` does (gvae/gvae_skipcon.py:240-259; the reference passes the three force columns)."""
    if not records:
        raise ValueError("no PDB records")
    dev = torch.device(device)
    if atom_types is None or residue_types is None:
        a_t, r_t = unique_types(records)
        atom_types, residue_types = atom_types or a_t, residue_types or r_t
    a_pos = {a: i for i, a in enumerate(atom_types)}
    r_pos = {r: i for i, r in enumerate(residue_types)}
    sizes = np.array([len(r) for r in records], dtype=np.int64)
    node_ptr = np.concatenate([[0], np.cumsum(sizes)])
    nt = int(node_ptr[-1])
    try:
        a_idx = np.fromiter((a_pos[a] for r in records for a in r.atom_name), dtype=np.int32, count=nt)
        r_idx = np.fromiter((r_pos[x] for r in records for x in r.residue_name), dtype=np.int32, count=nt)
    except KeyError as e:
        raise ValueError(f"atom / residue name {e} is not among the given categories") from None
    coords = np.concatenate([r.coords for r in records]) if nt else np.zeros((0, 3))
    fz = None
    if forces is not None:
        parts = []
        for r, f in zip(records, forces):
            f = np.zeros((len(r), 3)) if f is None else np.asarray(f, dtype=np.float64).reshape(-1, 3)
            # the reference asserts this per file (:122)
            assert len(f) == len(r), "Mismatch between number of atoms and energies."
            parts.append(f)
        fz = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(dev)
    irregular = [i for i, r in enumerate(records) if len(np.unique(r.serial)) != len(r)]
    ecnt = np.maximum(sizes - 1, 0)
    edge_ptr = np.concatenate([[0], np.cumsum(ecnt)])
    et = int(edge_ptr[-1])
    n_a, n_r = len(atom_types), len(residue_types)
    f_dim = n_a + n_r + 3
    ld = pad4(f_dim)
    x = torch.empty(nt, ld, dtype=torch.float32, device=dev)
    ei = torch.empty(2, et, dtype=torch.int64, device=dev)
    ea = torch.empty(et, dtype=torch.float32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    d_a, d_r = torch.from_numpy(a_idx).to(dev), torch.from_numpy(r_idx).to(dev)
    d_c = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    d_np, d_ep = torch.from_numpy(node_ptr).to(dev), torch.from_numpy(edge_ptr).to(dev)
    _lib.call("hgv_pdb_build", d_a.data_ptr(), d_r.data_ptr(), _lib.ptr(fz), d_c.data_ptr(), d_np.data_ptr(), d_ep.data_ptr(),
              len(records), nt, et, int(ecnt.max()) if len(ecnt) else 0, n_a, n_r, x.data_ptr(), ld, ei.data_ptr(),
              ea.data_ptr(), status.data_ptr(), _lib.stream())
    ea2 = ea.unsqueeze(1)
    if irregular:
        # files with repeated serial numbers: their edge lists come from the host restatement of the reference's
        # dictionary / undirected-graph logic; everything is re-packed once
        eis, eas = [], []
        ei_h, ea_h = ei.cpu().numpy(), ea.cpu().numpy()
        for i, r in enumerate(records):
            off = int(node_ptr[i])
            if i in irregular:
                e = _edges_with_repeated_serials(r.serial)
                d = np.array([np.linalg.norm(r.coords[u] - r.coords[v]) for u, v in e.T], dtype=np.float64)
                eis.append(e + off)
                eas.append(d.astype(np.float32))
            else:
                lo, hi = int(edge_ptr[i]), int(edge_ptr[i + 1])
                eis.append(ei_h[:, lo:hi])
                eas.append(ea_h[lo:hi])
        edge_ptr = np.concatenate([[0], np.cumsum([e.shape[1] for e in eis])])
        ei = torch.from_numpy(np.concatenate(eis, axis=1)).to(dev)
        ea2 = torch.from_numpy(np.concatenate(eas)).to(dev).unsqueeze(1)
    if int(status.item()) != 0:
        raise _lib.HgvError("pdb build: a category index was out of range")
    xv = x[:, :f_dim]
    mean = std = None
    if numerical_indices is not None:
        idx = torch.tensor(list(numerical_indices), dtype=torch.long, device=dev)
        feats = xv.index_select(1, idx)
        mean, std = feats.mean(dim=0), feats.std(dim=0)
        xv.index_copy_(1, idx, (feats - mean) / std)
        mean, std = mean.cpu(), std.cpu()
    shard = GraphShard(xv, ei, torch.from_numpy(node_ptr), torch.from_numpy(edge_ptr), ea2, mean, std, numerical_indices)
    shard.atom_types, shard.residue_types = list(atom_types), list(residue_types)
    return shard


def shard_from_pdb_files(paths: Sequence[str], device, with_forces: bool = True,
                         numerical_indices: Optional[Sequence[int]] = None) -> GraphShard:
    """The reference's conversion loop (:147-166) for a list of ``.pdb`` files, without the per-graph files: forces are
    read from ``<name>_forces.txt`` next to every file (``with_forces``), categories are collected over all files."""
    records = [parse_pdb(p) for p in paths]
    forces = [load_forces(p.replace(".pdb", "_forces.txt")) for p in paths] if with_forces else None
    return build_shard(records, device, forces, numerical_indices=numerical_indices)
