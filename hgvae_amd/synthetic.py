"""Seeded synthetic graph workloads (no dataset access is needed or attempted).

* :func:`protein_like_batch` - random-coil chains (3.8 A steps) with a symmetric radius graph
  whose cutoff is tuned per graph to a target mean in-degree, chain-ordered nodes, and the
  105-wide feature layout of the reference's private data (one-hot atom name (95) + one-hot
  residue (7) + 3 standardised force columns; gvae/create_graphs_for_gvae.py:126-131,
  numerical_indicies = [102,103,104] at gvae/gvae_skipcon.py:268).  The radius graph mimics
  autoencoder/convert_to_graph.py:9-29, the only non-path topology in the reference.
* :func:`local_random_graph` - one large graph with mostly-local neighbours for stressing
  the aggregation kernel.
* :func:`path_graph_batch` - directed paths, the topology of every bundled reference graph.
"""
from __future__ import annotations

import torch

from .data import Batch, Data


def _coil(n: int, gen: torch.Generator, device) -> torch.Tensor:
    v = torch.randn(n, 3, generator=gen, device=device)
    v = v / v.norm(dim=1, keepdim=True).clamp_min(1e-6) * 3.8
    return torch.cumsum(v, dim=0)


def protein_like_graph(n: int, mean_degree: int, f_in: int, gen: torch.Generator, device) -> Data:
    pos = _coil(n, gen, device)
    d = torch.cdist(pos, pos)
    d = torch.minimum(d, d.t())  # exactly symmetric, so every edge has its reverse
    d.fill_diagonal_(float("inf"))
    k = min(n * mean_degree, n * (n - 1))
    cutoff = torch.sort(d.flatten()).values[k - 1]   # k-th smallest distance (a device radix sort; kthvalue is ~20x slower here)
    row, col = (d <= cutoff).nonzero(as_tuple=True)
    edge_index = torch.stack([row, col])  # sorted by source; the target sort is real work
    n_atom, n_res = f_in - 10, 7
    x = torch.zeros(n, f_in, device=device)
    atom = torch.randint(0, n_atom, (n,), generator=gen, device=device)
    res = torch.randint(0, n_res, (n,), generator=gen, device=device)
    x[torch.arange(n, device=device), atom] = 1.0
    x[torch.arange(n, device=device), n_atom + res] = 1.0
    x[:, f_in - 3:] = torch.randn(n, 3, generator=gen, device=device)
    return Data(x=x, edge_index=edge_index)


def protein_like_batch(num_graphs: int, seed: int, device, nodes=(4500, 5500), mean_degree: int = 12,
                       f_in: int = 105) -> Batch:
    gen = torch.Generator(device=device).manual_seed(seed)
    sizes = torch.randint(nodes[0], nodes[1] + 1, (num_graphs,), generator=gen, device=device).tolist()
    return Batch.from_data_list([protein_like_graph(n, mean_degree, f_in, gen, device) for n in sizes])


def protein_like_batch_by_index(graph_ids, base_seed: int, device, nodes=(4500, 5500), mean_degree: int = 12,
                                f_in: int = 105) -> Batch:
    """The graphs ``graph_ids`` of an unbounded seeded data set: graph g (size included) is a function of
    ``base_seed + g`` alone, so any partition of a range of ids over processes yields the same graphs as one process
    drawing the whole range."""
    items = []
    for g in graph_ids:
        gen = torch.Generator(device=device).manual_seed(int(base_seed) + int(g))
        n = int(torch.randint(nodes[0], nodes[1] + 1, (1,), generator=gen, device=device).item())
        items.append(protein_like_graph(n, mean_degree, f_in, gen, device))
    return Batch.from_data_list(items)


def local_random_graph(n: int, degree: int, f: int, seed: int, device, window: int = 512, far: float = 0.2) -> Batch:
    gen = torch.Generator(device=device).manual_seed(seed)
    e = n * degree
    dst = torch.randint(0, n, (e,), generator=gen, device=device)
    near = (dst + torch.randint(-window, window + 1, (e,), generator=gen, device=device)).clamp(0, n - 1)
    farv = torch.randint(0, n, (e,), generator=gen, device=device)
    src = torch.where(torch.rand(e, generator=gen, device=device) < far, farv, near)
    x = torch.randn(n, f, generator=gen, device=device)
    return Batch.from_data_list([Data(x=x, edge_index=torch.stack([src, dst]))])


def path_graph_batch(sizes, f_in: int, seed: int, device) -> Batch:
    gen = torch.Generator(device=device).manual_seed(seed)
    items = []
    for n in sizes:
        a = torch.arange(n - 1, device=device)
        items.append(Data(x=torch.randn(n, f_in, generator=gen, device=device), edge_index=torch.stack([a, a + 1])))
    return Batch.from_data_list(items)
