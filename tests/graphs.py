"""Seeded graph generators shared by the tests (CPU tensors)."""
import torch


def path_batch(sizes, seed=0):
    """Disjoint directed path graphs (e -> e+1), the topology of every bundled reference graph."""
    eis, off = [], 0
    for n in sizes:
        a = torch.arange(n - 1, dtype=torch.long) + off
        eis.append(torch.stack([a, a + 1]))
        off += n
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    return torch.cat(eis, dim=1), off, batch


def random_graph(n, e, seed=0, self_loops=True, hub=True):
    """Random multigraph with duplicate edges, optional self loops, isolated nodes and one hub target."""
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, max(1, n - n // 8), (e,), generator=g)  # the last n/8 nodes get no in-edges
    if hub and e >= 8:
        dst[: e // 4] = 3 % n  # a hub: one row with e/4 entries
    if not self_loops:
        clash = src == dst
        src[clash] = (src[clash] + 1) % n
    ei = torch.stack([src, dst])
    return ei, n, torch.zeros(n, dtype=torch.long)


def local_graph(n, deg, seed=0, window=512, far=0.2):
    """Mean in-degree `deg`; most neighbours within +-window of the row (protein-like locality)."""
    g = torch.Generator().manual_seed(seed)
    e = n * deg
    dst = torch.randint(0, n, (e,), generator=g)
    near = dst + torch.randint(-window, window + 1, (e,), generator=g)
    near = near.clamp(0, n - 1)
    farv = torch.randint(0, n, (e,), generator=g)
    pick = torch.rand(e, generator=g) < far
    src = torch.where(pick, farv, near)
    return torch.stack([src, dst]), n, torch.zeros(n, dtype=torch.long)
