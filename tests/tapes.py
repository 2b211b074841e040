"""Noise tapes (dropout masks / normal draws in the order one train step consumes them) and the
real-input batches shared by the tests and tests/golden/make_fixtures.py."""
import os

import numpy as np
import torch

from oracle import pyg_ops as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def skipcon_tape(n, h, z, ns, p, gen, gat_edges=None):
    """gvae_skipcon order: encode (3 masks; the attention variant replaces the 2nd by the GAT
    attention mask [E', 1]), eps, 3 decoder masks, then the second encode."""
    def m(*shape):
        return torch.bernoulli(torch.full(shape, 1 - p), generator=gen)

    def enc():
        return [m(n, h), m(gat_edges, 1) if gat_edges is not None else m(n, h), m(n, h)]
    return enc() + [torch.randn(ns, n, z, generator=gen)] + [m(n, h), m(n, h), m(n, h)] + enc()


def pool_tape(n, h, z, ns, gen):
    """vae_with_pool order: encode (2 masks), eps, decoder mask, second encode."""
    def m(*shape):
        return torch.bernoulli(torch.full(shape, 0.5), generator=gen)
    return [m(n, h), m(n, 2 * h), torch.randn(ns, n, z, generator=gen), m(n, h), m(n, h), m(n, 2 * h)]


def custom_pool_tape(n, h, z, ns, gen):
    """vae_custom_pool order: encode, eps + decoder mask (forward), eps + decoder mask (recon_loss)."""
    def m(*shape):
        return torch.bernoulli(torch.full(shape, 0.5), generator=gen)
    return [m(n, h), m(n, 2 * h), torch.randn(ns, n, z, generator=gen), m(n, h),
            torch.randn(ns, n, z, generator=gen), m(n, h)]


def chig_batch(copies=4):
    """`copies` chignolin graphs (138 atoms, 137 path edges, 50 + 7 one-hots + 3 synthetic force
    columns N(0,1)*50 with seed 1234+i, standardised over the whole set), batched."""
    d = np.load(os.path.join(GOLDEN, "chig_atoms.npz"))
    a_idx = torch.from_numpy(d["atom_index"].astype(np.int64))
    r_idx = torch.from_numpy(d["residue_index"].astype(np.int64))
    n_a, n_r, n = int(d["n_atom_types"]), int(d["n_residue_types"]), len(a_idx)
    xs = []
    for i in range(copies):
        g = torch.Generator().manual_seed(1234 + i)
        x = torch.zeros(n, n_a + n_r + 3)
        x[torch.arange(n), a_idx] = 1
        x[torch.arange(n), n_a + r_idx] = 1
        x[:, -3:] = torch.randn(n, 3, generator=g) * 50.0
        xs.append(x)
    allf = torch.cat([x[:, -3:] for x in xs])
    mean, std = allf.mean(0), allf.std(0)
    items = []
    for x in xs:
        x[:, -3:] = (x[:, -3:] - mean) / std
        a = torch.arange(n - 1)
        items.append(O.Data(x=x, edge_index=torch.stack([a, a + 1])))
    return O.Batch.from_data_list(items)


def hp3_batch():
    """The real 1HP3_1 graph (454 x 6, all columns standardised) plus its first 200 nodes as a second graph."""
    d = np.load(os.path.join(GOLDEN, "1hp3_1.npz"))
    x, ei = torch.from_numpy(d["x"]), torch.from_numpy(d["edge_index"])
    x = (x - x.mean(0)) / x.std(0)
    return O.Batch.from_data_list([O.Data(x=x, edge_index=ei),
                                   O.Data(x=x[:200].clone(), edge_index=ei[:, :199].clone())])


def summarize_tensor(t):
    """Compact, order-sensitive summary of a tensor for the golden files."""
    f = t.detach().double().flatten()
    return np.concatenate([[f.sum().item(), f.abs().sum().item(), f.norm().item()], f[:61].numpy()])
