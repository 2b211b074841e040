"""No-GPU checks: the C-ABI library loads and exports every symbol the header declares, the
host-side containers behave like the PyG objects they mirror, and GPU-only operators refuse
CPU tensors instead of silently falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hgvae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hgv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hgvae_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/hgvae_hip.h but not exported"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)
    loaded = _lib.load()
    assert loaded.hgv_version() >= 1
    assert loaded.hgv_csr_build_ws(1000, 10) > 0 and loaded.hgv_loss_ws() > 0


def test_argument_errors_are_reported_without_a_gpu():
    from hgvae_amd import _lib
    lib = _lib.load()
    rc = lib.hgv_csr_build(None, 0, 0, -1, 5, 0, None, None, None, None, None, 0, None)
    assert rc == -1 and b"negative" in lib.hgv_last_error()
    with pytest.raises(_lib.HgvError):
        _lib.call("hgv_agg_fwd", None, None, 4, None, 8, 8, 0, None, None, None, None, 0, None, 8, None)


def test_operators_refuse_cpu_tensors():
    from hgvae_amd import _lib
    from hgvae_amd import functional as Fn
    from hgvae_amd.gvae_skipcon import VAE
    model = VAE(6, 8, 2)
    x = torch.randn(5, 6)
    ei = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 4]])
    with pytest.raises(_lib.HgvError):
        model(x, ei, None)
    with pytest.raises(_lib.HgvError):
        Fn.mse_loss(x, x)


def test_batch_matches_oracle_batch():
    from hgvae_amd import Batch, Data
    from oracle import pyg_ops as O
    gen = torch.Generator().manual_seed(0)
    items_a, items_b = [], []
    for n in (5, 1, 9):
        x = torch.randn(n, 4, generator=gen)
        ei = torch.randint(0, n, (2, 2 * n), generator=gen)
        ea = torch.rand(2 * n, 1, generator=gen)
        items_a.append(Data(x=x, edge_index=ei, edge_attr=ea))
        items_b.append(O.Data(x=x, edge_index=ei, edge_attr=ea))
    a, b = Batch.from_data_list(items_a), O.Batch.from_data_list(items_b)
    for name in ("x", "edge_index", "edge_attr", "batch", "ptr"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert a.num_graphs == 3 and a.num_nodes == 15 and a.num_node_features == 4


def test_dataset_normalises_selected_columns(tmp_path):
    from hgvae_amd.data import CustomGraphDataset, Data, collate_fn, load_graph, save_graph
    gen = torch.Generator().manual_seed(1)
    raw = []
    for i in range(3):
        x = torch.randn(10 + i, 5, generator=gen) * 3 + 7
        a = torch.arange(9 + i)
        save_graph(str(tmp_path / f"g{i}.pt"), Data(x=x, edge_index=torch.stack([a, a + 1]).t().contiguous().t()))
        raw.append(x)
    ds = CustomGraphDataset(str(tmp_path), [2, 3, 4])
    allf = torch.cat([r[:, 2:] for r in raw])
    assert torch.allclose(ds.mean, allf.mean(0)) and torch.allclose(ds.std, allf.std(0))
    d0 = ds[0]
    assert torch.allclose(d0.x[:, :2], raw[0][:, :2])
    assert torch.allclose(d0.x[:, 2:], (raw[0][:, 2:] - allf.mean(0)) / allf.std(0), atol=1e-6)
    assert torch.allclose(ds[0].x, d0.x)  # repeated access does not re-normalise the cached copy
    b = collate_fn([ds[0], ds[1]])
    assert b.x.size(0) == 21 and b.edge_index.max().item() == 20
    assert load_graph(str(tmp_path / "g0.pt")).edge_index.stride() in ((1, 2), (9, 1))


def test_read_pyg_pt_tensors_reads_raw_storages_without_unpickling(tmp_path):
    import zipfile
    import numpy as np
    from hgvae_amd.data import read_pyg_pt_tensors
    x = np.arange(24, dtype="<f4").reshape(4, 6)
    ei = np.array([[0, 1], [1, 2], [2, 3]], dtype="<i8")  # [E, 2] buffer, viewed as [2, E] strides (1, 2)
    ea = np.array([[10.0], [20.0], [30.0]], dtype="<f4")
    path = tmp_path / "g.pdb.pt"
    with zipfile.ZipFile(path, "w") as zf:
        zf.writestr("g.pdb/data.pkl", b"not a pickle we ever read")
        zf.writestr("g.pdb/data/0", x.tobytes())
        zf.writestr("g.pdb/data/1", ei.tobytes())
        zf.writestr("g.pdb/data/2", ea.tobytes())
    d = read_pyg_pt_tensors(str(path), 6)
    assert torch.equal(d.x, torch.from_numpy(x)) and d.edge_index.tolist() == [[0, 1, 2], [1, 2, 3]]
    assert d.edge_index.stride() == (1, 2)


def test_noise_tape_order_and_shape_checks():
    from hgvae_amd.noise import NoiseTape, PhiloxNoise
    t = NoiseTape([torch.ones(3, 2), torch.zeros(4, 3, 2)])
    m, _, _, _ = t.dropout((3, 2), torch.device("cpu"))
    assert m.dtype == torch.uint8 and m.sum() == 6
    with pytest.raises(RuntimeError):
        t.dropout((9, 9), torch.device("cpu"))
    p = PhiloxNoise(5)
    a, b = p.dropout((1,), None), p.normal((1,), None)
    assert a[0] is None and a[2] != b[2] and a[1] == b[1] == 5


def test_shard_range_balanced_and_contiguous():
    from hgvae_amd.distributed import shard_range
    for n, w in ((256, 8), (10, 4), (3, 8)):
        parts = [list(shard_range(n, r, w)) for r in range(w)]
        assert sum(parts, []) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_balanced_ranges_minimise_the_heaviest_rank():
    """Graphs of unequal size (454 ... 15 200 nodes in the bundled set) are split by edge count: contiguous,
    complete, every rank busy when there are enough graphs, and no contiguous split has a lighter heaviest rank."""
    import itertools
    import random
    from hgvae_amd.distributed import balanced_ranges

    def best_possible(costs, w):
        n, best = len(costs), None
        for cuts in itertools.combinations_with_replacement(range(n + 1), w - 1):
            b = [0, *cuts, n]
            m = max(sum(costs[b[i]:b[i + 1]]) for i in range(w))
            best = m if best is None else min(best, m)
        return best

    sizes = [453, 2626, 4696, 11212, 11454, 11939, 13238, 15199]          # edges of the eight bundled shapes
    assert balanced_ranges(sizes, 4) == [range(0, 4), range(4, 6), range(6, 7), range(7, 8)]
    assert balanced_ranges(sizes, 1) == [range(0, 8)]
    assert balanced_ranges([], 3) == [range(0, 0)] * 3
    rng = random.Random(3)
    for _ in range(200):
        n, w = rng.randint(0, 8), rng.randint(1, 5)
        costs = [rng.choice([0, 1, 7, 453, 15199, rng.randint(1, 99)]) for _ in range(n)]
        parts = balanced_ranges(costs, w)
        assert len(parts) == w and [i for r in parts for i in r] == list(range(n))
        assert max((sum(costs[i] for i in r) for r in parts), default=0) == best_possible(costs, w)
        if n >= w:
            assert all(len(r) for r in parts)
    with pytest.raises(ValueError):
        balanced_ranges([1, -1], 2)
