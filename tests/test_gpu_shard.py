"""Device-side mini-batch assembly (hgvae_amd.shard.GraphShard) against the host collate
(Batch.from_data_list, the restatement of gvae/gvae_skipcon.py:262-263) and the CSR build kernel."""
import numpy as np
import pytest
import torch

from graphs import path_batch, random_graph

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def make_graphs(f, with_attr, seed=0):
    """Ragged set: multigraphs with self loops and isolated nodes, paths, a single node without edges, an edgeless pair."""
    from hgvae_amd import Data
    gen = torch.Generator().manual_seed(seed)
    out = []
    specs = [("rand", 37, 200), ("path", 138, 0), ("rand", 5, 3), ("one", 1, 0), ("rand", 260, 3000), ("path", 2, 0),
             ("noedge", 2, 0), ("rand", 64, 64), ("path", 454, 0)]
    for i, (kind, n, e) in enumerate(specs):
        if kind == "rand":
            ei, n, _ = random_graph(n, e, seed=seed + i)
        elif kind == "path":
            ei, n, _ = path_batch([n])
        else:
            ei = torch.zeros(2, 0, dtype=torch.long)
        x = torch.randn(n, f, generator=gen) * 3 + 1
        ea = torch.randn(ei.size(1), 1, generator=gen) if with_attr else None
        out.append(Data(x=x, edge_index=ei, edge_attr=ea))
    return out


@pytest.mark.parametrize("f,with_attr", [(6, True), (105, False), (60, False), (1, True)])
def test_shard_batches_equal_host_collate_and_sorted_csr(f, with_attr):
    from hgvae_amd import Batch, Graph, GraphShard
    graphs = make_graphs(f, with_attr)
    shard = GraphShard.from_data_list(graphs, dev())
    assert len(shard) == len(graphs)
    for ids in ([0, 1, 2, 3, 4, 5, 6, 7, 8], [8, 3, 0], [4], [3], [6, 3, 6], [2, 2, 7, 1, 5, 0]):
        got = shard.batch(ids)
        want = Batch.from_data_list([graphs[i] for i in ids])
        assert torch.equal(got.x.cpu(), want.x)
        assert float(got.x.as_strided((got.x.size(0), got.x.stride(0)), (got.x.stride(0), 1))[:, f:].abs().sum()) == 0.0 \
            if got.x.size(0) else True                           # padding columns are zero
        assert torch.equal(got.edge_index.cpu(), want.edge_index)
        assert torch.equal(got.batch.cpu(), want.batch)
        assert torch.equal(got.ptr.cpu(), want.ptr)
        if with_attr:
            assert torch.equal(got.edge_attr.cpu(), want.edge_attr)
        ref = Graph(got.edge_index, got.x.size(0))               # radix-sorted CSR of the batch's edge list
        for a, b in ((got.graph.by_dst, ref.by_dst), (got.graph.by_src, ref.by_src)):
            assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col, b.col) and torch.equal(a.perm, b.perm)


def test_shard_normalisation_save_load_and_loader(tmp_path):
    """Dataset-wide standardisation of the selected columns (gvae/gvae_skipcon.py:240-259), round trip through a
    file, and a shuffled epoch that visits every graph once."""
    from hgvae_amd import GraphShard
    graphs = make_graphs(8, False, seed=3)
    idx = [5, 6, 7]
    shard = GraphShard.from_data_list(graphs, dev(), numerical_indices=idx)
    allx = torch.cat([g.x for g in graphs])
    mean, std = allx[:, idx].mean(0), allx[:, idx].std(0)
    want = allx.clone()
    want[:, idx] = (allx[:, idx] - mean) / std
    assert torch.equal(shard.batch(range(len(graphs))).x.cpu(), want)
    path = str(tmp_path / "shard.pt")
    shard.save(path)
    again = GraphShard.load(path, dev())
    assert torch.equal(again.batch([1, 4]).x, shard.batch([1, 4]).x)
    assert torch.equal(again.batch([1, 4]).edge_index, shard.batch([1, 4]).edge_index)
    loader = shard.loader(batch_size=4, shuffle=True, seed=1)
    assert len(loader) == 3
    sizes = sorted(int(b.x.size(0)) for b in loader)
    assert sum(sizes) == allx.size(0)
    first = [int(b.x.size(0)) for b in loader]
    second = [int(b.x.size(0)) for b in loader]
    assert first != second or len(set(first)) == 1                # a new permutation per epoch
    with pytest.raises(Exception):
        shard.batch([len(graphs)])


def test_train_vae_from_shard_matches_host_loader():
    """Same graphs, same seeds: an epoch driven by the shard loader (prebuilt CSR) gives the loss of an epoch driven
    by host-collated batches (CSR sorted per batch)."""
    import hgvae_amd
    from hgvae_amd import Batch, GraphShard
    from hgvae_amd.gvae_skipcon import VAE, train_vae
    graphs = [g for g in make_graphs(12, False, seed=5) if g.x.size(0) > 1]
    shard = GraphShard.from_data_list(graphs, dev())
    host = [Batch.from_data_list(graphs[i:i + 4]) for i in range(0, len(graphs), 4)]
    losses = []
    for loader in (host, shard.loader(batch_size=4)):
        torch.manual_seed(0)
        hgvae_amd.manual_seed(0)
        model = VAE(12, 32, 4).to(dev())
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        losses.append([train_vae(model, loader, opt) for _ in range(3)])
    assert losses[0] == losses[1]
