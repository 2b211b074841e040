"""Degenerate inputs through the operator API: graphs without edges, single-row matrices, batches of
edgeless graphs.  Expected values come from the oracle restatement of the PyG operators."""
import pytest
import torch

from oracle import pyg_ops as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("n,f_in,f_out", [(5, 6, 138), (1, 138, 56), (300, 138, 138)])
def test_sage_conv_without_edges(n, f_in, f_out):
    """No edges: every aggregate is 0, out = lin_r(x) + bias; gradients flow through the root path only."""
    from hgvae_amd.nn import SAGEConv
    torch.manual_seed(n)
    ref = O.SAGEConv(f_in, f_out)
    conv = SAGEConv(f_in, f_out)
    conv.load_state_dict(ref.state_dict())
    conv = conv.to(dev())
    ei = torch.zeros(2, 0, dtype=torch.long)
    x = torch.randn(n, f_in)
    xr = x.clone().requires_grad_(True)
    xg = x.to(dev()).requires_grad_(True)
    want = ref(xr, ei)
    got = conv(xg, ei.to(dev()))
    assert torch.allclose(got.cpu(), want, rtol=1e-5, atol=1e-5)
    g = torch.randn(n, f_out)
    want.backward(g)
    got.backward(g.to(dev()))
    assert torch.allclose(xg.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(conv.lin_r.weight.grad.cpu(), ref.lin_r.weight.grad, rtol=1e-4, atol=1e-4)
    assert float(conv.lin_l.weight.grad.abs().max()) == 0.0


def test_gat_conv_without_edges_uses_self_loops_only():
    from hgvae_amd.nn import GATConv
    torch.manual_seed(0)
    ref = O.GATConv(12, 16)
    conv = GATConv(12, 16)
    conv.load_state_dict(ref.state_dict())
    conv = conv.to(dev()).eval()
    ref.eval()
    x = torch.randn(7, 12)
    ei = torch.zeros(2, 0, dtype=torch.long)
    assert torch.allclose(conv(x.to(dev()), ei.to(dev())).cpu(), ref(x, ei), rtol=1e-5, atol=1e-5)


def test_shard_batch_of_edgeless_graphs():
    from hgvae_amd import Batch, Data, GraphShard
    graphs = [Data(x=torch.randn(n, 6), edge_index=torch.zeros(2, 0, dtype=torch.long)) for n in (3, 1, 4)]
    shard = GraphShard.from_data_list(graphs, dev())
    got = shard.batch([2, 0])
    want = Batch.from_data_list([graphs[2], graphs[0]])
    assert torch.equal(got.x.cpu(), want.x) and got.edge_index.shape == (2, 0)
    assert torch.equal(got.batch.cpu(), want.batch)
    assert torch.equal(got.graph.by_dst.rowptr.cpu(), torch.zeros(8, dtype=torch.int32))
    assert torch.equal(got.graph.by_src.rowptr.cpu(), torch.zeros(8, dtype=torch.int32))


def test_batch_norm_single_row_raises_like_torch():
    from hgvae_amd import functional as Fn
    bn = torch.nn.BatchNorm1d(8).to(dev()).train()
    with pytest.raises(ValueError):
        Fn.bn_act(torch.randn(1, 8, device=dev()), bn, True, 0.5, True)


@pytest.mark.parametrize("conv", ["graphconv", "gcn"])
def test_graphconv_and_gcn_without_edges(conv):
    """GraphConv: out = lin_root(x) + bias; GCNConv: only the self loop remains, out = x W^T + b."""
    from hgvae_amd import nn as N
    torch.manual_seed(1)
    ref = O.GraphConv(6, 1) if conv == "graphconv" else O.GCNConv(6, 1)
    mine = N.GraphConv(6, 1) if conv == "graphconv" else N.GCNConv(6, 1)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev())
    x = torch.randn(9, 6)
    ei = torch.zeros(2, 0, dtype=torch.long)
    assert torch.allclose(mine(x.to(dev()), ei.to(dev())).cpu(), ref(x, ei), rtol=1e-5, atol=1e-6)


def test_sag_pooling_single_node_and_edgeless_graphs():
    """k = ceil(0.5 n): a one-node graph keeps its node, an edgeless graph pools by score alone."""
    from hgvae_amd import nn as N
    torch.manual_seed(2)
    ref = O.SAGPooling(8, ratio=0.5)
    mine = N.SAGPooling(8, ratio=0.5)
    mine.load_state_dict(ref.state_dict())
    mine = mine.to(dev())
    x = torch.randn(1 + 5 + 4, 8)
    batch = torch.tensor([0] + [1] * 5 + [2] * 4)
    ei = torch.tensor([[6, 7, 8], [7, 8, 9]])                   # graph 0: one node; graph 1: no edges; graph 2: a path
    want = ref(x, ei, None, batch)
    got = mine(x.to(dev()), ei.to(dev()), None, batch.to(dev()))
    assert torch.equal(got[4].cpu(), want[4])                    # perm
    assert torch.allclose(got[0].cpu(), want[0], rtol=1e-5, atol=1e-6)
    assert torch.equal(got[1].cpu(), want[1]) and torch.equal(got[3].cpu(), want[3])


@pytest.mark.parametrize("F", [6, 128, 138, 276])
@pytest.mark.parametrize("weighted", [False, True])
def test_aggregation_with_hub_rows(F, weighted):
    """Rows far longer than the average (two hubs of ~2 000 and ~700 entries among rows of ~8) take the workgroup-per-row
    pass; everything else the row-per-lane-group kernel.  Against float64 index_add_, forward, mean and transposed."""
    import torch
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(F)
    n, e = 3000, 24000
    src = torch.randint(0, n, (e,), generator=gen)
    dst = torch.randint(0, n, (e,), generator=gen)
    dst[:2000] = 17                                   # hub in the by-target CSR
    src[2000:2700] = 2999                             # hub in the by-source CSR (last row)
    ei = torch.stack([src, dst]).to(dev)
    g = Graph(ei, n)
    assert g.by_dst.long_rows(n).threshold == 64
    x = torch.randn(n, F, generator=gen).to(dev).requires_grad_(True)
    if weighted:
        w = torch.rand(e, generator=gen).to(dev)
        out = Fn._WeightedAggregate.apply(x, g, w[g.by_dst.perm.long()])
        want = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, ei[1], x.detach().double()[ei[0]] * w.double()[:, None])
    else:
        out = Fn.aggregate(x, g, mean=True)
        deg = torch.bincount(ei[1], minlength=n).clamp(min=1).double()[:, None]
        want = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, ei[1], x.detach().double()[ei[0]]) / deg
    torch.cuda.synchronize()
    assert g.by_dst.long_rows(n).maybe_any() and int(g.by_dst.long_rows(n).count) == 1
    scale = float(want.abs().max())
    assert float((out.detach().double() - want).abs().max()) <= 2e-6 * scale
    gout = torch.randn(n, F, generator=gen).to(dev)
    out.backward(gout)
    if weighted:
        want_g = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, ei[0], gout.double()[ei[1]] * w.double()[:, None])
    else:
        want_g = torch.zeros(n, F, dtype=torch.float64, device=dev).index_add_(0, ei[0], (gout.double() / deg)[ei[1]])
    assert float((x.grad.double() - want_g).abs().max()) <= 2e-6 * float(want_g.abs().max())
    # a second run gives the same bits (the partial sums of a hub row are combined in a fixed order)
    again = Fn._WeightedAggregate.apply(x.detach(), g, w[g.by_dst.perm.long()]) if weighted else Fn.aggregate(x.detach(), g, mean=True)
    assert torch.equal(again, out.detach())


def test_graph_without_hubs_learns_that_the_long_pass_is_not_needed():
    import torch
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    dev = torch.device("cuda:0")
    n = 5000
    a = torch.arange(n - 1, device=dev)
    g = Graph(torch.stack([a, a + 1]), n)
    x = torch.randn(n, 16, device=dev)
    first = Fn.aggregate(x, g, mean=True)
    torch.cuda.synchronize()
    assert not g.by_dst.long_rows(n).maybe_any()
    assert torch.equal(Fn.aggregate(x, g, mean=True), first)


def test_input_that_is_a_column_slice_of_a_wider_matrix():
    """x = wide[:, :F] is 16-byte aligned with a leading dimension far beyond F: what lies behind column F is foreign data,
    here 1e30.  It must neither reach the result nor set the row scale of the fp16 projections: the forward pass equals the
    one on a contiguous copy bit for bit."""
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(12)
    n, f = 300, 10
    wide = torch.randn(n, 32, device=dev, generator=gen)
    wide[:, f:] = 1e30
    x = wide[:, :f]
    a = torch.arange(n - 1, device=dev)
    ei = torch.cat([torch.stack([a, a + 1]), torch.stack([a + 1, a])], 1)
    torch.manual_seed(12)
    model = SkipconVAE(f, 32, 4).to(dev).eval()
    with torch.no_grad():
        got = model.encode(x, ei)
        want = model.encode(x.contiguous(), ei)
    for g, w in zip(got, want):
        assert torch.isfinite(g).all()
        assert torch.equal(g, w)
