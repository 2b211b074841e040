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
