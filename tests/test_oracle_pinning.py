"""Pins the CPU oracle (no GPU needed): hand-computed known answers, an independent
dense-adjacency formulation, float64 gradient checks, structural invariants, and the committed
golden files.  The reference ships no tests or golden vectors and PyG is absent, so these are
what stands in for an executed PyG ("parity unpinned")."""
import os

import numpy as np
import pytest
import torch

from graphs import path_batch, random_graph
from oracle import models as OM
from oracle import pyg_ops as O
from tapes import GOLDEN, chig_batch, custom_pool_tape, hp3_batch, pool_tape, skipcon_tape, summarize_tensor


def test_sage_mean_known_answer_path_and_star():
    ei = torch.tensor([[0, 1], [1, 2]])
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    agg = O.scatter_mean(x[ei[0]], ei[1], 3)
    assert torch.equal(agg, torch.tensor([[0.0, 0.0], [1.0, 2.0], [3.0, 4.0]]))
    star = torch.tensor([[1, 2, 3, 3], [0, 0, 0, 0]])  # duplicate edge 3->0 counts twice
    xs = torch.tensor([[0.0], [1.0], [2.0], [6.0]])
    assert torch.equal(O.scatter_mean(xs[star[0]], star[1], 4), torch.tensor([[15.0 / 4], [0.0], [0.0], [0.0]]))


def test_sage_conv_matches_dense_formulation():
    ei, n, _ = random_graph(40, 300, seed=1)
    torch.manual_seed(0)
    conv = O.SAGEConv(7, 5).double()
    x = torch.randn(n, 7, dtype=torch.float64)
    dense = O.dense_sage_mean(x, ei) @ conv.lin_l.weight.t() + conv.lin_l.bias + x @ conv.lin_r.weight.t()
    assert torch.allclose(conv(x, ei), dense, atol=1e-12)


def test_graphconv_and_gcn_match_dense_formulation():
    ei, n, _ = random_graph(30, 200, seed=2, self_loops=True)
    torch.manual_seed(0)
    x = torch.randn(n, 6, dtype=torch.float64)
    gc = O.GraphConv(6, 3).double()
    a = O.dense_adjacency(ei, n)
    assert torch.allclose(gc(x, ei), (a @ x) @ gc.lin_rel.weight.t() + gc.lin_rel.bias + x @ gc.lin_root.weight.t(), atol=1e-12)
    gcn = O.GCNConv(6, 3).double()
    with torch.no_grad():
        gcn.bias.normal_()
    ah = O.dense_adjacency(O.add_remaining_self_loops(ei, n), n)
    dis = ah.sum(1).pow(-0.5)
    ref = (dis.view(-1, 1) * ah * dis.view(1, -1)) @ (x @ gcn.lin.weight.t()) + gcn.bias
    assert torch.allclose(gcn(x, ei), ref, atol=1e-12)


def test_gat_matches_dense_softmax_and_rows_sum_to_one():
    ei, n, _ = random_graph(25, 150, seed=3, self_loops=True)
    torch.manual_seed(0)
    gat = O.GATConv(6, 4, heads=1, dropout=0.0).double().eval()
    x = torch.randn(n, 6, dtype=torch.float64)
    dense = O.dense_gat(x, ei, gat.lin_src.weight, gat.att_src, gat.att_dst, gat.bias)
    assert torch.allclose(gat(x, ei), dense, atol=1e-10)
    # closed form: two sources into node 2 with logits l0, l1 (+ its self loop l2)
    e = torch.tensor([0.3, -1.2, 0.7], dtype=torch.float64)
    alpha = O.segment_softmax(e, torch.tensor([2, 2, 2]), 3)
    assert torch.allclose(alpha, torch.softmax(e, 0)) and abs(alpha.sum().item() - 1) < 1e-12


def test_topk_and_filter_adj_known_answers():
    score = torch.tensor([0.1, 0.9, 0.5, 0.5, -0.2, 0.3, 0.8])
    batch = torch.tensor([0, 0, 0, 0, 0, 1, 1])
    # graph 0: ceil(2.5) = 3 best: 1 (0.9), then the tie 2,3 in index order; graph 1: ceil(1) = 1 -> node 6
    assert O.topk(score, 0.5, batch).tolist() == [1, 2, 3, 6]
    ei = torch.tensor([[0, 1, 2, 3, 6, 5], [1, 2, 3, 1, 5, 6]])
    out, ea = O.filter_adj(ei, torch.arange(6.0).view(-1, 1), torch.tensor([1, 2, 3, 6]), 7)
    assert out.tolist() == [[0, 1, 2], [1, 2, 0]] and ea.view(-1).tolist() == [1.0, 2.0, 3.0]


def test_select_weight_known_answer():
    """PyG >= 2.4 scores with tanh(s * w / ||w||): a negative scalar keeps the LOWEST GNN outputs."""
    torch.manual_seed(0)
    pool = O.SAGPooling(1, ratio=0.5, GNN=O.GraphConv, select_weight=True)
    with torch.no_grad():
        pool.gnn.lin_rel.weight.zero_(); pool.gnn.lin_rel.bias.zero_(); pool.gnn.lin_root.weight.fill_(1.0)   # s = x
        pool.select.weight.fill_(-0.25)
    x = torch.tensor([[0.3], [-1.0], [2.0], [0.1]])
    ei = torch.zeros(2, 0, dtype=torch.long)
    out = pool(x, ei, None, torch.zeros(4, dtype=torch.long))
    assert out[4].tolist() == [1, 3]                                      # lowest two: -1.0, 0.1
    assert torch.allclose(out[5], torch.tanh(torch.tensor([1.0, -0.1])))  # score = tanh(-s)
    assert torch.allclose(out[0], x[out[4]] * out[5].view(-1, 1))
    with torch.no_grad():
        pool.select.weight.fill_(3.0)                                     # only the sign of w matters
    assert pool(x, ei, None, torch.zeros(4, dtype=torch.long))[4].tolist() == [2, 0]
    assert "select.weight" in pool.state_dict()


def test_kl_and_reparam_invariants():
    z = torch.zeros(9, 4)
    assert OM.kl_divergence(z, z).item() == 0.0
    m, ls = torch.randn(9, 4), torch.randn(9, 4) * 0.3
    assert OM.kl_divergence(m, ls).item() > 0
    eps = torch.randn(10, 9, 4)
    zz = OM.reparameterize(m, ls, 10, O.NoiseTape([eps]))
    assert torch.allclose(zz, m + ls.exp() * eps.mean(0), atol=1e-6)


def test_batch_from_data_list_offsets():
    a = O.Data(x=torch.ones(3, 2), edge_index=torch.tensor([[0, 1], [1, 2]]))
    b = O.Data(x=torch.zeros(2, 2), edge_index=torch.tensor([[0], [1]]))
    bt = O.Batch.from_data_list([a, b])
    assert bt.edge_index.tolist() == [[0, 1, 3], [1, 2, 4]] and bt.batch.tolist() == [0, 0, 0, 1, 1]


def test_csr_reference_is_stable():
    ei = torch.tensor([[5, 4, 3, 2, 1], [1, 0, 1, 0, 1]])
    rp, col, perm = O.csr_reference(ei, 6)
    assert rp.tolist() == [0, 2, 5, 5, 5, 5, 5] and col.tolist() == [4, 2, 5, 3, 1] and perm.tolist() == [1, 3, 0, 2, 4]


@pytest.mark.parametrize("conv", ["sage", "graphconv", "gcn", "gat"])
def test_conv_gradcheck_float64(conv):
    ei, n, _ = random_graph(12, 40, seed=4, self_loops=True)
    torch.manual_seed(1)
    mod = {"sage": O.SAGEConv, "graphconv": O.GraphConv, "gcn": O.GCNConv,
           "gat": lambda i, o: O.GATConv(i, o, heads=1, dropout=0.0)}[conv](3, 2).double().eval()
    x = torch.randn(n, 3, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: mod(t, ei), (x,), eps=1e-6, atol=1e-6)


def _run_golden(tag):
    g = np.load(os.path.join(GOLDEN, f"golden_{tag}.npz"))
    if tag in ("skipcon", "attention"):
        b = chig_batch(4)
        n = b.x.size(0)
        h, p = (138, 0.5) if tag == "skipcon" else (256, 0.49)
        torch.manual_seed(0)
        model = (OM.SkipconVAE if tag == "skipcon" else OM.SkipconAttentionVAE)(60, h, 28).train()
        tape = skipcon_tape(n, h, 28, 10, p, torch.Generator().manual_seed(77),
                            gat_edges=(b.edge_index.size(1) + n) if tag == "attention" else None)
        lt = OM.skipcon_loss(model, b.x, b.edge_index, b.batch, 1.0, O.NoiseTape(tape))
    else:
        b = hp3_batch()
        n = b.x.size(0)
        torch.manual_seed(0)
        model = (OM.PoolVAE if tag == "pool" else OM.CustomPoolVAE)(6, 138, 28).train()
        tape = (pool_tape if tag == "pool" else custom_pool_tape)(n, 138, 28, 10, torch.Generator().manual_seed(79))
        lt = (OM.pool_loss(model, b.x, b.edge_index, b.batch, 1.0, O.NoiseTape(tape)) if tag == "pool"
              else OM.custom_pool_loss(model, b.x, b.edge_index, b.batch, O.NoiseTape(tape)))
    return g, model, lt


@pytest.mark.parametrize("tag", ["skipcon", "attention", "pool", "custom_pool"])
def test_oracle_reproduces_golden(tag):
    g, model, lt = _run_golden(tag)
    lt[0].backward()
    assert abs(lt[0].item() - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert abs(lt[2].item() - float(g["kl"])) <= 1e-6 * abs(float(g["kl"])) + 1e-9
    for name, p in model.named_parameters():
        key = "grad/" + name
        if key not in g.files:
            assert p.grad is None
            continue
        got = p.grad.numpy() if p.numel() <= 4096 else summarize_tensor(p.grad)
        assert np.allclose(got, g[key], rtol=1e-5, atol=1e-7), name


def test_spectral_restatement_known_answers():
    """Normalised Laplacian of an undirected cycle: 1 on the diagonal, -1/2 next to it, eigenvalues 1 - cos(2 pi k / n);
    the reference's one-eigenvalue l_frequency is 0 / 0."""
    import math
    from oracle import spectral as OS
    n = 12
    a = torch.arange(n)
    ei = torch.cat([torch.stack([a, (a + 1) % n]), torch.stack([(a + 1) % n, a])], dim=1)
    lap = OS.normalized_laplacian(ei, n)
    want = torch.eye(n)
    want[a, (a + 1) % n] = -0.5
    want[(a + 1) % n, a] = -0.5
    assert torch.allclose(lap, want, atol=1e-7)
    assert abs(float(OS.compute_fiedler_value(ei, n)) - (1 - math.cos(2 * math.pi / n))) < 1e-6
    assert math.isnan(float(OS.l_frequency(ei, ei, n, n)))
    # a node without out-edges has degree 0 -> scaling 0, its row and column of A D^-1/2 vanish
    lap = OS.normalized_laplacian(torch.tensor([[0], [1]]), 2)
    assert torch.equal(lap, torch.eye(2) - torch.tensor([[0.0, 0.0], [0.0, 0.0]]))
