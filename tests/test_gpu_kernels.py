"""Kernel-level parity: HIP path (through the C ABI) vs the CPU oracle, same seeded inputs.

Integer/index results are compared bit-exactly; the unweighted aggregation forward is also
bit-exact (same summation order as scatter_add_); other fp32 results use the norm-wise
tolerance  max|a-b| <= rtol * max|b| + atol  with rtol 1e-5 forward / 1e-4 gradients.
"""
import pytest
import torch

from graphs import local_graph, path_batch, random_graph
from oracle import pyg_ops as O

pytestmark = pytest.mark.gpu

FWD_RTOL, BWD_RTOL, ATOL = 1e-5, 1e-4, 1e-6


def close(a, b, rtol, atol=ATOL):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= rtol * ref + atol, f"max err {err:.3e} vs ref scale {ref:.3e} (rtol {rtol})"


def dev():
    return torch.device("cuda:0")


# ------------------------------------------------------------------ CSR build
CSR_CASES = {
    "path_batch": lambda: path_batch([5, 1, 138, 454, 2]),
    "random_small": lambda: random_graph(37, 200, seed=1),
    "random_two_pass": lambda: random_graph(5000, 60000, seed=2),
    "random_three_pass": lambda: random_graph(70000, 300000, seed=3),
    "local_100k": lambda: local_graph(100000, 20, seed=4),
    "single_node_loops": lambda: (torch.zeros(2, 7, dtype=torch.long), 1, torch.zeros(1, dtype=torch.long)),
    "tile_boundary": lambda: random_graph(300, 4096 * 3, seed=5),
    "tile_boundary_plus1": lambda: random_graph(300, 4096 * 3 + 1, seed=6),
}


@pytest.mark.parametrize("name", list(CSR_CASES))
@pytest.mark.parametrize("by_src", [False, True])
def test_csr_build_bit_exact(name, by_src):
    from hgvae_amd.graph import Graph
    ei, n, _ = CSR_CASES[name]()
    g = Graph(ei.to(dev()), n)
    csr = g.by_src if by_src else g.by_dst
    rp, col, perm = O.csr_reference(ei, n, by_src)
    g.check()
    assert torch.equal(csr.rowptr.cpu(), rp)
    assert torch.equal(csr.perm.cpu(), perm)
    assert torch.equal(csr.col.cpu(), col)


def test_csr_build_strided_edge_index_and_empty():
    from hgvae_amd.graph import Graph
    ei_t = torch.randint(0, 50, (333, 2), generator=torch.Generator().manual_seed(7))
    ei = ei_t.to(dev()).t()  # [2, E] with strides (1, 2): the layout of the bundled graphs
    assert ei.stride() == (1, 2)
    g = Graph(ei, 50)
    rp, col, perm = O.csr_reference(ei_t.t().contiguous(), 50)
    assert torch.equal(g.by_dst.rowptr.cpu(), rp) and torch.equal(g.by_dst.col.cpu(), col)
    g0 = Graph(torch.zeros(2, 0, dtype=torch.long, device=dev()), 9)
    assert torch.equal(g0.by_dst.rowptr.cpu(), torch.zeros(10, dtype=torch.int32))


def test_csr_build_flags_out_of_range():
    from hgvae_amd import _lib
    from hgvae_amd.graph import Graph
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]], device=dev())
    g = Graph(ei, 3)
    with pytest.raises(_lib.HgvError):
        g.check()


def test_sort_pairs_matches_stable_sort():
    from hgvae_amd import _lib
    gen = torch.Generator().manual_seed(11)
    for n, lo, hi in [(1, 0, 32), (1000, 0, 32), (50000, 0, 32), (9000, 8, 24), (4097, 0, 8)]:
        k32 = torch.randint(0, 2**32, (n,), generator=gen, dtype=torch.int64)
        kbits = torch.where(k32 >= 2**31, k32 - 2**32, k32).to(torch.int32).to(dev())  # uint32 bit pattern
        ko = torch.empty_like(kbits)
        vo = torch.empty(n, dtype=torch.int32, device=dev())
        lib = _lib.load()
        ws = torch.empty(lib.hgv_sort_pairs_ws(n), dtype=torch.uint8, device=dev())
        _lib.call("hgv_sort_pairs_u32", kbits.data_ptr(), None, n, lo, hi, ko.data_ptr(), vo.data_ptr(),
                  ws.data_ptr(), ws.numel(), _lib.stream())
        field = (k32 >> lo) & ((1 << (hi - lo)) - 1)
        ref = torch.sort(field, stable=True).indices.to(torch.int32)
        assert torch.equal(vo.cpu(), ref), (n, lo, hi)


# ------------------------------------------------------------------ aggregation
@pytest.mark.parametrize("F", [1, 6, 28, 56, 105, 128, 138, 256, 276])
@pytest.mark.parametrize("mean", [True, False])
def test_aggregate_forward_bit_exact(F, mean):
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    ei, n, _ = random_graph(700, 9000, seed=F)
    x = torch.randn(n, F, generator=torch.Generator().manual_seed(F))
    ref = (O.scatter_mean if mean else O.scatter_sum)(x.index_select(0, ei[0]), ei[1], n)
    g = Graph(ei.to(dev()), n)
    out = Fn.aggregate(x.to(dev()), g, mean=mean)
    assert out.shape == ref.shape
    # rows up to the long-row threshold are summed in edge order: the bits of a sequential scatter_add_.  The hub (row 3,
    # 2 250 entries) is summed in sixteen in-order blocks whose partial sums are then added: same value to rounding.
    hub = 3
    assert int(g.in_degree[hub]) > g.by_dst.long_rows(n).threshold
    rest = torch.arange(n) != hub
    assert torch.equal(out.cpu()[rest], ref[rest]), (out.cpu() - ref).abs().max()
    exact = (O.scatter_mean if mean else O.scatter_sum)(x.double().index_select(0, ei[0]), ei[1], n)[hub]
    err_blocked, err_sequential = (out.cpu()[hub].double() - exact).abs().max(), (ref[hub].double() - exact).abs().max()
    assert float(err_blocked) <= max(2.0 * float(err_sequential), 1e-6 * float(exact.abs().max()))
    assert torch.equal(Fn.aggregate(x.to(dev()), g, mean=mean), out)


def test_aggregate_known_answer_path():
    """3-node path 0->1->2: mean aggregation is [0, x0, x1] (node 0 has no in-edge)."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    ei = torch.tensor([[0, 1], [1, 2]])
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
    out = Fn.aggregate(x.to(dev()), Graph(ei.to(dev()), 3), mean=True).cpu()
    assert torch.equal(out, torch.tensor([[0.0, 0.0], [1.0, 2.0], [3.0, 4.0]]))


@pytest.mark.parametrize("F", [6, 56, 138])
@pytest.mark.parametrize("mean", [True, False])
def test_aggregate_backward(F, mean):
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    ei, n, _ = random_graph(500, 6000, seed=100 + F)
    gen = torch.Generator().manual_seed(F)
    x = torch.randn(n, F, generator=gen, requires_grad=True)
    gout = torch.randn(n, F, generator=gen)
    ref = (O.scatter_mean if mean else O.scatter_sum)(x.index_select(0, ei[0]), ei[1], n)
    ref.backward(gout)
    xd = x.detach().to(dev()).requires_grad_(True)
    out = Fn.aggregate(xd, Graph(ei.to(dev()), n), mean=mean)
    out.backward(gout.to(dev()))
    close(xd.grad, x.grad, BWD_RTOL)


def test_aggregate_large_local_graph_vs_oracle():
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    ei, n, _ = local_graph(20000, 20, seed=9)
    x = torch.randn(n, 128, generator=torch.Generator().manual_seed(1))
    ref = O.scatter_mean(x.index_select(0, ei[0]), ei[1], n)
    out = Fn.aggregate(x.to(dev()), Graph(ei.to(dev()), n), mean=True)
    assert torch.equal(out.cpu(), ref)


# ------------------------------------------------------------------ relu/dropout/batch-norm
@pytest.mark.parametrize("H", [6, 138, 256, 276])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("p", [0.0, 0.5, 0.49])
def test_bn_act_forward_backward(H, relu, p):
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    gen = torch.Generator().manual_seed(H + int(p * 100))
    n = 1237
    y = (torch.randn(n, H, generator=gen) * 1.7 + 0.3).requires_grad_(True)
    mask = torch.bernoulli(torch.full((n, H), 1 - p), generator=gen)
    gout = torch.randn(n, H, generator=gen)
    bn = torch.nn.BatchNorm1d(H)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(H, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(H, generator=gen))
    bn_d = torch.nn.BatchNorm1d(H)
    bn_d.load_state_dict(bn.state_dict())
    bn_d = bn_d.to(dev())
    # oracle: torch modules with the explicit mask
    t = torch.relu(y) if relu else y
    t = O.dropout_with(t, p, True, O.NoiseTape([mask]))
    ref = bn(t)
    ref.backward(gout)
    yd = y.detach().to(dev()).requires_grad_(True)
    with use_noise(NoiseTape([mask])):
        out = Fn.bn_act(yd, bn_d, relu, p, training=True)
    out.backward(gout.to(dev()))
    close(out, ref, FWD_RTOL)
    close(yd.grad, y.grad, BWD_RTOL)
    close(bn_d.weight.grad, bn.weight.grad, BWD_RTOL)
    close(bn_d.bias.grad, bn.bias.grad, BWD_RTOL)
    close(bn_d.running_mean, bn.running_mean, FWD_RTOL)
    close(bn_d.running_var, bn.running_var, FWD_RTOL)
    assert int(bn_d.num_batches_tracked) == 1
    # eval mode uses the running statistics
    bn.eval()
    ref_e = bn(torch.relu(y) if relu else y)
    out_e = Fn.bn_act(yd.detach(), bn_d, relu, p, training=False)
    close(out_e, ref_e, FWD_RTOL)


@pytest.mark.parametrize("n,H,relu,p", [(1237, 138, True, 0.5), (4999, 256, True, 0.49), (70, 6, False, 0.3), (20011, 276, True, 0.5)])
@pytest.mark.parametrize("consumers", ["both_twice", "second_once", "only_first"])
def test_bn_act_pair_equals_two_calls_bit_for_bit(n, H, relu, p, consumers):
    """Two dropout draws of one pre-activation in one pass: outputs, saved statistics, running buffers, the summed input
    gradient and the affine gradients are those of two ordinary calls on the same matrix (same Philox offsets)."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import new_padded
    from hgvae_amd.noise import PhiloxNoise, use_noise
    gen = torch.Generator().manual_seed(n + H)
    y0 = new_padded(n, H, dev())
    y0.copy_((torch.randn(n, H, generator=gen) * 1.3 + 0.2).to(dev()))
    gs = [torch.randn(n, H, generator=gen).to(dev()) for _ in range(4)]

    def run(paired):
        bn = torch.nn.BatchNorm1d(H).to(dev())
        with torch.no_grad():
            bn.weight.copy_(torch.rand(H, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(H, generator=torch.Generator().manual_seed(2)))
        y = y0.detach().requires_grad_(True)
        with use_noise(PhiloxNoise(77)):
            if paired:
                (a, a2), (b, b2) = Fn.bn_act_pair(y, bn, relu, p)
            else:
                ya, yb = Fn.fork(y)
                a, a2 = Fn.bn_act(ya, bn, relu, p, True, fork=True)
                b, b2 = Fn.bn_act(yb, bn, relu, p, True, fork=True)
        if consumers == "both_twice":
            loss = (a * gs[0]).sum() + (a2 * gs[1]).sum() + (b * gs[2]).sum() + (b2 * gs[3]).sum()
        elif consumers == "second_once":
            loss = (a * gs[0]).sum() + (a2 * gs[1]).sum() + (b * gs[2]).sum()
        else:
            loss = (a2 * gs[1]).sum()
        loss.backward()
        return (a.detach(), b.detach(), y.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone(),
                int(bn.num_batches_tracked))

    got, want = run(True), run(False)
    assert not torch.equal(got[0], got[1])                      # two different draws
    for g, w in zip(got[:-1], want[:-1]):
        assert torch.equal(g, w)
    assert got[-1] == want[-1] == 2


def test_bn_act_large_mean_is_stable():
    """Columns with mean >> std: shifted sums must not lose the variance."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(3)
    y = torch.randn(5000, 8, generator=gen) * 0.01 + 100.0
    bn = torch.nn.BatchNorm1d(8)
    ref = bn(y)
    out = Fn.bn_act(y.to(dev()), torch.nn.BatchNorm1d(8).to(dev()), False, 0.0, training=True)
    close(out, ref, 1e-3)


def test_bn_act_philox_dropout_statistics_and_determinism():
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    y = torch.ones(4096, 64, device=dev())
    bn = torch.nn.BatchNorm1d(64, affine=False).to(dev())
    outs = []
    for _ in range(2):
        with use_noise(PhiloxNoise(seed=123)):
            outs.append(Fn.bn_act(y, bn, False, 0.5, training=True))
    assert torch.equal(outs[0], outs[1])
    # after batch norm a Bernoulli(0.5)*2 column is +-1: check keep fraction ~ 0.5
    keep = (outs[0] > 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01


# ------------------------------------------------------------------ reparam / KL / MSE
def test_reparam_kl_mse_forward_backward():
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    from oracle.models import kl_divergence, reparameterize
    gen = torch.Generator().manual_seed(5)
    n, z, f, ns = 777, 28, 60, 10
    ml = (torch.randn(n, 2 * z, generator=gen) * 0.5).requires_grad_(True)
    eps = torch.randn(ns, n, z, generator=gen)
    x = torch.randn(n, f, generator=gen)
    w = torch.randn(z, f, generator=gen)
    mean, log_std = ml.chunk(2, dim=-1)
    zz = reparameterize(mean, log_std, ns, O.NoiseTape([eps]))
    rec = zz @ w
    loss_ref = torch.nn.functional.mse_loss(rec, x) + 0.7 * kl_divergence(mean, log_std)
    loss_ref.backward()

    mld = ml.detach().to(dev()).requires_grad_(True)
    with use_noise(NoiseTape([eps])):
        zd = Fn.reparameterize(mld, z, ns)
    recd = zd @ w.to(dev())
    loss = Fn.mse_loss(recd, x.to(dev())) + 0.7 * Fn.kl_divergence(mld, z)
    loss.backward()
    close(zd, zz, FWD_RTOL)
    close(loss, loss_ref, FWD_RTOL)
    close(mld.grad, ml.grad, BWD_RTOL)


def test_kl_known_answers():
    from hgvae_amd import functional as Fn
    ml = torch.zeros(10, 8, device=dev())
    assert Fn.kl_divergence(ml, 4).item() == 0.0  # mu = 0, log_std = 0
    ml[:, :4] = 1.0
    assert abs(Fn.kl_divergence(ml, 4).item() - 2.0) < 1e-6  # 0.5 * sum mu^2 = 2


def test_reparam_philox_moments():
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    ml = torch.zeros(20000, 56, device=dev())
    with use_noise(PhiloxNoise(seed=9)):
        z = Fn.reparameterize(ml, 28, 10)
    # mean of 10 standard normals: mean 0, std 1/sqrt(10)
    assert abs(z.mean().item()) < 5e-3
    assert abs(z.std().item() - 10 ** -0.5) < 5e-3


# ------------------------------------------------------------------ convolutions (module level)
def _copy_params(dst, src):
    dst.load_state_dict(src.state_dict())
    return dst.to(dev())


@pytest.mark.parametrize("graph", ["paths", "random_loops"])
@pytest.mark.parametrize("conv", ["sage", "graphconv", "gcn"])
def test_conv_forward_backward(conv, graph):
    from hgvae_amd import nn as HN
    gen = torch.Generator().manual_seed(21)
    ei, n, _ = path_batch([30, 41]) if graph == "paths" else random_graph(300, 2500, seed=8, self_loops=True)
    fin, fout = 23, 17
    torch.manual_seed(3)
    ref = {"sage": O.SAGEConv, "graphconv": O.GraphConv, "gcn": O.GCNConv}[conv](fin, fout)
    mod = _copy_params({"sage": HN.SAGEConv, "graphconv": HN.GraphConv, "gcn": HN.GCNConv}[conv](fin, fout), ref)
    x = torch.randn(n, fin, generator=gen, requires_grad=True)
    gout = torch.randn(n, fout, generator=gen)
    out_ref = ref(x, ei)
    out_ref.backward(gout)
    xd = x.detach().to(dev()).requires_grad_(True)
    out = mod(xd, ei.to(dev()))
    out.backward(gout.to(dev()))
    close(out, out_ref, FWD_RTOL)
    close(xd.grad, x.grad, BWD_RTOL)
    refp = dict(ref.named_parameters())
    for name, p in mod.named_parameters():
        close(p.grad, refp[name].grad, BWD_RTOL)


@pytest.mark.parametrize("graph", ["paths", "random_loops", "hub"])
@pytest.mark.parametrize("p", [0.0, 0.49])
def test_gat_forward_backward(graph, p):
    from hgvae_amd import nn as HN
    from hgvae_amd.noise import NoiseTape, use_noise
    gen = torch.Generator().manual_seed(31)
    if graph == "paths":
        ei, n, _ = path_batch([30, 41])
    elif graph == "random_loops":
        ei, n, _ = random_graph(300, 2500, seed=9, self_loops=True, hub=False)
    else:
        ei, n, _ = random_graph(200, 4000, seed=10, self_loops=True, hub=True)
    c = 64
    torch.manual_seed(4)
    ref = O.GATConv(c, c, heads=1, dropout=p).train()
    mod = _copy_params(HN.GATConv(c, c, heads=1, dropout=p), ref).train()
    n_edges = int((ei[0] != ei[1]).sum()) + n
    mask = torch.bernoulli(torch.full((n_edges, 1), 1 - p), generator=gen)
    x = torch.randn(n, c, generator=gen, requires_grad=True)
    gout = torch.randn(n, c, generator=gen)
    out_ref = ref(x, ei, O.NoiseTape([mask]))
    out_ref.backward(gout)
    xd = x.detach().to(dev()).requires_grad_(True)
    with use_noise(NoiseTape([mask])):
        out = mod(xd, ei.to(dev()))
    out.backward(gout.to(dev()))
    close(out, out_ref, FWD_RTOL)
    close(xd.grad, x.grad, BWD_RTOL)
    refp = dict(ref.named_parameters())
    for name, prm in mod.named_parameters():
        close(prm.grad, refp[name].grad, BWD_RTOL)


@pytest.mark.parametrize("c,p", [(64, 0.0), (138, 0.49), (256, 0.3)])
def test_gat_hub_rows_take_the_workgroup_kernels(c, p):
    """A node that 3 000 others point to and that points to 2 500 others (far above 4 x the mean degree): its row of the
    by-target CSR and its row of the by-source CSR go through k_gat_*_long.  Forward and every gradient against the
    oracle, and against the lane-group kernels on the same inputs."""
    from hgvae_amd import functional as Fn
    from hgvae_amd import nn as HN
    from hgvae_amd.noise import NoiseTape, use_noise
    gen = torch.Generator().manual_seed(77 + c)
    n = 4000
    base = torch.randint(0, n, (2, 12000), generator=gen)
    into_hub = torch.stack([torch.randperm(n, generator=gen)[:3000], torch.full((3000,), 17)])
    from_hub = torch.stack([torch.full((2500,), 17), torch.randperm(n, generator=gen)[:2500]])
    second = torch.stack([torch.randperm(n, generator=gen)[:700], torch.full((700,), 3999)])
    ei = torch.cat([base, into_hub, from_hub, second], 1)
    torch.manual_seed(4)
    ref = O.GATConv(c, c, heads=1, dropout=p).train()
    mod = _copy_params(HN.GATConv(c, c, heads=1, dropout=p), ref).train()
    n_edges = int((ei[0] != ei[1]).sum()) + n
    mask = torch.bernoulli(torch.full((n_edges, 1), 1 - p), generator=gen)
    x = torch.randn(n, c, generator=gen, requires_grad=True)
    gout = torch.randn(n, c, generator=gen)
    out_ref = ref(x, ei, O.NoiseTape([mask]))
    out_ref.backward(gout)
    results = []
    names = []
    real_call = Fn._lib.call

    def spy_call(name, *a, _real=real_call):
        names.append(name)
        return _real(name, *a)
    for long_rows in (True, False):
        Fn._gat_long = long_rows
        Fn._lib.call = spy_call
        try:
            mod.zero_grad()
            xd = x.detach().to(dev()).requires_grad_(True)
            with use_noise(NoiseTape([mask])):
                out = mod(xd, ei.to(dev()))
            out.backward(gout.to(dev()))
        finally:
            Fn._gat_long = True
            Fn._lib.call = real_call
        results.append((out.detach().clone(), xd.grad.clone(), {k: v.grad.clone() for k, v in mod.named_parameters()}))
        if long_rows:
            assert "hgv_gat_fwd_long" in names and "hgv_gat_bwd_long" in names
            close(out, out_ref, FWD_RTOL)
            close(xd.grad, x.grad, BWD_RTOL)
            refp = dict(ref.named_parameters())
            for name, prm in mod.named_parameters():
                close(prm.grad, refp[name].grad, BWD_RTOL)
    (oa, ga, pa), (ob, gb, pb) = results
    assert float((oa - ob).abs().max()) <= 2e-6 * float(ob.abs().max())
    assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max())
    for k in pb:
        assert float((pa[k] - pb[k]).abs().max()) <= 1e-5 * max(float(pb[k].abs().max()), 1e-30), k
    # the hub rows are on the lists the kernels were given
    from hgvae_amd.graph import graph_of
    eid = ei.to(dev())                                 # the graph keeps only a weak reference to its edge list
    lg = graph_of(eid, n).with_self_loops()
    for csr in (lg.by_dst, lg.by_src):
        lr = csr.long_rows(n)
        rows = set(lr.rows[: int(lr.count)].tolist())
        assert 17 in rows, (lr.threshold, rows)


def test_gat_attention_rows_sum_to_one_known_answer():
    """2-edge graph 0->2, 1->2 plus loops: with a_src = a_dst = 0 every alpha is 1/row length."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import Graph
    g = Graph(torch.tensor([[0, 1], [2, 2]], device=dev()), 3).with_self_loops()
    xw = torch.tensor([[1.0, 0.0], [0.0, 1.0], [3.0, 3.0]], device=dev())
    att = torch.zeros(2, 2, device=dev())
    out = Fn.gat_aggregate(xw, att, g, 0.2, 0.0, False).cpu()
    assert torch.allclose(out, torch.tensor([[1.0, 0.0], [0.0, 1.0], [4.0 / 3, 4.0 / 3]]), atol=1e-6)


@pytest.mark.skipif(False, reason='')
def test_gat_accepts_single_lin_weight_checkpoint():
    from hgvae_amd import nn as HN
    m = HN.GATConv(8, 8)
    sd = m.state_dict()
    new_w = sd.pop("lin_src.weight").clone() + 1
    sd.pop("lin_dst.weight")
    sd["lin.weight"] = new_w
    m.load_state_dict(sd)
    assert torch.equal(m.lin_src.weight.data, new_w)


# ------------------------------------------------------------------ pooling
def test_topk_perm_matches_oracle_with_ties():
    from hgvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(41)
    sizes = [454, 1, 2, 2627, 7, 300]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    score = torch.tanh(torch.randn(sum(sizes), generator=gen))
    score[10:40] = score[10]        # ties inside a graph
    score[500:520] = 0.0
    score[520:530] = -0.0           # signed zeros compare equal
    bd = batch.to(dev())
    for ratio in (0.5, 0.3, 1.0):
        ref = O.topk(score, ratio, batch)
        for global_sort in (False, True):          # one workgroup per graph in LDS / radix passes over all scores
            perm = Fn.topk_perm(score.to(dev()), ratio, bd, global_sort=global_sort)
            assert torch.equal(perm.cpu(), ref), (ratio, global_sort)


def test_topk_perm_graph_sizes_around_the_lds_table():
    """Power-of-two boundaries of the in-LDS sort, its 16384-node capacity and a graph beyond it (global path)."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(42)
    for sizes in ([1, 2, 3, 1023, 1024, 1025], [16384, 5], [15200, 13239], [16385, 40]):
        batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
        score = torch.tanh(torch.randn(sum(sizes), generator=gen))
        score[::7] = score[0]                   # many ties
        ref = O.topk(score, 0.5, batch)
        perm = Fn.topk_perm(score.to(dev()), 0.5, batch.to(dev()))
        assert torch.equal(perm.cpu(), ref), sizes


def test_topk_perm_with_an_undersized_max_nodes_stays_inside_the_graph():
    """The C entry point trusts max_nodes for the size of its LDS table; a caller that understates it must get valid
    node ids of the right graph (in no particular order) and the other graphs' results untouched."""
    from hgvae_amd import _lib
    gen = torch.Generator().manual_seed(44)
    sizes = [300, 900, 120]
    n, b = sum(sizes), len(sizes)
    ptr = torch.tensor([0, 300, 1200, 1320])
    k = torch.tensor([150, 450, 60])
    kptr = torch.tensor([0, 150, 600, 660])
    batch = torch.repeat_interleave(torch.arange(b), torch.tensor(sizes))
    score = torch.randn(n, generator=gen)
    ref = O.topk(score, 0.5, batch)
    d = dev()
    perm = torch.full((660,), -7, dtype=torch.int64, device=d)
    args = [t.to(d) for t in (score, batch, ptr, k, kptr)]
    _lib.call("hgv_topk_perm", *[t.data_ptr() for t in args], n, b, 512, perm.data_ptr(), None, 0, _lib.stream())
    got = perm.cpu()
    assert torch.equal(got[:150], ref[:150]) and torch.equal(got[600:], ref[600:])
    assert int(got[150:600].min()) >= 300 and int(got[150:600].max()) < 1200


def test_filter_adj_and_unpool_bit_exact():
    from hgvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(43)
    ei, n, _ = random_graph(500, 6000, seed=12)
    perm = torch.randperm(n, generator=gen)[:230]
    ea = torch.randn(ei.size(1), 1, generator=gen)
    ref_ei, ref_ea = O.filter_adj(ei, ea, perm, n)
    out_ei, out_ea = Fn.filter_adj(ei.to(dev()), ea.to(dev()), perm.to(dev()), n)
    assert torch.equal(out_ei.cpu(), ref_ei) and torch.equal(out_ea.cpu(), ref_ea)
    # strided edge_index (the layout of the bundled graphs) and an empty selection
    ei_s = ei.t().contiguous().to(dev()).t()
    out2, _ = Fn.filter_adj(ei_s, None, perm.to(dev()), n)
    assert torch.equal(out2.cpu(), ref_ei)
    out3, _ = Fn.filter_adj(ei.to(dev()), None, perm[:0].to(dev()), n)
    assert out3.shape == (2, 0)
    # capacity-bounded mode (no host read-back): kept edges first, zeros behind, count on the device
    with Fn.bounded_pooling():
        b_ei, b_ea = Fn.filter_adj(ei.to(dev()), ea.to(dev()), perm.to(dev()), n)
    m = int(b_ei.hgv_count.item())
    assert m == ref_ei.size(1) and b_ei.shape == (2, ei.size(1))
    assert torch.equal(b_ei[:, :m].cpu(), ref_ei) and int(b_ei[:, m:].abs().max()) == 0
    assert torch.equal(b_ea[:m].cpu(), ref_ea)
    # unpool gathers
    x = torch.randn(n, 138, generator=gen)
    assert torch.equal(Fn.gather_rows(x.to(dev()), perm.to(dev())).cpu(), x[perm])
    assert torch.equal(Fn.gather_i64(perm.to(dev()), out_ei).cpu(), perm[ref_ei])


@pytest.mark.parametrize("scorer", ["graphconv", "gcn"])
def test_sag_pooling_matches_oracle(scorer):
    from hgvae_amd import nn as HN
    gen = torch.Generator().manual_seed(47)
    ei, n, batch = path_batch([454, 300, 77])
    h = 138
    torch.manual_seed(5)
    ref = O.SAGPooling(h, GNN=O.GraphConv if scorer == "graphconv" else O.GCNConv)
    mod = _copy_params(HN.SAGPooling(h, GNN=HN.GraphConv if scorer == "graphconv" else HN.GCNConv), ref)
    x = torch.randn(n, h, generator=gen, requires_grad=True)
    ea = torch.rand(ei.size(1), 1, generator=gen)
    rx, rei, rea, rb, rperm, rs = ref(x, ei, ea, batch)
    xd = x.detach().to(dev()).requires_grad_(True)
    ox, oei, oea, ob, operm, os_ = mod(xd, ei.to(dev()), ea.to(dev()), batch.to(dev()))
    # scores agree to fp32 tolerance; the selection is exact given the device scores
    score_dev = torch.tanh(mod.gnn(xd, ei.to(dev())).reshape(-1)).detach().cpu()
    close(score_dev, torch.tanh(ref.gnn(x, ei).view(-1)), FWD_RTOL)
    assert torch.equal(operm.cpu(), O.topk(score_dev, 0.5, batch))
    if torch.equal(operm.cpu(), rperm):  # no near-tie flipped the order: everything must match
        close(ox, rx, FWD_RTOL)
        assert torch.equal(oei.cpu(), rei) and torch.equal(ob.cpu(), rb) and torch.equal(oea.cpu(), rea)
        close(os_, rs, FWD_RTOL)
        gout = torch.randn(rx.shape, generator=gen)
        rx.backward(gout)
        ox.backward(gout.to(dev()))
        close(xd.grad, x.grad, BWD_RTOL)
        refp = dict(ref.named_parameters())
        for name, prm in mod.named_parameters():
            close(prm.grad, refp[name].grad, BWD_RTOL)
    else:
        pytest.fail("top-k order differs from the oracle on a seeded input without ties")


def test_sag_pooling_with_select_weight_matches_oracle_and_loads_either_checkpoint():
    """PyG >= 2.4 checkpoints carry ``select.weight``; the score is tanh(gnn(x) * w / ||w||)."""
    from hgvae_amd import nn as HN
    gen = torch.Generator().manual_seed(48)
    ei, n, batch = path_batch([300, 154, 9])
    h = 138
    torch.manual_seed(6)
    ref = O.SAGPooling(h, GNN=O.GraphConv, select_weight=True)
    with torch.no_grad():
        ref.select.weight.fill_(-0.7)
    mod = HN.SAGPooling(h, GNN=HN.GraphConv)              # built the PyG <= 2.3 way ...
    assert "select.weight" not in mod.state_dict()
    mod.load_state_dict(ref.state_dict())                 # ... and switched by the checkpoint's key
    assert float(mod.select.weight.detach()) == pytest.approx(-0.7)
    mod = mod.to(dev())
    x = torch.randn(n, h, generator=gen)
    rx, rei, _, rb, rperm, rs = ref(x, ei, None, batch)
    ox, oei, _, ob, operm, os_ = mod(x.to(dev()), ei.to(dev()), None, batch.to(dev()))
    assert torch.equal(operm.cpu(), rperm)
    close(ox, rx, FWD_RTOL)
    close(os_, rs, FWD_RTOL)
    assert torch.equal(oei.cpu(), rei) and torch.equal(ob.cpu(), rb)
    # the older layout still loads into a model built with the selector only if the key is present
    old = O.SAGPooling(h, GNN=O.GraphConv)
    HN.SAGPooling(h, GNN=HN.GraphConv).load_state_dict(old.state_dict())
    with pytest.raises(RuntimeError):
        HN.SAGPooling(h, GNN=HN.GraphConv, select_weight=True).load_state_dict(old.state_dict())


# ------------------------------------------------------------------ fp32 MFMA projections
GEMM_SHAPES = [  # (M, K1, K2, N)
    (1, 6, None, 1), (127, 28, None, 138), (129, 138, 138, 138), (1000, 105, 105, 138), (777, 138, 138, 56),
    (513, 276, 276, 56), (300, 138, 138, 276), (2049, 60, 60, 256), (640, 34, None, 138), (5000, 138, None, 105),
    # hidden width 256 (attention variant): the split weight image is streamed through LDS chunk by chunk
    (2500, 256, 256, 256), (191, 256, None, 256), (193, 105, 105, 256), (12 * 16 * 3 + 5, 288, 288, 138), (40, 256, 256, 200),
]


@pytest.fixture(params=["f16x3", "bf16x6", "fp32"])
def projection_kind(request):
    from hgvae_amd import functional as Fn
    Fn.set_projection(request.param)
    yield request.param
    Fn.set_projection(Fn.default_projection())


@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_dual_linear_forward_backward(shape, projection_kind):
    """y = xa Wa^T + xb Wb^T + b on the MFMA kernels vs an fp64 CPU product (MFMA fp32 is an exact
    fma chain, so the error is the usual fp32 accumulation error ~1e-7 * sum|a b|)."""
    from hgvae_amd import functional as Fn
    m, k1, k2, n = shape
    gen = torch.Generator().manual_seed(m + n)
    xa = torch.randn(m, k1, generator=gen, requires_grad=True)
    wfull = torch.randn(n, k1 + (k2 or 0), generator=gen, requires_grad=True)   # one weight, two column slices
    bias = torch.randn(n, generator=gen, requires_grad=True)
    xb = torch.randn(m, k2, generator=gen, requires_grad=True) if k2 else None
    gout = torch.randn(m, n, generator=gen)
    xd = [t.detach().double().requires_grad_(True) if t is not None else None for t in (xa, wfull, bias, xb)]
    ref = xd[0] @ xd[1][:, :k1].t() + xd[2]
    if k2:
        ref = ref + xd[3] @ xd[1][:, k1:].t()
    ref.backward(gout.double())
    dv = [t.detach().to(dev()).requires_grad_(True) if t is not None else None for t in (xa, wfull, bias, xb)]
    out = Fn.dual_linear(dv[0], dv[1][:, :k1], dv[3], dv[1][:, k1:] if k2 else None, dv[2])
    out.backward(gout.to(dev()))
    close(out, ref, 2e-6)
    for got, want in zip(dv, xd):
        if got is not None:
            close(got.grad, want.grad, 5e-6)
    # pad columns of the output buffer are zero
    if n % 4:
        base = out.as_strided((m, (n + 3) // 4 * 4), (out.stride(0), 1))
        assert float(base.detach()[:, n:].abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["f16x3", "bf16x6", "fp32"])
@pytest.mark.parametrize("shape", [(160_984, 138, 138, 138), (160_984, 105, 105, 138), (131_072 + 7, 138, None, 56),
                                   (70_001, 276, None, 138), (100_003, 256, 256, 256), (40_001, 138, None, 138),
                                   (32_768, 140, 129, 129), (32_767, 138, 138, 138), (50_000, 138, None, 112),
                                   (40_000, 28, None, 138), (40_003, 138, None, 28), (36_000, 64, 40, 100), (36_005, 144, 144, 144),
                                   (33_000, 8, 8, 16), (40_000, 112, None, 138)])
def test_projection_kernels_at_benchmark_rows(shape, kind):
    """Row counts of the benchmark shard: every wave of the resident-weight kernel streams several
    row tiles (the load ring runs across tile boundaries) and the weight-gradient kernel splits the
    rows over ~680 slices plus the 1-4 leftover rows.  Checked against an fp64 product on the device
    (rocBLAS dgemm: an independent implementation) on the full output."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import new_padded
    m, k1, k2, n = shape
    gen = torch.Generator(device=dev()).manual_seed(m + n)

    def rnd(rows, cols):
        t = new_padded(rows, cols, dev())
        t.copy_(torch.randn(rows, cols, generator=gen, device=dev()))
        return t

    xa, xb = rnd(m, k1), (rnd(m, k2) if k2 else None)
    w = torch.randn(n, k1 + (k2 or 0), generator=gen, device=dev())
    bias = torch.randn(n, generator=gen, device=dev())
    g = rnd(m, n)
    Fn.set_projection(kind)
    try:
        img = Fn._image(w[:, :k1], w[:, k1:] if k2 else None, False)
        img_t = Fn._image(w[:, :k1], None, True)
        assert img.kind == kind
    finally:
        Fn.set_projection(Fn.default_projection())
    y = Fn._gemm_nt(xa, img, xb, bias, n)
    ref = xa.double() @ w[:, :k1].double().t() + bias.double()
    if k2:
        ref = ref + xb.double() @ w[:, k1:].double().t()
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 2e-6 * scale * (k1 + (k2 or 0)) ** 0.5
    gx = Fn._gemm_nt(g, img_t, None, None, k1)                            # grad-input product
    refx = g.double() @ w[:, :k1].double()
    assert float((gx.double() - refx).abs().max()) <= 2e-6 * float(refx.abs().max()) * n ** 0.5
    Fn.set_projection(kind)              # the weight-gradient kernels follow the projection arithmetic as well
    try:
        gw, gb = Fn._gemm_tn(g, xa, True)
        refw, refb = g.double().t() @ xa.double(), g.double().sum(0)
        # M-long fp32 accumulations in slices, fp64 across slices: error ~ 1e-7 * sqrt(rows per slice) * |terms|
        assert float((gw.double() - refw).abs().max()) <= 2e-6 * float(refw.abs().max()) + 1e-7 * m ** 0.5 * 4
        assert float((gb.double() - refb).abs().max()) <= 2e-6 * float(refb.abs().max()) + 1e-7 * m ** 0.5 * 4
        if k2:
            # both halves in one pass over g, written as column blocks of one matrix, then accumulated once more
            both = torch.zeros(n, k1 + k2, device=dev())
            bsum = torch.zeros(n, device=dev())
            Fn._gemm_tn2(g, xa, xb, True, both[:, :k1], both[:, k1:], bsum)
            refw2 = g.double().t() @ torch.cat([xa, xb], 1).double()
            tol = 2e-6 * float(refw2.abs().max()) + 1e-7 * m ** 0.5 * 4
            assert float((both.double() - refw2).abs().max()) <= tol
            assert float((bsum.double() - refb).abs().max()) <= 2e-6 * float(refb.abs().max()) + 1e-7 * m ** 0.5 * 4
            Fn._gemm_tn2(g, xa, xb, True, both[:, :k1], both[:, k1:], bsum, accumulate=True)
            assert float((both.double() - 2 * refw2).abs().max()) <= 2 * tol
    finally:
        Fn.set_projection(Fn.default_projection())


@pytest.mark.parametrize("m,n,k", [(1283529, 56, 138), (5003, 56, 138), (4097, 28, 61), (3, 56, 138), (5, 7, 9), (801, 16, 40), (333, 1, 138), (70001, 64, 140),
                                   (2049, 144, 33)])
def test_two_gradient_matrices_against_one_operand(m, n, k):
    """hgv_gemm_tn_gg: [Ga | Gb]^T A in one pass over A equals the two separate products (fp64 reference), with column
    sums of both, overwriting and accumulating, outputs as row blocks of one matrix."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import new_padded
    gen = torch.Generator(device=dev()).manual_seed(m + n + k)

    def rnd(rows, cols):
        t = new_padded(rows, cols, dev())
        t.copy_(torch.randn(rows, cols, generator=gen, device=dev()))
        return t

    ga, gb, x = rnd(m, n), rnd(m, n), rnd(m, k)
    both = torch.full((2 * n, k), 7.0, device=dev())
    sums = torch.full((2, n), 7.0, device=dev())
    Fn._gemm_tn_gg(ga, gb, x, both[:n], both[n:], sums[0], sums[1])
    ref = torch.cat([ga, gb], 1).double().t() @ x.double()
    refs = torch.stack([ga.double().sum(0), gb.double().sum(0)])
    tol = 2e-6 * float(ref.abs().max()) + 1e-7 * m ** 0.5 * 4
    tols = 2e-6 * float(refs.abs().max()) + 1e-7 * m ** 0.5 * 4
    assert float((both.double() - ref).abs().max()) <= tol
    assert float((sums.double() - refs).abs().max()) <= tols
    Fn._gemm_tn_gg(ga, gb, x, both[:n], both[n:], sums[0], sums[1], accumulate=True)
    assert float((both.double() - 2 * ref).abs().max()) <= 2 * tol
    assert float((sums.double() - 2 * refs).abs().max()) <= 2 * tols
    # fresh outputs: where the shape allows, one product on the 16-bit matrix cores (hgv_gemm_tn16_gg)
    wa, wb, ca, cb = Fn._gemm_tn_gg(ga, gb, x)
    assert float((torch.cat([wa, wb]).double() - ref).abs().max()) <= 1.5 * tol         # 1e-8 of the sum of |terms| at 1.28 M rows
    assert float((torch.stack([ca, cb]).double() - refs).abs().max()) <= 1.5 * tols
    sep_a, csa = Fn._gemm_tn(ga, x, True)
    sep_b, csb = Fn._gemm_tn(gb, x, True)
    for got, want in ((wa, sep_a), (wb, sep_b), (ca, csa), (cb, csb)):
        assert float((got.double() - want.double()).abs().max()) <= 2.5 * tol


def _scaled_cases():
    """Operands that exercise the power-of-two scales of the f16x3 kernels (name, rows, maker of A, maker of G)."""
    def randn(m, k, gen):
        return torch.randn(m, k, device=dev(), generator=gen)

    def chunks(sign):           # every 32-wide k chunk 300x larger (smaller) than the one before
        def make(m, k, gen):
            return randn(m, k, gen) * (300.0 ** (sign * torch.arange(k, device=dev()).div(32, rounding_mode="floor"))).float()
        return make

    def row_exp(m, span):       # the same exponent for a row in both sources of a product
        return ((torch.arange(m, device=dev()) * 7919) % (2 * span) - span).float().unsqueeze(1)

    def row_scaled(m, k, gen):  # rows between 1e-30 and 1e30
        return randn(m, k, gen) * 10.0 ** row_exp(m, 30)

    def one_hot(m, k, gen):     # sparse 0/1 rows, rows of zeros, rows of different magnitude
        x = (torch.rand(m, k, device=dev(), generator=gen) < 0.02).float()
        x[::7] = 0
        return x * 10.0 ** row_exp(m, 20)

    def growing(m, k, gen):     # later rows far larger than earlier ones
        return randn(m, k, gen) * (10.0 ** (torch.arange(m, device=dev()).float() / m * 12 - 6)).unsqueeze(1)

    def zeros_first(m, k, gen):
        x = randn(m, k, gen)
        x[: m // 3] = 0
        return x

    def col_scaled(m, k, gen):
        return randn(m, k, gen) * (10.0 ** torch.randint(-6, 1, (1, k), device=dev(), generator=gen).float())

    return [("rising chunks", 2048, chunks(1), randn), ("falling chunks", 2048, chunks(-1), randn),
            ("rows 1e-30..1e30", 4099, row_scaled, randn), ("one-hot rows", 4096, one_hot, randn),
            ("tiny", 4096, lambda m, k, gen: randn(m, k, gen) * 1e-12, lambda m, k, gen: randn(m, k, gen) * 1e-9),
            ("growing rows", 20011, growing, growing), ("falling rows", 20011, lambda m, k, gen: growing(m, k, gen).flip(0), randn),
            ("zero rows first", 20000, zeros_first, zeros_first), ("columns 1e-6..1", 20000, randn, col_scaled)]


@pytest.mark.parametrize("tn_f16", [False, True])
@pytest.mark.parametrize("case", _scaled_cases(), ids=lambda c: c[0])
@pytest.mark.parametrize("widths", [(138, 138, 138), (105, 0, 138), (256, 256, 256), (7, 0, 3)])
def test_scaled_fp16_products_keep_fp32_accuracy(case, widths, tn_f16, monkeypatch):
    """f16x3 scales every operand by a power of two before it is split into fp16 pieces: one scale per row of A (chosen
    from the row's first k chunk and lowered when a later chunk would overflow) and one per weight row.  Operands with
    rows or chunks of very different magnitude must still come out at fp32 accuracy: every output element within 1e-6 of
    ITS OWN sum of |terms| (the two sources of a row share their magnitude here: a term whose operands both lie more than
    2^16 below the largest entry of their rows keeps an absolute, not a relative, accuracy).  Weight gradients (2e-6 per element, 1e-6 of the largest sum for the matrix as a whole) run
    on the three-piece bf16 kernel by default; the optional fp16 weight-gradient kernel (one scale per workgroup and
    operand, lowered while the rows stream by) is held to the same bar on every case whose columns lie within its
    stated range (2^16 below the largest entry of a 144-column group)."""
    from hgvae_amd import functional as Fn
    name, m, make_a, make_g = case
    if tn_f16 and name in ("rising chunks", "falling chunks"):
        pytest.skip("columns 300^8 apart: outside the stated range of the one-scale-per-workgroup kernel")
    monkeypatch.setattr(Fn, "_tn_f16", tn_f16)
    k1, k2, n = widths
    gen = torch.Generator(device=dev()).manual_seed(7)
    xa, xb = Fn.as_padded(make_a(m, k1, gen)), (Fn.as_padded(make_a(m, k2, gen)) if k2 else None)
    wa = torch.randn(n, k1, device=dev(), generator=gen) * 0.1
    wb = torch.randn(n, k2, device=dev(), generator=gen) * 0.1 if k2 else None
    bias = torch.randn(n, device=dev(), generator=gen) * 0.01
    g = Fn.as_padded(make_g(m, n, gen))
    Fn.set_projection("f16x3")
    try:
        img = Fn._image(wa, wb, False)
        assert img.kind == "f16x3"
        y = Fn._gemm_nt(xa, img, xb, bias, n)
        x_all = torch.cat([xa[:, :k1], xb[:, :k2]], 1) if k2 else xa[:, :k1]
        w_all = torch.cat([wa, wb], 1) if k2 else wa
        want = x_all.double() @ w_all.double().t() + bias.double()
        terms = x_all.double().abs() @ w_all.double().abs().t() + bias.double().abs()
        assert torch.isfinite(y).all()
        assert float(((y[:, :n].double() - want).abs() / terms.clamp_min(1e-300)).max()) <= 1e-6
        if k2:
            ga, gb_, gbias = Fn._gemm_tn2(g, xa, xb, True)
            gw = torch.cat([ga, gb_], 1)
        else:
            gw, gbias = Fn._gemm_tn(g, xa, True)
        wantw = g[:, :n].double().t() @ x_all.double()
        termsw = g[:, :n].double().abs().t() @ x_all.double().abs()
        err = (gw.double() - wantw).abs()
        assert float(err.max()) <= 1e-6 * float(termsw.max())
        assert float((err / termsw.clamp_min(1e-300)).max()) <= 2e-6
        wantb, termsb = g[:, :n].double().sum(0), g[:, :n].double().abs().sum(0)
        assert float(((gbias.double() - wantb).abs() / termsb.clamp_min(1e-300)).max()) <= 2e-6
    finally:
        Fn.set_projection(Fn.default_projection())


@pytest.mark.parametrize("log2_range", [17, 20, 24, 30])
def test_scaled_fp16_products_small_entries_behind_a_large_one(log2_range):
    """The input-conditional clause of the f16x3 projections, pinned.  A row is scaled by ONE power of two chosen from its
    largest entry, so an entry more than 2^16 below that maximum reaches the fp16 pieces as a subnormal: it keeps an
    ABSOLUTE accuracy of 2^(e - 37) (e = floor(log2 of the row maximum)), not a relative one.  The adversarial pairing
    puts zero weights against the large entries, so the result consists of the small entries only and the usual measure
    (error / sum of |terms|) no longer hides the loss: the documented bound must hold, the error must really exceed the
    fp32-level bar where the clause applies (else the clause could be dropped), and the six-product bf16 kernel and the
    'auto' arithmetic - which sends such a launch to that kernel - must stay at fp32 level."""
    from hgvae_amd import functional as Fn
    m, k, n = 3000, 138, 138
    gen = torch.Generator(device=dev()).manual_seed(1000 + log2_range)
    big = 1.0e6
    small = big * 2.0 ** (-log2_range)
    x = (torch.rand(m, k, device=dev(), generator=gen) + 0.5) * small
    x[:, ::32] = big * (torch.rand(m, (k + 31) // 32, device=dev(), generator=gen) + 1.0)     # one large entry per 32-wide k chunk
    w = torch.randn(n, k, device=dev(), generator=gen)
    w[:, ::32] = 0.0                                                                          # ... meets a zero weight
    xa = Fn.as_padded(x)
    want = x.double() @ w.double().t()
    terms = x.double().abs() @ w.double().abs().t()
    e = torch.floor(torch.log2(x.abs().amax(dim=1, keepdim=True).double()))
    clause = w.double().abs().sum(dim=1).unsqueeze(0) * torch.exp2(e - 37)                    # sum_k |w_nk| 2^(e - 37)
    errs = {}
    try:
        for kind in ("f16x3", "bf16x6", "auto"):
            Fn.set_projection(kind)
            y = Fn._gemm_nt(xa, Fn._image(w, None, False), None, None, n)
            errs[kind] = (y[:, :n].double() - want).abs()
    finally:
        Fn.set_projection(Fn.default_projection())
    assert float((errs["f16x3"] / (clause + 1e-6 * terms)).max()) <= 1.0, "f16x3 beyond its documented absolute bound"
    assert float((errs["bf16x6"] / terms).max()) <= 1e-6
    assert float((errs["auto"] / terms).max()) <= 1e-6, "the guarded arithmetic must route this launch to the six-product kernel"
    if log2_range >= 24:
        assert float((errs["f16x3"] / terms).max()) > 1e-5       # the clause is real: fp32 would be exact to 1e-7 here


@pytest.mark.parametrize("m,n,k1,k2", [(1, 5, 3, 2), (4, 16, 16, 16), (5, 17, 33, 7), (37, 138, 138, 28), (1000, 56, 138, 138),
                                       (4099, 138, 105, 105), (3001, 276, 138, 6), (2500, 1, 276, 138), (9000, 256, 256, 256),
                                       (6000, 138, 276, 138), (5000, 150, 144, 145),
                                       # the smallest row counts the bf16 kernel takes (one to a few 32-row steps per workgroup, partial last step)
                                       (128, 138, 138, 138), (130, 138, 138, 138), (255, 144, 144, 144), (257, 138, 105, 138)])
def test_weight_gradient_pair_small_shapes(m, n, k1, k2):
    """hgv_gemm_tn2 on ragged shapes (rows not a multiple of 4, widths off the tile sizes, different K per source)."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import new_padded
    gen = torch.Generator(device=dev()).manual_seed(m * 7 + n)

    def rnd(rows, cols):
        t = new_padded(rows, cols, dev())
        t.copy_(torch.randn(rows, cols, generator=gen, device=dev()))
        return t

    g, xa, xb = rnd(m, n), rnd(m, k1), rnd(m, k2)
    ga, gb, bs = Fn._gemm_tn2(g, xa, xb, True)
    for got, x in ((ga, xa), (gb, xb)):
        ref = g.double().t() @ x.double()
        assert float((got.double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())) + 1e-7 * m ** 0.5 * 4
    refb = g.double().sum(0)
    assert float((bs.double() - refb).abs().max()) <= 2e-6 * max(1.0, float(refb.abs().max())) + 1e-7 * m ** 0.5 * 4


# ------------------------------------------------------------------ flat Adam / gradient clipping
@pytest.mark.parametrize("cfg", [dict(), dict(weight_decay=0.01), dict(max_grad_norm=1.0), dict(capturable=True)])
def test_flat_adam_matches_torch_adam(cfg):
    """Same parameters, same gradients, 6 steps: FlatAdam (one HIP launch over the flat buffers) against
    torch.optim.Adam (+ clip_grad_norm_) on the CPU in fp32 (autoencoder/vae_custom_pool.py:228-229)."""
    from hgvae_amd.optim import FlatAdam
    gen = torch.Generator().manual_seed(7)
    shapes = [(138, 105), (138,), (56, 138), (1,), (28, 17, 3)]
    ref = [torch.nn.Parameter(torch.randn(*s_, generator=gen)) for s_ in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev())) for p in ref]
    clip = cfg.get("max_grad_norm")
    ropt = torch.optim.Adam(ref, lr=3e-3, weight_decay=cfg.get("weight_decay", 0.0))
    mopt = FlatAdam(mine, lr=3e-3, **cfg)
    for step in range(6):
        grads = [torch.randn(*s_, generator=gen) * (10.0 if step % 2 else 0.1) for s_ in shapes]
        mopt.zero_grad()
        for p, q, g in zip(ref, mine, grads):
            p.grad = g.clone()
            q.grad.add_(g.to(dev()))          # gradients accumulate into the bucket views, as autograd does
        if clip is not None:
            want_norm = torch.nn.utils.clip_grad_norm_(ref, clip)
        ropt.step()
        mopt.step()
        if clip is not None:
            close(mopt.last_grad_norm, want_norm, 1e-6)
        for p, q in zip(ref, mine):
            close(q.detach(), p.detach(), 2e-6)
    sd = mopt.state_dict()
    assert sd["step"] == 6 and sd["exp_avg"].numel() == sum(p.numel() for p in ref)


def test_flat_adam_invalidates_packed_weights():
    """The update writes through raw pointers; the projection's packed-weight cache must still see it."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.optim import FlatAdam
    gen = torch.Generator().manual_seed(3)
    w = torch.nn.Parameter(torch.randn(12, 20, generator=gen).to(dev()))
    x = torch.randn(33, 20, generator=gen).to(dev())
    opt = FlatAdam([w], lr=0.5)
    y0 = Fn.dual_linear(x, w)
    y0.sum().backward()
    opt.step()
    y1 = Fn.dual_linear(x, w).detach()
    close(y1, x.double().cpu() @ w.detach().double().cpu().t(), 2e-6)
    assert float((y1 - y0.detach()).abs().max()) > 1e-2


def test_bn_act_counts_batches_and_accumulates_affine_grads_in_place():
    """num_batches_tracked is advanced by the statistics kernel (BatchNorm1d semantics), and with
    direct_weight_grads the affine gradients are added to existing .grad buffers."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator().manual_seed(2)
    bn = torch.nn.BatchNorm1d(10).to(dev())
    y = torch.randn(50, 10, generator=gen).to(dev()).requires_grad_(True)
    gout = torch.randn(50, 10, generator=gen).to(dev())
    out = Fn.bn_act(y, bn, True, 0.0, True)
    out.backward(gout)
    assert int(bn.num_batches_tracked) == 1
    want_w, want_b = bn.weight.grad.clone(), bn.bias.grad.clone()
    bn.weight.grad.fill_(1.0)
    bn.bias.grad.fill_(2.0)
    out = Fn.bn_act(y, bn, True, 0.0, True)
    with Fn.direct_weight_grads():
        out.backward(gout)
    assert int(bn.num_batches_tracked) == 2
    close(bn.weight.grad, (want_w + 1.0).cpu(), 1e-6)
    close(bn.bias.grad, (want_b + 2.0).cpu(), 1e-6)
