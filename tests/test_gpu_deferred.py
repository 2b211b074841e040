"""Levels whose normalisation is deferred to their consumers: the projection applies relu -> dropout while it stores,
the level keeps d = dropout(relu(y)) and the vectors of out = d * s + t, the consumers fold s / t in.  Every piece is
checked against the plain path (or float64 torch) and the whole step against the path that writes every matrix."""
import pytest
import torch

from graphs import path_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _defer_at_every_size(monkeypatch):
    """Production defers levels from 65 536 rows upwards (smaller ones are launch-bound); the tests use it at every size."""
    from hgvae_amd import functional as Fn
    monkeypatch.setattr(Fn, "DEFER_MIN_ROWS", 0)


def dev():
    return torch.device("cuda:0")


def _rand(shape, gen, scale=1.0):
    return torch.randn(*shape, device=dev(), generator=gen) * scale


class _PaddedCopy(torch.autograd.Function):
    """A fresh matrix in the padded layout holding x; the identity for autograd."""

    @staticmethod
    def forward(ctx, x):
        from hgvae_amd.graph import new_padded
        out = new_padded(x.size(0), x.size(1), x.device)
        out.copy_(x)
        return out

    @staticmethod
    def backward(ctx, g):
        return g


def _bn(h, gen):
    bn = torch.nn.BatchNorm1d(h).to(dev()).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(h, device=dev(), generator=gen) + 0.5)
        bn.bias.copy_(_rand((h,), gen, 0.1))
    return bn


def _stats(y, bn, relu, p, noise, fork=False, running=True):
    from hgvae_amd import functional as Fn
    mask, seed, offset, ctr = noise
    rm, rv, nbt = (bn.running_mean, bn.running_var, bn.num_batches_tracked) if running else (None, None, None)
    return Fn._BnStats.apply(y, bn.weight, bn.bias, rm, rv, relu, p, mask, seed, offset, ctr, bn.eps, 0.1, nbt, fork)


@pytest.mark.parametrize("n,h", [(64, 12), (4416, 138), (70001, 138), (1000, 256), (333, 31)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("noise_kind", ["mask", "philox"])
def test_in_place_activation_and_statistics_equal_the_plain_level(n, h, relu, noise_kind):
    """hgv_bn_stats_act leaves d = dropout(relu(y)) in y's place (dropped entries negative zeros when there is no relu, kept
    zeros positive) and hgv_bn_apply on it gives bit for bit what hgv_bn_act_fwd gives on y with the same noise: same
    statistics, same running buffers; col_s / col_t reproduce the normalisation as one multiply-add."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator(device=dev()).manual_seed(n + h)
    y = Fn.as_padded(_rand((n, h), gen))
    y[::5, ::3] = 0.0
    bn = _bn(h, gen)
    bn2 = torch.nn.BatchNorm1d(h).to(dev()).train()
    bn2.load_state_dict(bn.state_dict())
    p = 0.5
    mask = (torch.rand(n, h, device=dev(), generator=gen) >= p).to(torch.uint8)
    noise = (mask, 0, 0, None) if noise_kind == "mask" else (None, 77, 3 << 20, None)
    out = Fn.bn_act(y, bn, relu, p, True, noise=noise)
    yc = Fn.new_padded(n, h, dev())
    yc.copy_(y)
    handle, st, mean, rstd = _stats(yc, bn2, relu, p, noise)
    got = Fn.Deferred(handle, st[0], st[1], mean, rstd, bn2.weight, bn2.bias).tensor()
    assert torch.equal(got, out)
    assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    assert int(bn2.num_batches_tracked) == 1
    d = handle.detach()
    assert d.data_ptr() == yc.data_ptr()                                  # in place
    act = torch.relu(y) if relu else y
    neg_zero = d.view(torch.int32) == -2 ** 31
    if noise_kind == "mask":
        want = act * 2.0 * mask
        assert torch.equal(d.abs(), want.abs())
        if relu:
            assert not bool(neg_zero.any())
        else:
            assert torch.equal(neg_zero, mask == 0)
    else:
        kept = ~neg_zero if not relu else None
        if kept is not None:
            assert abs(float(kept.float().mean()) - (1 - p)) < 0.02
            assert torch.equal(d[kept], (act * 2.0)[kept])
    folded = d.double() * st[0, :h].double() + st[1, :h].double()
    assert float((folded - out.detach().double()).abs().max()) <= 2e-6 * max(float(out.detach().abs().max()), 1.0)
    if st.size(1) > h:
        assert float(st[:, h:].abs().max()) == 0.0


@pytest.mark.parametrize("relu", [True, False])
def test_backward_of_a_deferred_level_equals_the_plain_level(relu):
    """hgv_bn_act_bwd_pre reads relu / dropout off the stored values: same dy, dgamma, dbeta, bit for bit, as
    hgv_bn_act_bwd on y with the mask - with a second incoming gradient and with exact zeros in y."""
    from hgvae_amd import functional as Fn
    n, h, p = 5000, 138, 0.5
    gen = torch.Generator(device=dev()).manual_seed(9)
    src = _rand((n, h), gen)
    src[::7, ::3] = 0.0                                            # exact zeros: kept ones must not look dropped
    mask = (torch.rand(n, h, device=dev(), generator=gen) >= p).to(torch.uint8)
    g1, g2 = _rand((n, h), gen), _rand((n, h), gen)
    bn = _bn(h, gen)
    res = []
    for deferred in (False, True):
        leaf = src.clone().requires_grad_(True)
        y = _PaddedCopy.apply(leaf)
        bn.zero_grad()
        if deferred:
            outs = _stats(y, bn, relu, p, (mask, 0, 0, None), fork=True, running=False)
            oa, ob = outs[0], outs[1]
        else:
            oa, ob = Fn.bn_act(y, bn, relu, p, True, fork=True, noise=(mask, 0, 0, None))
        # both paths receive the gradients of `out`: by convention that is what flows into a deferred level's handles
        torch.autograd.backward([oa, ob], [g1, g2])
        res.append((leaf.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,h", [(5000, 138), (70001, 138), (777, 64)])
@pytest.mark.parametrize("which", ["both", "first", "second"])
def test_two_draws_of_one_pre_activation_as_two_deferred_levels(n, h, which):
    """hgv_bn_stats_act2 / hgv_bn_act_bwd2_pre: what hgv_bn_act_fwd2 / hgv_bn_act_bwd2 give on the same noise - outputs,
    running buffers, input gradient and dgamma / dbeta bit for bit (the sums taken from the stored values, no consumers
    registered) - also when only one of the two draws reaches the loss."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    gen = torch.Generator(device=dev()).manual_seed(n)
    src = _rand((n, h), gen)
    src[::7, ::3] = 0.0
    gs = [_rand((n, h), gen) for _ in range(4)]
    bn = _bn(h, gen)
    bn2 = torch.nn.BatchNorm1d(h).to(dev()).train()
    bn2.load_state_dict(bn.state_dict())
    res = []
    for deferred, norm in ((False, bn), (True, bn2)):
        leaf = src.clone().requires_grad_(True)
        y = _PaddedCopy.apply(leaf)
        with use_noise(PhiloxNoise(1234)):
            if deferred:
                with Fn.deferred_levels():
                    pair = Fn.bn_level_pair(y, norm, True, 0.5)
                assert all(isinstance(t, Fn.Deferred) for draw in pair for t in draw)
                outs = [t.tensor() for draw in pair for t in draw]
            else:
                pair = Fn.bn_act_pair(y, norm, True, 0.5)
                outs = [t for draw in pair for t in draw]
        sel = {"both": [0, 1, 2, 3], "first": [0, 1], "second": [3]}[which]
        torch.autograd.backward([outs[i] for i in sel], [gs[i] for i in sel])
        res.append(([o.detach().clone() for o in outs], leaf.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()))
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    assert int(bn2.num_batches_tracked) == 2
    for a, b in zip(res[0][1:], res[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,f_in,n_out", [(3001, 105, 138), (40_007, 105, 138), (50_001, 100, 129), (5000, 60, 138), (120, 105, 138),
                                          (7, 105, 138)])
@pytest.mark.parametrize("relu", [True, False])
def test_projection_stores_both_draws_of_a_shared_level(n, f_in, n_out, relu, monkeypatch):
    """hgv_gemm_nt16h_act2_stats (+ two hgv_dropout_bits, two hgv_bn_stats_merge) against the projection followed by
    hgv_bn_stats_act2: both stored matrices are the same bits (signed zeros included), the statistics, s / t and the running
    buffers agree to rounding, every gradient likewise; source widths outside 97...112 keep the pass of their own."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    from hgvae_amd.noise import PhiloxNoise, use_noise
    gen = torch.Generator(device=dev()).manual_seed(n + n_out)
    ei, _, _ = path_batch([n])
    ei = torch.cat([ei, torch.randint(0, n, (2, 4000))], 1).to(dev())
    graph = graph_of(ei, n)
    x0 = _rand((n, f_in), gen)
    w1, w2, bias = _rand((n_out, f_in), gen, 0.1), _rand((n_out, f_in), gen, 0.1), _rand((n_out,), gen, 0.1)
    gs = [_rand((n, n_out), gen) for _ in range(4)]
    results = []
    for in_product in (True, False):
        monkeypatch.setattr(Fn, "_pair_in_product", in_product)
        bn = _bn(n_out, torch.Generator(device=dev()).manual_seed(5))
        x = x0.clone().requires_grad_(True)
        wa, wb, b = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        monkeypatch.setattr(Fn._lib, "call", spy_call)
        with use_noise(PhiloxNoise(4321)), Fn.deferred_levels():
            pair, y = Fn.bn_level_pair(lambda: Fn.sage_conv(x, graph, wa, b, wb, mean=True), bn, relu, 0.5, rows=n)
        monkeypatch.setattr(Fn._lib, "call", real_call)
        covered = f_in > 96
        assert ("hgv_gemm_nt16h_act2_stats" in names) == (in_product and covered), names
        assert ("hgv_bn_stats_act2" in names) == (not (in_product and covered)), names
        assert names.count("hgv_dropout_bits") == (2 if in_product and covered else 0)
        levels = [t for draw in pair for t in draw]
        assert all(isinstance(t, Fn.Deferred) for t in levels)
        ds = [levels[0].d.detach().clone(), levels[2].d.detach().clone()]
        torch.autograd.backward([t.tensor() for t in levels], gs)
        results.append((ds, [levels[0].s.clone(), levels[0].t.clone(), levels[2].s.clone(), levels[2].t.clone(), bn.running_mean.clone(),
                             bn.running_var.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(), x.grad.clone(), wa.grad.clone(),
                             wb.grad.clone(), b.grad.clone()], int(bn.num_batches_tracked)))
    (da, va, ta), (db, vb, tb) = results
    assert ta == tb == 2
    for u, v in zip(da, db):
        assert torch.equal(u.view(torch.int32), v.view(torch.int32))
    for u, v in zip(va, vb):
        assert float((u.double() - v.double()).abs().max()) <= 3e-6 * max(float(v.abs().max()), 1e-30), (float((u - v).abs().max()), float(v.abs().max()))


@pytest.mark.parametrize("op,relu,n_out", [("sage", True, 138), ("concat", False, 138), ("dual", True, 144), ("dual", False, 129),
                                           ("sage", True, 64), ("concat", True, 160)])
@pytest.mark.parametrize("in_product_stats", [False, True])
@pytest.mark.parametrize("n", [3001, 40_007])
def test_projection_stores_the_activated_level_itself(op, relu, n_out, in_product_stats, n, monkeypatch):
    """hgv_dropout_bits + hgv_gemm_nt16h_act + hgv_bn_stats_pre against the projection followed by hgv_bn_stats_act:
    the stored d (signed zeros included), s / t, the running statistics and every gradient are the same bits; widths the
    activated store does not cover take the other path by themselves.  With the column sums taken inside the product
    (hgv_gemm_nt16h_act_stats + hgv_bn_stats_merge) d is still the same bits and the statistics agree to rounding."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    from hgvae_amd.noise import PhiloxNoise, use_noise
    k = 138               # 40 007 rows: long enough for the projections that keep the weights in registers
    gen = torch.Generator(device=dev()).manual_seed(n_out + int(relu))
    ei, nn_, _ = path_batch([n])
    ei = torch.cat([ei, torch.randint(0, n, (2, 4000))], 1).to(dev())
    graph = graph_of(ei, n)
    xa0, xb0 = _rand((n, k), gen), _rand((n, k), gen)
    w1, w2 = _rand((n_out, k), gen, 0.1), _rand((n_out, k), gen, 0.1)
    bias = _rand((n_out,), gen, 0.1)
    g1, g2 = _rand((n, n_out), gen), _rand((n, n_out), gen)
    results = []
    monkeypatch.setattr(Fn, "_nt_stats", in_product_stats)
    for act in (True, False):
        monkeypatch.setattr(Fn, "_nt_act", act)
        bn = _bn(n_out, torch.Generator(device=dev()).manual_seed(5))
        xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(True)
        wa, wb, b = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True), bias.clone().requires_grad_(True)
        if op == "sage":
            make = lambda: Fn.sage_conv(xa, graph, wa, b, wb, mean=True)                  # noqa: E731
        elif op == "concat":
            wcat = torch.cat([wa, wb], 1)
            make = lambda: Fn.concat_linear(xa, xb, wcat, b)                              # noqa: E731
        else:
            make = lambda: Fn.dual_linear(xa, wa, xb, wb, b)                              # noqa: E731
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        monkeypatch.setattr(Fn._lib, "call", spy_call)
        with use_noise(PhiloxNoise(4321)), Fn.deferred_levels():
            lvl_a, lvl_b = Fn.bn_level(make, bn, relu, 0.5, True, fork=True, act_ok=True)
        monkeypatch.setattr(Fn._lib, "call", real_call)
        covered = bool(Fn._lib.load().hgv_gemm_nt16h_act_supported(n_out, k, 0 if op == "none" else k))
        assert covered == (n_out in (138, 129)), (n_out, covered)      # one-part images of nine column tiles
        assert (("hgv_gemm_nt16h_act_stats" if in_product_stats else "hgv_gemm_nt16h_act") in names) == (act and covered)
        assert (("hgv_bn_stats_merge" if in_product_stats else "hgv_bn_stats_pre") in names) == (act and covered)
        assert isinstance(lvl_a, Fn.Deferred)
        d = lvl_a.d.detach().clone()
        torch.autograd.backward([lvl_a.tensor(), lvl_b.tensor()], [g1, g2])
        grads = [t.grad.clone() for t in (xa, wa, wb, b) if t.grad is not None] + ([xb.grad.clone()] if xb.grad is not None else [])
        results.append((d, lvl_a.s.clone(), lvl_a.t.clone(), bn.running_mean.clone(), bn.running_var.clone(),
                        bn.weight.grad.clone(), bn.bias.grad.clone(), grads))
    a, b_ = results
    assert torch.equal(a[0].view(torch.int32), b_[0].view(torch.int32))          # the same bits, signed zeros included
    assert len(a[7]) == len(b_[7])

    def same(u, v):
        if not in_product_stats:
            return torch.equal(u, v)
        return float((u.double() - v.double()).abs().max()) <= 3e-6 * max(float(v.abs().max()), 1e-30)
    for u, v in zip(a[1:7], b_[1:7]):
        assert same(u, v), (float((u - v).abs().max()), float(v.abs().max()))
    for u, v in zip(a[7], b_[7]):
        assert same(u, v), (float((u - v).abs().max()), float(v.abs().max()))


@pytest.mark.parametrize("p,offset", [(0.5, 0.0), (0.5, 300.0), (0.2, 300.0), (0.5, -1e4)])
@pytest.mark.parametrize("n,k", [(70001, 138), (130, 138), (50_003, 28)])
def test_in_product_statistics_with_a_large_common_offset(p, offset, n, k, monkeypatch):
    """The column sums taken inside the product are plain sums of d and d^2 per wave: a pre-activation whose mean is
    hundreds of standard deviations from zero (a large bias) must still give the batch statistics of the shifted sums of
    the statistics kernel - dropout's zeros are part of the variance, so mean^2 / variance of d stays below (1 - p) / p."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    n_out = 138
    gen = torch.Generator(device=dev()).manual_seed(int(offset) % 1000 + n)
    xa0 = _rand((n, k), gen)
    w = _rand((n_out, k), gen, 0.1)
    bias = torch.full((n_out,), offset, device=dev()) + _rand((n_out,), gen)
    out = []
    for in_product in (True, False):
        monkeypatch.setattr(Fn, "_nt_stats", in_product)
        bn = _bn(n_out, torch.Generator(device=dev()).manual_seed(5))
        xa = xa0.clone().requires_grad_(True)
        with use_noise(PhiloxNoise(77)), Fn.deferred_levels():
            lvl = Fn.bn_level(lambda: Fn.dual_linear(xa, w, None, None, bias), bn, False, p, True, act_ok=True)
        assert isinstance(lvl, Fn.Deferred)
        out.append((lvl.mean.clone(), lvl.rstd.clone(), lvl.s.clone(), lvl.t.clone(), bn.running_var.clone(), lvl.d.detach().clone()))
    a, b = out
    assert torch.equal(a[5].view(torch.int32), b[5].view(torch.int32))
    d64 = b[5][:, :n_out].double()
    mean64, var64 = d64.mean(0), d64.var(0, unbiased=False)
    for got in (a, b):
        assert float(((got[0].double() - mean64).abs() / var64.sqrt()).max()) <= 2e-6
        assert float((got[1].double() * (var64 + bn.eps).sqrt() - 1).abs().max()) <= 3e-6
    assert float(((a[2] - b[2]).abs() / b[2].abs().clamp_min(1e-30)).max()) <= 5e-6
    assert float(((a[3].double() - b[3].double()).abs()).max()) <= 5e-6 * max(1.0, float((mean64.abs() * b[2].abs().double()[:n_out]).max()))


@pytest.mark.parametrize("op", ["dual", "concat", "concat1", "head", "sage"])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("n,h", [(3001, 138), (40_007, 138), (5003, 64), (2999, 129), (7, 138)])
def test_grad_input_product_applies_the_level_backward(op, relu, n, h, monkeypatch):
    """hgv_bn_bwd_sums_alg + hgv_gemm_nt16h_bwd (the only consumer's grad-input product stores the level's dy) against the
    product followed by hgv_bn_act_bwd_pre_alg: every gradient behind the level - its producer's input, weight and bias,
    the batch-norm affine parameters - and the consumer's own are the same bits.  With a second consumer the product that
    runs last adds the share the first one wrote (one source of 129..160 columns or two of 129..144, each consumer on a
    handle of its own); other shapes, and two consumers of one handle, leave the level its own pass."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    from hgvae_amd.noise import PhiloxNoise, use_noise
    if op == "sage" and h != 138:
        pytest.skip("the two-source product with the level's backward is built for sources of 129...144 columns")
    gen = torch.Generator(device=dev()).manual_seed(n + int(relu))
    ei, _, _ = path_batch([n])
    ei = torch.cat([ei, torch.randint(0, n, (2, 4000))], 1).to(dev())
    graph = graph_of(ei, n)
    x0 = _rand((n, h), gen)
    w0, b0 = _rand((h, h), gen, 0.1), _rand((h,), gen, 0.1)
    other = _rand((n, h), gen)
    n_out = {"dual": 105, "concat": 138, "concat1": 138, "head": 56, "sage": 138}[op]
    w1, w2, b1 = _rand((n_out, h), gen, 0.1), _rand((n_out, h), gen, 0.1), _rand((n_out,), gen, 0.1)
    gout = _rand((n, n_out), gen)
    results = []
    for fused, twice in ((True, False), (False, False), (True, True), (False, True)):
        monkeypatch.setattr(Fn, "_bwd_fuse", fused)
        bn = _bn(h, torch.Generator(device=dev()).manual_seed(5))
        x = x0.clone().requires_grad_(True)
        wp, bp = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        wa, wb, b = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True), b1.clone().requires_grad_(True)
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        monkeypatch.setattr(Fn._lib, "call", spy_call)
        with use_noise(PhiloxNoise(99)), Fn.deferred_levels():
            forked = op in ("sage", "concat")        # a second handle that nobody takes leaves one consumer
            lvl = Fn.bn_level(lambda: Fn.dual_linear(x, wp, None, None, bp), bn, relu, 0.5, True, act_ok=True, fork=forked)
            lvl, lvl2 = lvl if forked else (lvl, lvl)
            assert isinstance(lvl, Fn.Deferred)
            if op == "dual":
                y = Fn.dual_linear(lvl, wa, None, None, b)
            elif op in ("concat", "concat1"):
                y = Fn.concat_linear(lvl, other, torch.cat([wa, wb], 1), b)
            else:
                y = Fn.sage_conv(lvl, graph, wa, b, wb, mean=True)
            loss = (y * gout).sum()
            if twice:                    # a second consumer of the level (its backward runs first: the node is younger)
                # through the level's second handle the two gradients reach it apart; through the same handle autograd adds them
                loss = loss + (Fn.dual_linear(lvl2, wa[:, :h], None, None, None) * gout).sum() * 0.5
            loss.backward()
        monkeypatch.setattr(Fn._lib, "call", real_call)
        applied = fused and (not twice or op in ("sage", "concat"))
        assert ("hgv_gemm_nt16h_bwd" in names) == applied, names
        assert ("hgv_bn_act_bwd_pre_alg" in names) == (not applied), names
        results.append([t.grad.clone() for t in (x, wp, bp, bn.weight, bn.bias, wa, b)] + ([wb.grad.clone()] if wb.grad is not None else []))
    for a, b in ((0, 1), (2, 3)):
        for u, v in zip(results[a], results[b]):
            assert torch.equal(u, v), (a, float((u - v).abs().max()), float(v.abs().max()))


@pytest.mark.parametrize("first,h_in", [(False, 138), (False, 140), (True, 138), (False, 61)])
def test_sage_convolution_on_a_deferred_operand(first, h_in):
    """SAGEConv over (d, s, t) = SAGEConv over the written-out d * s + t: forward, input gradient (with respect to `out`),
    both weight gradients and the bias gradient, to fp32 rounding."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import graph_of
    gen = torch.Generator().manual_seed(21)
    ei, n, _ = path_batch([300, 41, 2, 1, 500])
    extra = torch.randint(0, n, (2, 4000), generator=gen)
    ei = torch.cat([ei, extra], 1).to(dev())
    g = graph_of(ei, n)
    dgen = torch.Generator(device=dev()).manual_seed(22)
    h_out = 56 if first else 138
    d0 = torch.relu(_rand((n, h_in), dgen)) * (torch.rand(n, h_in, device=dev(), generator=dgen) > 0.5) * 2.0
    bn = torch.nn.BatchNorm1d(h_in).to(dev()).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(h_in, device=dev(), generator=dgen) + 0.5)
        bn.bias.copy_(_rand((h_in,), dgen, 0.2))
    wl, wr = _rand((h_out, h_in), dgen, 0.1).requires_grad_(True), _rand((h_out, h_in), dgen, 0.1).requires_grad_(True)
    bl = _rand((h_out,), dgen, 0.1).requires_grad_(True)
    gout = _rand((n, h_out), dgen)
    res = []
    for deferred in (True, False):
        d = Fn.new_padded(n, h_in, dev())
        d.copy_(d0)
        d.requires_grad_(True)
        outs = _stats(_PaddedCopy.apply(d), bn, False, 0.0, (None, 0, 0, None), running=False)     # d as it is
        x = Fn.Deferred(outs[0], outs[1][0], outs[1][1], outs[2], outs[3], bn.weight, bn.bias)
        for t in (wl, wr, bl):
            t.grad = None
        y = Fn.sage_conv(x if deferred else x.tensor(), g, wl, bl, wr, mean=True)
        (y * gout).sum().backward()
        res.append((y.detach().clone(), d.grad.clone(), wl.grad.clone(), wr.grad.clone(), bl.grad.clone()))
    for a, b, what in zip(res[0], res[1], ("y", "gd", "gwl", "gwr", "gb")):
        scale = max(float(b.abs().max()), 1e-30)
        assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("n,f", [(5000, 105), (333, 7), (40_001, 60)])
@pytest.mark.parametrize("weight", [1.0, 0.7, -2.0])
def test_mse_gradient_written_by_the_forward_pass(n, f, weight, monkeypatch):
    """hgv_mse_fwd_grad + hgv_mse_bwd_unless_unit against hgv_mse_fwd + hgv_mse_bwd: same loss and the same gradient bits
    whether the loss enters the total with weight one (the forward pass has written the gradient, the backward pass only
    looks at the weight) or with another weight (the backward pass rewrites it) - also on a second backward pass over the
    same graph."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator(device=dev()).manual_seed(n + f)
    r0, x = _rand((n, f), gen), _rand((n, f), gen)
    res = []
    for early in (True, False):
        monkeypatch.setattr(Fn, "_mse_early", early)
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        monkeypatch.setattr(Fn._lib, "call", spy_call)
        r = r0.clone().requires_grad_(True)
        loss = Fn.mse_loss(Fn.as_padded(r), x)
        total = loss * weight if weight != 1.0 else loss + 0.0
        total.backward(retain_graph=True)
        g1 = r.grad.clone()
        r.grad = None
        (loss * 3.0).backward()
        monkeypatch.setattr(Fn._lib, "call", real_call)
        assert ("hgv_mse_fwd_grad" in names) == early and ("hgv_mse_bwd_unless_unit" in names) == early
        res.append((loss.detach().clone(), g1, r.grad.clone()))
    for u, v in zip(res[0], res[1]):
        assert torch.equal(u, v)
    ref = torch.nn.functional.mse_loss(r0.double(), x.double())
    assert abs(float(res[0][0]) - float(ref)) <= 1e-6 * float(ref)
    want = 2.0 * weight * (r0.double() - x.double()) / (n * f)
    assert float((res[0][1].double() - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.parametrize("variant", ["skipcon", "attention", "pool", "custom_pool"])
def test_train_step_with_deferred_levels_matches_the_plain_path(variant):
    """One reference train step with explicit noise: deferral on against deferral off - loss, every gradient, the
    BatchNorm buffers - and the deferred path really ran (no normalised matrix is written for the deferred levels)."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd import vae
    from tapes import custom_pool_tape, pool_tape, skipcon_tape
    gen = torch.Generator().manual_seed(31)
    ei, n, batch = path_batch([120, 77, 300, 2, 45])
    extra = torch.randint(0, n, (2, 900), generator=gen)
    ei = torch.cat([ei, extra], 1)
    f_in, h, z, ns = 12, 40, 6, 3
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(31)
    if variant == "skipcon":
        model, loss_fn = vae.SkipconVAE(f_in, h, z, ns), vae.train_step_loss
        tape = skipcon_tape(n, h, z, ns, 0.5, gen)
    elif variant == "attention":
        model, loss_fn = vae.SkipconAttentionVAE(f_in, h, z, ns), vae.train_step_loss
        e_loops = int((ei[0] != ei[1]).sum()) + n
        tape = skipcon_tape(n, h, z, ns, 0.49, gen, gat_edges=e_loops)
    elif variant == "pool":
        model, loss_fn = vae.PoolVAE(f_in, h, z, ns), vae.train_step_loss
        tape = pool_tape(n, h, z, ns, gen)
    else:
        model = vae.CustomPoolVAE(f_in, h, z, ns)
        loss_fn = lambda m, a, b, c, _l: vae.custom_pool_step_loss(m, a, b, c)      # noqa: E731
        tape = custom_pool_tape(n, h, z, ns, gen)
    model = model.to(dev()).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xd, eid, bd = x.to(dev()), ei.to(dev()), batch.to(dev())
    results, counts = [], []
    for deferred in (True, False):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        Fn.set_deferred_levels(deferred)
        made = []
        orig = Fn.Deferred.__init__

        def spy(self, *a, _orig=orig, _made=made, **k):
            _made.append(1)
            _orig(self, *a, **k)
        Fn.Deferred.__init__ = spy
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        Fn._lib.call = spy_call
        try:
            with use_noise(NoiseTape([t.clone() for t in tape])):
                loss = loss_fn(model, xd, eid, bd, 1.0)
            loss.backward()
        finally:
            Fn.Deferred.__init__ = orig
            Fn._lib.call = real_call
            Fn.set_deferred_levels(True)
        counts.append(len(made))
        if deferred:
            # levels whose consumers are all projections on (d, s, t) take their backward sums from those consumers'
            # products; levels read another way (pooling reads the written-out matrix) from the matrices
            assert "hgv_bn_act_bwd_pre_alg" in names, variant
            if variant in ("pool", "custom_pool"):
                assert "hgv_bn_act_bwd_pre" in names
        results.append((loss.detach().clone(), {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()},
                        {k: b.clone() for k, b in model.named_buffers()}))
    assert counts[0] > 0 and counts[1] == 0
    (la, ga, ba), (lb, gb, bb) = results
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    for k in gb:
        if gb[k] is None:
            assert ga[k] is None or float(ga[k].abs().max()) == 0.0, k
            continue
        scale = max(float(gb[k].abs().max()), 1e-30)
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-4 * scale + 1e-7, (k, float((ga[k] - gb[k]).abs().max()), scale)
    for k in bb:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-5, atol=1e-6), k


def test_step_is_the_same_bits_with_and_without_the_overlaps_and_fusions(monkeypatch):
    """At the benchmark's widths (105 features, hidden 138, latent 28) a whole train step gives the same loss and gradients, bit for bit,
    whether the decoder's keep bits are generated levels ahead or in front of each product, and whether the levels with one
    consumer have their backward applied by that consumer's grad-input product or by their own pass."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    from hgvae_amd import vae
    gen = torch.Generator().manual_seed(11)
    ei, n, batch = path_batch([700, 77, 300, 2, 45])
    ei = torch.cat([ei, torch.randint(0, n, (2, 2000), generator=gen)], 1)
    x = torch.randn(n, 105, generator=gen)
    torch.manual_seed(11)
    model = vae.SkipconVAE(105, 138, 28, 3).to(dev()).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xd, eid, bd = x.to(dev()), ei.to(dev()), batch.to(dev())
    results = []
    for ahead, fuse in ((True, True), (False, True), (True, False), (False, False)):
        monkeypatch.setattr(Fn, "_bits_ahead", ahead)
        monkeypatch.setattr(Fn, "_bwd_fuse", fuse)
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        monkeypatch.setattr(Fn._lib, "call", spy_call)
        with use_noise(PhiloxNoise(2024)):
            loss = vae.train_step_loss(model, xd, eid, bd, 1.0)
        loss.backward()
        monkeypatch.setattr(Fn._lib, "call", real_call)
        # fused: the three decoder levels, the third encoder level of both passes and the second level of both (in the first
        # pass it also feeds a skip connection: the product that runs last adds the other's share)
        assert names.count("hgv_gemm_nt16h_bwd") == (7 if fuse else 0), names.count("hgv_gemm_nt16h_bwd")
        assert names.count("hgv_dropout_bits") == 9          # seven levels, the first one shared by the passes with two draws
        assert names.count("hgv_gemm_nt16h_act2_stats") == 1
        assert not Fn._act_queue
        results.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                        {k: b.clone() for k, b in model.named_buffers()}))
    for other in results[1:]:
        assert torch.equal(results[0][0], other[0])
        assert results[0][1].keys() == other[1].keys()
        for k in other[1]:
            assert torch.equal(results[0][1][k], other[1][k]), k
        for k in other[2]:
            assert torch.equal(results[0][2][k], other[2][k]), k


@pytest.mark.parametrize("variant", ["skipcon", "attention"])
def test_counter_noise_step_defers_the_shared_first_level_for_both_passes(variant):
    """With counter-based noise the first encoder level serves both encode passes from one pass over the convolution
    output (two dropout draws): deferred (hgv_bn_stats_act2 / hgv_bn_act_bwd2_pre, sums from the consumers' products)
    against the written-out pair - same loss, gradients, BatchNorm buffers."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import PhiloxNoise, use_noise
    from hgvae_amd import vae
    gen = torch.Generator().manual_seed(5)
    ei, n, batch = path_batch([120, 77, 300, 2, 45])
    ei = torch.cat([ei, torch.randint(0, n, (2, 900), generator=gen)], 1)
    f_in, h, z, ns = 12, 40, 6, 3
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(5)
    cls = vae.SkipconVAE if variant == "skipcon" else vae.SkipconAttentionVAE
    model = cls(f_in, h, z, ns).to(dev()).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xd, eid, bd = x.to(dev()), ei.to(dev()), batch.to(dev())
    results = []
    for deferred in (True, False):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        Fn.set_deferred_levels(deferred)
        names = []
        real_call = Fn._lib.call

        def spy_call(name, *a, _names=names, _real=real_call):
            _names.append(name)
            return _real(name, *a)
        Fn._lib.call = spy_call
        try:
            with use_noise(PhiloxNoise(99)):
                loss = vae.train_step_loss(model, xd, eid, bd, 1.0)
            loss.backward()
        finally:
            Fn._lib.call = real_call
            Fn.set_deferred_levels(True)
        if deferred:
            assert "hgv_bn_stats_act2" in names and "hgv_bn_act_bwd2_pre" in names and "hgv_bn_act_fwd2" not in names
        else:
            assert "hgv_bn_act_fwd2" in names and "hgv_bn_stats_act2" not in names
        results.append((loss.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                        {k: b.clone() for k, b in model.named_buffers()}))
    (la, ga, ba), (lb, gb, bb) = results
    assert abs(float(la) - float(lb)) <= 1e-5 * abs(float(lb))
    assert ga.keys() == gb.keys()
    for k in gb:
        scale = max(float(gb[k].abs().max()), 1e-30)
        assert float((ga[k] - gb[k]).abs().max()) <= 1e-4 * scale + 1e-7, (k, float((ga[k] - gb[k]).abs().max()), scale)
    for k in bb:
        assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize("graph", ["paths", "random", "ragged", "edgeless"])
def test_skipcon_step_with_deferred_levels_matches_the_oracle(graph):
    """The oracle comparisons of tests/test_gpu_model.py, run with deferral at every size (the fixture of this file): with
    the oracle's explicit masks the step takes the same kernels as production - the projections store the activated
    levels and their column sums, the backward sums come from the consumers' products, and the levels with projections as
    consumers get their backward from those consumers' grad-input products."""
    import test_gpu_model as M
    from hgvae_amd import functional as Fn
    names = []
    real_call = Fn._lib.call

    def spy_call(name, *a, _real=real_call):
        names.append(name)
        return _real(name, *a)
    Fn._lib.call = spy_call
    try:
        M.test_skipcon_train_step_matches_oracle(graph)
    finally:
        Fn._lib.call = real_call
    for expected in ("hgv_dropout_bits", "hgv_gemm_nt16h_act_stats", "hgv_bn_stats_merge", "hgv_bn_act_bwd_pre_alg", "hgv_agg_fwd_ind",
                     "hgv_bn_bwd_sums_alg", "hgv_gemm_nt16h_bwd"):
        assert expected in names, expected


def test_other_variants_with_deferred_levels_match_the_oracle():
    import test_gpu_golden as G
    import test_gpu_model as M
    M.test_skipcon_train_step_matches_float64_oracle_on_protein_like_batch()
    for sizes in ([138] * 3, [1, 2, 90, 1]):
        M.test_attention_train_step_matches_oracle(sizes)
    for sizes in ([454, 200, 91], [1, 2, 77, 1, 3]):
        M.test_pool_vae_train_step_and_pool_outputs_match_oracle(sizes)
    for sizes in ([300, 154], [1, 61, 2, 1]):
        M.test_custom_pool_vae_train_step_matches_oracle(sizes)
    for tag in ("skipcon", "attention", "pool", "custom_pool"):
        G.test_train_step_matches_golden(tag)


def test_captured_step_with_deferred_levels_replays_and_learns():
    import test_gpu_model as M
    M.test_graphed_train_step_replays_with_fresh_noise_and_learns()
    for variant in ("skipcon", "attention"):
        M.test_captured_step_with_the_second_pass_on_a_branch_stream(variant)
