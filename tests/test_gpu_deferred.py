"""Levels whose normalisation is deferred to their consumers: the projection applies relu -> dropout while it stores,
the level keeps d = dropout(relu(y)) and the vectors of out = d * s + t, the consumers fold s / t in.  Every piece is
checked against the plain path (or float64 torch) and the whole step against the path that writes every matrix."""
import pytest
import torch

from graphs import path_batch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _rand(shape, gen, scale=1.0):
    return torch.randn(*shape, device=dev(), generator=gen) * scale


@pytest.mark.parametrize("m,k1,k2,n", [(37, 60, 0, 138), (1000, 138, 138, 138), (515, 28, 0, 138), (300, 138, 138, 56),
                                       (4099, 105, 105, 138), (700, 256, 256, 256), (64, 12, 0, 32)])
@pytest.mark.parametrize("relu", [True, False])
def test_projection_with_fused_activation_explicit_mask(m, k1, k2, n, relu):
    """hgv_gemm_nt16h_act with an explicit keep mask = dropout(relu(plain product)); without a relu the dropped entries
    are negative zeros and kept zeros positive ones."""
    from hgvae_amd import functional as Fn
    if not Fn._act_ok(n, k1, k2):
        pytest.skip("image of two parts: no fused activation for this shape")
    gen = torch.Generator(device=dev()).manual_seed(m + n)
    xa, xb = Fn.as_padded(_rand((m, k1), gen)), (Fn.as_padded(_rand((m, k2), gen)) if k2 else None)
    wa, wb = _rand((n, k1), gen, 0.1), (_rand((n, k2), gen, 0.1) if k2 else None)
    bias = _rand((n,), gen, 0.1)
    p = 0.5
    mask = (torch.rand(m, n, device=dev(), generator=gen) >= p).to(torch.uint8)
    img = Fn._image(wa, wb, False)
    assert img.kind == "f16x3"
    y = Fn._gemm_nt(xa, img, xb, bias, n)
    act = Fn.ActSpec(relu, p, mask, 0, 0, None)
    d = Fn._gemm_nt(xa, img, xb, bias, n, act)
    want = (torch.relu(y) if relu else y) * 2.0 * mask
    assert torch.equal(d[:, :n] == 0, want[:, :n] == 0)
    assert torch.equal(d[:, :n].abs(), want[:, :n].abs())                 # same bits up to the sign of zeros
    neg_zero = (d[:, :n].view(torch.int32) == -2 ** 31)
    if relu:
        assert not bool(neg_zero.any())
    else:
        assert torch.equal(neg_zero, mask == 0)
    if d.stride(0) > n:                                                   # pad columns stay zero
        pads = d.as_strided((m, d.stride(0) - n), (d.stride(0), 1), d.storage_offset() + n)
        assert float(pads.abs().max()) == 0.0


def test_projection_with_fused_activation_philox():
    """Counter-based keep decisions: reproducible for one (seed, offset), different for another, keep rate 1 - p, and
    unrelated between neighbouring rows / column groups."""
    from hgvae_amd import functional as Fn
    m, k, n, p = 20000, 138, 138, 0.49
    gen = torch.Generator(device=dev()).manual_seed(3)
    xa, w = Fn.as_padded(_rand((m, k), gen)), _rand((n, k), gen, 0.1)
    img = Fn._image(w, None, False)
    y = Fn._gemm_nt(xa, img, None, None, n)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev())

    def run(seed, offset):
        return Fn._gemm_nt(xa, img, None, None, n, Fn.ActSpec(False, p, None, seed, offset, ctr.data_ptr()))

    a, b, c = run(7, 1 << 20), run(7, 1 << 20), run(7, 2 << 20)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    kept_a, kept_c = a[:, :n].view(torch.int32) != -2 ** 31, c[:, :n].view(torch.int32) != -2 ** 31
    assert not torch.equal(kept_a, kept_c)
    rate = float(kept_a.float().mean())
    assert abs(rate - (1 - p)) < 3e-3, rate
    scale = 1.0 / (1.0 - p)
    ref = y[:, :n] * torch.tensor(scale, dtype=torch.float32)
    assert torch.equal(a[:, :n][kept_a], ref[kept_a])
    col_rate = kept_a.float().mean(0)
    assert float((col_rate - (1 - p)).abs().max()) < 0.02                  # every column sees the same probability
    agree = (kept_a[:, :-1] == kept_a[:, 1:]).float().mean()
    assert abs(float(agree) - (p * p + (1 - p) ** 2)) < 5e-3               # neighbouring columns are independent
    ctr.fill_(1 << 33)
    assert not torch.equal(kept_a, run(7, 1 << 20)[:, :n].view(torch.int32) != -2 ** 31)    # the device counter moves the stream


@pytest.mark.parametrize("n,h", [(64, 12), (4416, 138), (70001, 138), (1000, 256), (333, 31)])
def test_statistics_of_an_activated_matrix_equal_the_plain_level(n, h):
    """hgv_bn_stats / hgv_bn_apply on d = dropout(relu(y)) give bit for bit what hgv_bn_act_fwd gives on y with the same
    mask; col_s / col_t reproduce the normalisation as one multiply-add."""
    from hgvae_amd import functional as Fn
    gen = torch.Generator(device=dev()).manual_seed(n + h)
    y = Fn.as_padded(_rand((n, h), gen))
    bn = torch.nn.BatchNorm1d(h).to(dev()).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(h, device=dev(), generator=gen) + 0.5)
        bn.bias.copy_(_rand((h,), gen, 0.1))
    bn2 = torch.nn.BatchNorm1d(h).to(dev()).train()
    bn2.load_state_dict(bn.state_dict())
    p = 0.5
    mask = (torch.rand(n, h, device=dev(), generator=gen) >= p).to(torch.uint8)
    out = Fn.bn_act(y, bn, True, p, True, noise=(mask, 0, 0, None))
    d = Fn.new_padded(n, h, dev())
    d.copy_(torch.relu(y) * 2.0 * mask)
    outs = Fn._BnStats.apply(d, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, True, p, bn2.eps, 0.1,
                             bn2.num_batches_tracked, False)
    handle, st, mean, rstd = outs
    got = Fn.Deferred(handle, st[0], st[1], mean, rstd, bn2.weight, bn2.bias).tensor()
    assert torch.equal(got, out)
    assert torch.equal(bn.running_mean, bn2.running_mean) and torch.equal(bn.running_var, bn2.running_var)
    assert int(bn2.num_batches_tracked) == 1
    folded = d.double() * st[0, :h].double() + st[1, :h].double()
    assert float((folded - out.double()).abs().max()) <= 2e-6 * max(float(out.abs().max()), 1.0)
    assert float(st[:, h:].abs().max()) == 0.0 if st.size(1) > h else True


def test_backward_of_an_activated_level_equals_the_plain_level():
    """hgv_bn_act_bwd_pre reads relu / dropout off the stored values: same dy, dgamma, dbeta as hgv_bn_act_bwd on y with the
    mask, with and without a relu and with a second incoming gradient."""
    from hgvae_amd import functional as Fn
    n, h, p = 5000, 138, 0.5
    gen = torch.Generator(device=dev()).manual_seed(9)
    for relu in (True, False):
        y = Fn.as_padded(_rand((n, h), gen)).requires_grad_(True)
        y.data[::7, ::3] = 0.0                                     # exact zeros: kept ones must not look dropped
        mask = (torch.rand(n, h, device=dev(), generator=gen) >= p).to(torch.uint8)
        g1, g2 = _rand((n, h), gen), _rand((n, h), gen)
        bn = torch.nn.BatchNorm1d(h).to(dev()).train()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(h, device=dev(), generator=gen) + 0.5)
        oa, ob = Fn.bn_act(y, bn, relu, p, True, fork=True, noise=(mask, 0, 0, None))
        (oa * g1).sum().add((ob * g2).sum()).backward()
        want = (y.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        bn.zero_grad()
        act = torch.relu(y.detach()) if relu else y.detach()
        dm = act * 2.0
        dm = torch.where(dm == 0, torch.zeros_like(dm), dm)        # kept zeros are +0
        d = Fn.new_padded(n, h, dev())
        d.copy_(torch.where(mask.bool(), dm, torch.full_like(dm, 0.0 if relu else -0.0)))
        d.requires_grad_(True)
        outs = Fn._BnStats.apply(d, bn.weight, bn.bias, None, None, relu, p, bn.eps, 0.1, None, True)
        (outs[0] * g1).sum().add((outs[1] * g2).sum()).backward()
        # the two handles stand for `out`: what flows back is taken as d(out), so weight them with s: the plain path's
        # incoming gradients were those of out = d * s + t as well
        assert torch.equal(d.grad, want[0]) and torch.equal(bn.weight.grad, want[1]) and torch.equal(bn.bias.grad, want[2])


@pytest.mark.parametrize("first", [False, True])
def test_sage_convolution_on_a_deferred_operand(first):
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
    h_in, h_out = 138, (56 if first else 138)
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
        outs = Fn._BnStats.apply(d, bn.weight, bn.bias, None, None, True, 0.5, bn.eps, 0.1, None, False)
        x = Fn.Deferred(outs[0], outs[1][0], outs[1][1], outs[2], outs[3], bn.weight, bn.bias)
        for t in (wl, wr, bl):
            t.grad = None
        y = Fn.sage_conv(x if deferred else x.tensor(), g, wl, bl, wr, mean=True)
        (y * gout).sum().backward()
        res.append((y.detach().clone(), d.grad.clone(), wl.grad.clone(), wr.grad.clone(), bl.grad.clone()))
    for a, b, what in zip(res[0], res[1], ("y", "gd", "gwl", "gwr", "gb")):
        scale = max(float(b.abs().max()), 1e-30)
        assert float((a - b).abs().max()) <= 2e-5 * scale, (what, float((a - b).abs().max()), scale)


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
        try:
            with use_noise(NoiseTape([t.clone() for t in tape])):
                loss = loss_fn(model, xd, eid, bd, 1.0)
            loss.backward()
        finally:
            Fn.Deferred.__init__ = orig
            Fn.set_deferred_levels(True)
        counts.append(len(made))
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
