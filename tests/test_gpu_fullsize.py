"""The headline configuration at its full single-GPU size (global batch 256 on one device: 1.285 M nodes, 15.4 M edges,
hidden width 138): every kernel family of the train step against an independent evaluation at that size - torch's own
device operators in float64 (sorting, index_add_, matmul, batch norm), which share no code with the HIP kernels.
Sizes beyond 2^31 bytes per matrix row block and beyond 2^24 rows are exactly what the small parity cases cannot reach."""
import pytest
import torch

pytestmark = pytest.mark.gpu

N, DEG, H, F_IN = 1_285_158, 12, 138, 105
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def big():
    from graphs import local_graph
    from hgvae_amd.graph import Graph
    ei, n, _ = local_graph(N, DEG, seed=21, window=64, far=0.02)
    ei = ei.to(DEV)
    g = Graph(ei, n)
    g.check()
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(n, H, device=DEV, generator=gen)
    yield ei, g, x, gen
    torch.cuda.empty_cache()


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def test_csr_at_full_size(big):
    ei, g, _, _ = big
    e = ei.size(1)
    assert e == N * DEG
    for csr, key_row, other_row in ((g.by_dst, 1, 0), (g.by_src, 0, 1)):
        order = torch.sort(ei[key_row], stable=True).indices          # torch's stable sort = the contract
        assert torch.equal(csr.perm.long(), order)
        assert torch.equal(csr.col.long(), ei[other_row][order])
        rp = torch.zeros(N + 1, dtype=torch.long, device=DEV)
        rp[1:] = torch.cumsum(torch.bincount(ei[key_row], minlength=N), 0)
        assert torch.equal(csr.rowptr.long(), rp)


def test_aggregation_forward_and_transposed_at_full_size(big):
    from hgvae_amd import functional as Fn
    ei, g, x, gen = big
    xr = x.detach().clone().requires_grad_(True)
    out = Fn.aggregate(xr, g, mean=True)
    deg = torch.bincount(ei[1], minlength=N).clamp(min=1).double().unsqueeze(1)
    want = torch.zeros(N, H, dtype=torch.float64, device=DEV).index_add_(0, ei[1], x.double()[ei[0]]) / deg
    assert rel(out, want) <= 2e-6
    gout = torch.randn(N, H, device=DEV, generator=gen)
    out.backward(gout)
    want_g = torch.zeros(N, H, dtype=torch.float64, device=DEV).index_add_(0, ei[0], (gout.double() / deg)[ei[1]])
    assert rel(xr.grad, want_g) <= 2e-6
    del want, want_g


@pytest.mark.parametrize("k2", [0, H])
def test_projection_products_at_full_size(big, k2):
    from hgvae_amd import functional as Fn
    _, _, x, gen = big
    xa = x.detach().clone().requires_grad_(True)
    xb = torch.randn(N, k2, device=DEV, generator=gen).requires_grad_(True) if k2 else None
    wa = (torch.randn(H, H, device=DEV, generator=gen) * 0.1).requires_grad_(True)
    wb = (torch.randn(H, k2, device=DEV, generator=gen) * 0.1).requires_grad_(True) if k2 else None
    bias = torch.randn(H, device=DEV, generator=gen).requires_grad_(True)
    y = Fn.dual_linear(xa, wa, xb, wb, bias)
    gout = torch.randn(N, H, device=DEV, generator=gen)
    y.backward(gout)
    rows = torch.cat([torch.arange(0, 4096, device=DEV), torch.randint(0, N, (8192,), device=DEV, generator=gen),
                      torch.arange(N - 4096, N, device=DEV)])
    want = xa.detach().double()[rows] @ wa.detach().double().t() + bias.detach().double()
    want_gx = gout.double()[rows] @ wa.detach().double()
    if k2:
        want = want + xb.detach().double()[rows] @ wb.detach().double().t()
        assert rel(xb.grad[rows], gout.double()[rows] @ wb.detach().double()) <= 2e-6
        assert rel(wb.grad, gout.double().t() @ xb.detach().double()) <= 1e-5
    assert rel(y[rows], want) <= 2e-6
    assert rel(xa.grad[rows], want_gx) <= 2e-6
    # the whole reduction over 1.285 M rows: fp32 partial sums per row slice, float64 across slices
    assert rel(wa.grad, gout.double().t() @ xa.detach().double()) <= 1e-5
    assert rel(bias.grad, gout.double().sum(0)) <= 1e-5


def test_two_source_product_rows_do_not_depend_on_the_row_count(big):
    """At this size the two halves of the 138 + 138 -> 138 weight are walked in step by paired workgroups; at a few
    thousand rows they are not.  A row of the product depends on that row of the inputs only, so slices computed on
    their own must match the full-size result bit for bit (any ordering effect of the pairing would show here)."""
    from hgvae_amd import functional as Fn
    _, _, x, gen = big
    xb = torch.randn(N, H, device=DEV, generator=gen)
    wa = torch.randn(H, H, device=DEV, generator=gen) * 0.1
    wb = torch.randn(H, H, device=DEV, generator=gen) * 0.1
    bias = torch.randn(H, device=DEV, generator=gen)
    with torch.no_grad():
        full = Fn.dual_linear(x, wa, xb, wb, bias)
        again = Fn.dual_linear(x, wa, xb, wb, bias)
        assert torch.equal(full, again)
        for lo, hi in ((0, 5000), (N // 2 - 37, N // 2 + 4099), (N - 3001, N)):
            part = Fn.dual_linear(x[lo:hi].contiguous(), wa, xb[lo:hi].contiguous(), wb, bias)
            assert torch.equal(part, full[lo:hi])


def test_two_source_weight_gradient_is_reproducible_at_full_size(big):
    """At this size the two workgroups that reduce one row slice against the two weight halves are placed on one XCD
    and kept in step (they read the same rows of the gradient).  Which of them runs ahead varies from launch to
    launch; the sums must not: two launches agree bit for bit, and a gradient with three quarters of its entries zero
    (the state behind relu and dropout) is still the float64 product."""
    from hgvae_amd import functional as Fn
    _, _, x, gen = big
    xb = torch.randn(N, H, device=DEV, generator=gen)
    gout = torch.randn(N, H, device=DEV, generator=gen)
    gout = gout * (torch.rand(N, H, device=DEV, generator=gen) < 0.25)
    grads = []
    for _ in range(2):
        wa = (torch.randn(H, H, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)) * 0.1).requires_grad_(True)
        wb = (torch.randn(H, H, device=DEV, generator=torch.Generator(device=DEV).manual_seed(6)) * 0.1).requires_grad_(True)
        bias = torch.zeros(H, device=DEV, requires_grad=True)
        Fn.dual_linear(x, wa, xb, wb, bias).backward(gout)
        grads.append((wa.grad.clone(), wb.grad.clone(), bias.grad.clone()))
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    assert rel(grads[0][0], gout.double().t() @ x.double()) <= 1e-5
    assert rel(grads[0][1], gout.double().t() @ xb.double()) <= 1e-5
    assert rel(grads[0][2], gout.double().sum(0)) <= 1e-5


def test_relu_dropout_batchnorm_at_full_size(big):
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    _, _, x, gen = big
    y = (x * 1.7 + 0.3).detach().requires_grad_(True)
    mask = (torch.rand(N, H, device=DEV, generator=gen) >= 0.5).to(torch.uint8)
    bn = torch.nn.BatchNorm1d(H).to(DEV)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(H, device=DEV, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(H, device=DEV, generator=gen) * 0.1)
    with use_noise(NoiseTape([mask])):
        out = Fn.bn_act(y, bn, True, 0.5, True)
    gout = torch.randn(N, H, device=DEV, generator=gen)
    out.backward(gout)
    y64 = y.detach().double().requires_grad_(True)
    d = torch.relu(y64) * mask.double() * 2.0
    mean, var = d.mean(0), d.var(0, unbiased=False)
    ref = (d - mean) / torch.sqrt(var + bn.eps) * bn.weight.detach().double() + bn.bias.detach().double()
    ref.backward(gout.double())
    assert rel(out, ref) <= 2e-6
    assert rel(y.grad, y64.grad) <= 5e-6
    assert rel(bn.running_mean, 0.1 * mean) <= 2e-6
    assert rel(bn.running_var, 0.9 + 0.1 * d.var(0, unbiased=True)) <= 2e-6


def test_train_step_runs_and_is_reproducible_at_full_size(big):
    """The whole train step at this size: finite loss, bitwise identical when repeated with the same noise stream
    (no atomics anywhere), and the loss of the union batch equals the loss recomputed by the float64 torch formulas
    from the model's own reconstruction / head outputs."""
    import hgvae_amd
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    ei, g, x, gen = big
    xin = torch.randn(N, F_IN, device=DEV, generator=gen)
    torch.manual_seed(0)
    model = SkipconVAE(F_IN, H, 28).to(DEV).train()
    losses, grads = [], []
    for _ in range(2):
        for bnm in model.modules():
            if isinstance(bnm, torch.nn.BatchNorm1d):
                bnm.reset_running_stats()
        model.zero_grad(set_to_none=True)
        hgvae_amd.manual_seed(77)
        loss = train_step_loss(model, xin, g, None, 1.0)
        loss.backward()
        losses.append(loss.detach().clone())
        grads.append(torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
    assert torch.isfinite(losses[0]) and torch.equal(losses[0], losses[1])
    assert torch.equal(grads[0], grads[1]) and bool(torch.isfinite(grads[0]).all())
    model.eval()
    with torch.no_grad():
        recon, mean, log_std = model(xin, g)
        got = hgvae_amd.functional.mse_loss(recon, xin) + model.kl_divergence(mean, log_std)
        want = ((recon.double() - xin.double()) ** 2).mean() + \
            (-0.5 * (1 + 2 * log_std.double() - mean.double() ** 2 - torch.exp(log_std.double()) ** 2).sum(-1)).mean()
    assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want))


def test_paced_two_part_product_at_hidden_width_256():
    """A 256 -> 256 projection keeps its weight image in LDS as two parts along N; from 256 MB of A upwards the two parts
    are walked in step by paired workgroups (the second reader of a row finds it in L2).  Rows must not depend on who
    ran ahead: two launches agree bit for bit, slices computed on their own equal the full result, and sampled rows match
    a float64 product."""
    from hgvae_amd import functional as Fn
    m, h = 300_000, 256
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(m, h, device=DEV, generator=gen)
    w = torch.randn(h, h, device=DEV, generator=gen) * 0.1
    bias = torch.randn(h, device=DEV, generator=gen)
    with torch.no_grad():
        full = Fn.dual_linear(x, w, None, None, bias)
        assert torch.equal(full, Fn.dual_linear(x, w, None, None, bias))
        for lo, hi in ((0, 4100), (m // 2 - 11, m // 2 + 2049), (m - 1000, m)):
            assert torch.equal(Fn.dual_linear(x[lo:hi].contiguous(), w, None, None, bias), full[lo:hi])
    rows = torch.cat([torch.arange(0, 2048, device=DEV), torch.randint(0, m, (4096,), device=DEV, generator=gen),
                      torch.arange(m - 2048, m, device=DEV)])
    want = x.double()[rows] @ w.double().t() + bias.double()
    assert rel(full[rows], want) <= 2e-6
    del full, x
    torch.cuda.empty_cache()


def test_paced_products_on_two_streams_at_once():
    """Paced launches keep their progress hints in a copy of the counters chosen from the stream handle, so two of them
    running at the same time on two streams do not read each other's hints; either way the hints never decide a value:
    both results equal the one-stream result bit for bit."""
    from hgvae_amd import functional as Fn
    m, h = 300_000, 256
    gen = torch.Generator(device=DEV).manual_seed(10)
    xs = [torch.randn(m, h, device=DEV, generator=gen) for _ in range(2)]
    w = torch.randn(h, h, device=DEV, generator=gen) * 0.1
    with torch.no_grad():
        want = [Fn.dual_linear(x, w) for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(DEV) for _ in range(2)]
        got = [None, None]
        for rep in range(3):
            for k, st in enumerate(streams):
                with torch.cuda.stream(st):
                    got[k] = Fn.dual_linear(xs[k], w)
            torch.cuda.synchronize()
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    del xs, want, got
    torch.cuda.empty_cache()
