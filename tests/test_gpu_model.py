"""Whole-train-step parity of the HIP path against the CPU oracle (same weights, same noise).

Tolerances (norm-wise, per tensor): forward/loss rel 1e-5, parameter gradients rel 1e-4,
the bar stated for this path (fp32 with re-associated sums).
"""
import pytest
import torch

from graphs import path_batch, random_graph
from oracle import models as OM
from oracle import pyg_ops as O

pytestmark = pytest.mark.gpu

FWD_RTOL, BWD_RTOL = 1e-5, 1e-4


def close(a, b, rtol, atol=1e-6, what=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item() if a.numel() else 0.0
    ref = b.abs().max().item() if b.numel() else 0.0
    assert err <= rtol * ref + atol, f"{what}: max err {err:.3e} vs scale {ref:.3e} (rtol {rtol})"


def skipcon_tape(n, h, z, ns, p, gen, level2_dropout=True, gat_edges=None):
    """Noise in the order one reference train step draws it (forward pass, then the second encode)."""
    def m(*shape):
        return torch.bernoulli(torch.full(shape, 1 - p), generator=gen)
    def enc():
        items = [m(n, h)]
        if gat_edges is not None:
            items.append(m(gat_edges, 1))
        elif level2_dropout:
            items.append(m(n, h))
        items.append(m(n, h))
        return items
    return enc() + [torch.randn(ns, n, z, generator=gen)] + [m(n, h), m(n, h), m(n, h)] + enc()


def randomize_bn(model, gen):
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=gen) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.1)


@pytest.mark.parametrize("graph", ["paths", "random", "ragged", "edgeless"])
def test_skipcon_train_step_matches_oracle(graph):
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    gen = torch.Generator().manual_seed(0)
    if graph == "paths":
        ei, n, batch = path_batch([138] * 4)
    elif graph == "ragged":                      # single-node and two-node graphs next to a long one
        ei, n, batch = path_batch([1, 2, 454, 1, 3])
    elif graph == "edgeless":                    # no edge at all: every aggregate is zero
        n, batch, ei = 37, torch.zeros(37, dtype=torch.long), torch.zeros(2, 0, dtype=torch.long)
    else:
        ei, n, batch = random_graph(600, 5000, seed=3)
    f_in, h, z, ns = 60, 138, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(0)
    ref = OM.SkipconVAE(f_in, h, z, ns)
    randomize_bn(ref, gen)
    ref.train()
    tape = skipcon_tape(n, h, z, ns, 0.5, gen)
    loss_ref, rec_ref, kl_ref = OM.skipcon_loss(ref, x, ei, batch, 1.0, O.NoiseTape(tape))
    loss_ref.backward()

    dev = torch.device("cuda:0")
    mod = SkipconVAE(f_in, h, z, ns)
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    # running buffers were already advanced by the oracle's step: restart both from defaults
    for k in state:
        if k.endswith("running_mean"):
            state[k].zero_()
        elif k.endswith("running_var"):
            state[k].fill_(1.0)
        elif k.endswith("num_batches_tracked"):
            state[k].zero_()
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    xd, eid, bd = x.to(dev), ei.to(dev), batch.to(dev)
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, xd, eid, bd, 1.0)
    loss.backward()
    close(loss, loss_ref, FWD_RTOL, what="loss")
    ref_params = dict(ref.named_parameters())
    for name, prm in mod.named_parameters():
        assert prm.grad is not None, name
        close(prm.grad, ref_params[name].grad, BWD_RTOL, what=name)
    ref_bufs = dict(ref.named_buffers())
    for name, buf in mod.named_buffers():
        close(buf.float(), ref_bufs[name].float(), FWD_RTOL, what=name)


def test_skipcon_train_step_matches_float64_oracle_on_protein_like_batch():
    """Eight protein-like graphs (~40 k nodes, ~480 k edges, mean degree 12, F_in = 105): the shape of the benchmark batch at a
    quarter of its size.  Reference = the oracle run in float64 (at this size the fp32 CPU oracle is itself ~1e-3 off)."""
    from hgvae_amd import synthetic
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    dev = torch.device("cuda:0")
    b = synthetic.protein_like_batch(8, seed=1000, device=dev)
    x, ei, bt = b.x.cpu(), b.edge_index.cpu(), b.batch.cpu()
    n = x.size(0)
    gen = torch.Generator().manual_seed(10)
    torch.manual_seed(10)
    ref32 = OM.SkipconVAE(105, 138, 28, 10)
    randomize_bn(ref32, gen)
    state = _fresh_state(ref32)
    tape = skipcon_tape(n, 138, 28, 10, 0.5, gen)
    ref = OM.SkipconVAE(105, 138, 28, 10).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
    ref.train()
    tape64 = [t.double() if t.is_floating_point() else t for t in tape]
    loss_ref = OM.skipcon_loss(ref, x.double(), ei, bt, 1.0, O.NoiseTape(tape64))[0]
    loss_ref.backward()
    mod = SkipconVAE(105, 138, 28, 10)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, b.x, b.edge_index, b.batch, 1.0)
    loss.backward()
    close(loss, loss_ref, 2e-6, what="loss")
    rp = dict(ref.named_parameters())
    for name, prm in mod.named_parameters():
        close(prm.grad, rp[name].grad, 5e-5, what=name)


def test_skipcon_eval_forward_matches_oracle():
    from hgvae_amd.vae import SkipconVAE
    gen = torch.Generator().manual_seed(1)
    ei, n, batch = path_batch([50, 77, 138])
    x = torch.randn(n, 60, generator=gen)
    torch.manual_seed(1)
    ref = OM.SkipconVAE(60, 138, 28)
    randomize_bn(ref, gen)
    with torch.no_grad():
        for mod_ in ref.modules():
            if isinstance(mod_, torch.nn.BatchNorm1d):
                mod_.running_mean.copy_(torch.randn(mod_.running_mean.shape, generator=gen) * 0.1)
                mod_.running_var.copy_(torch.rand(mod_.running_var.shape, generator=gen) + 0.5)
    ref.eval()
    eps = torch.randn(10, n, 28, generator=gen)
    with torch.no_grad():
        rec_ref, mean_ref, ls_ref = ref(x, ei, batch, O.NoiseTape([eps]))
    from hgvae_amd.noise import NoiseTape, use_noise
    dev = torch.device("cuda:0")
    mod = SkipconVAE(60, 138, 28)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(dev).eval()
    with torch.no_grad(), use_noise(NoiseTape([eps])):
        rec, mean, ls = mod(x.to(dev), ei.to(dev), batch.to(dev))
    close(mean, mean_ref, FWD_RTOL, what="mean")
    close(ls, ls_ref, FWD_RTOL, what="log_std")
    close(rec, rec_ref, FWD_RTOL, what="reconstruction")


def test_fork_sums_gradients_in_padded_layout():
    """functional.fork: two consumers of one tensor; the backward sum equals autograd's and stays 16-byte aligned."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import ld_of
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randn(301, 138, generator=gen)
    w1, w2 = torch.randn(138, 7, generator=gen).to(dev), torch.randn(138, 5, generator=gen).to(dev)
    seen = {}

    def run(forked):
        x = x0.to(dev).requires_grad_(True)
        y = x * 1.5
        if forked:
            y.register_hook(lambda g: seen.update(ld=ld_of(g)))
            a, b = Fn.fork(y)
        else:
            a = b = y
        ((a @ w1).sin().sum() + (b @ w2).cos().sum()).backward()
        return x.grad

    close(run(True), run(False), 1e-6, what="fork gradient")
    assert seen["ld"] % 4 == 0 and seen["ld"] >= 138
    # one consumer only: the other gradient is absent, nothing is added
    x = x0.to(dev).requires_grad_(True)
    a, _ = Fn.fork(x * 2.0)
    a.sum().backward()
    assert torch.equal(x.grad, torch.full_like(x.grad, 2.0))
    # no grad mode / no grad needed: plain pass-through
    t = x0.to(dev)
    p, q = Fn.fork(t)
    assert p is t and q is t


def test_node_embeddings_match_oracle():
    """Embedding export of the clustering step (gvae/gvae_skipcon.py:333-339): eval encode + reparameterize."""
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconVAE, node_embeddings
    gen = torch.Generator().manual_seed(4)
    ei, n, batch = path_batch([138])
    x = torch.randn(n, 60, generator=gen)
    torch.manual_seed(4)
    ref = OM.SkipconVAE(60, 138, 28)
    randomize_bn(ref, gen)
    ref.eval()
    eps = torch.randn(10, n, 28, generator=gen)
    with torch.no_grad():
        enc = ref.encode(x, ei, O.NoiseTape([]))
        z_ref = OM.reparameterize(enc[-2], enc[-1], 10, O.NoiseTape([eps]))
    mod = SkipconVAE(60, 138, 28)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(torch.device("cuda:0")).train()
    with use_noise(NoiseTape([eps])):
        z = node_embeddings(mod, x, ei)
    assert mod.training and tuple(z.shape) == (n, 28)
    close(z, z_ref, FWD_RTOL, what="embeddings")


def test_train_vae_loop_runs_and_learns():
    """train_vae / validate_vae / test_vae_rmsd drive a loader exactly like the reference's loops."""
    from hgvae_amd import Batch, Data
    from hgvae_amd.gvae_skipcon import VAE, test_vae_rmsd, train_vae, validate_vae
    import hgvae_amd
    gen = torch.Generator().manual_seed(2)
    graphs = []
    for i in range(8):
        ei, n, _ = path_batch([60 + i])
        graphs.append(Data(x=torch.randn(n, 12, generator=gen), edge_index=ei))
    loader = [Batch.from_data_list(graphs[i:i + 4]) for i in (0, 4)]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    hgvae_amd.manual_seed(0)
    model = VAE(12, 32, 4).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = [train_vae(model, loader, opt) for _ in range(15)]
    assert all(l == l for l in losses)
    assert losses[-1] < losses[0]
    v = validate_vae(model, loader)
    r = test_vae_rmsd(model, loader)
    assert v == v and r == r and r > 0


def _fresh_state(ref):
    state = {k: v.clone() for k, v in ref.state_dict().items()}
    for k in state:
        if k.endswith("running_mean"):
            state[k].zero_()
        elif k.endswith("running_var"):
            state[k].fill_(1.0)
        elif k.endswith("num_batches_tracked"):
            state[k].zero_()
    return state


def _compare_step(ref, mod, loss_ref, loss, skip_grad=()):
    close(loss, loss_ref, FWD_RTOL, what="loss")
    ref_params = dict(ref.named_parameters())
    for name, prm in mod.named_parameters():
        rg = ref_params[name].grad
        if rg is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        assert prm.grad is not None, name
        close(prm.grad, rg, BWD_RTOL, what=name)
    ref_bufs = dict(ref.named_buffers())
    for name, buf in mod.named_buffers():
        close(buf.float(), ref_bufs[name].float(), FWD_RTOL, what=name)


@pytest.mark.parametrize("sizes", [[138] * 3, [1, 2, 90, 1]])
def test_attention_train_step_matches_oracle(sizes):
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconAttentionVAE, train_step_loss
    # Seed: with seed 10 one pre-activation of a later level of the ragged batch lies within 3e-7 of zero, so its relu
    # mask - and with it a fixed 8e-3 of one bias gradient - is decided by the rounding of the first projection
    # (measured: additive noise of 3e-7 on that projection's output flips it for one noise seed in four, under every
    # projection arithmetic; the three arithmetics land on different sides).  Seeds 11-15 have no such element.
    seed = 11
    gen = torch.Generator().manual_seed(seed)
    ei, n, batch = path_batch(sizes)
    f_in, h, z, ns = 60, 256, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(seed)
    ref = OM.SkipconAttentionVAE(f_in, h, z, ns)
    randomize_bn(ref, gen)
    ref.train()
    tape = skipcon_tape(n, h, z, ns, 0.49, gen, gat_edges=ei.size(1) + n)
    state = _fresh_state(ref)
    if len(set(sizes)) > 1:
        # The ragged batch (94 rows, one- and two-node graphs) is ill-conditioned for the fp32 oracle itself: its encoder
        # gradients are up to 8e-3 (relative) away from a float64 run of the same restatement, the HIP path 3e-6.
        # Reference = the float64 run.
        ref = OM.SkipconAttentionVAE(f_in, h, z, ns).double()
        ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
        ref.train()
        tape64 = [t.double() if t.is_floating_point() else t for t in tape]
        loss_ref = OM.skipcon_loss(ref, x.double(), ei, batch, 1.0, O.NoiseTape(tape64))[0]
    else:
        loss_ref = OM.skipcon_loss(ref, x, ei, batch, 1.0, O.NoiseTape(tape))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    mod = SkipconAttentionVAE(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, x.to(dev), ei.to(dev), batch.to(dev), 1.0)
    loss.backward()
    _compare_step(ref, mod, loss_ref, loss)


def pool_tape(n, h, z, ns, gen, second_encode=True, second_decode=False):
    def m(*shape):
        return torch.bernoulli(torch.full(shape, 0.5), generator=gen)
    enc = lambda: [m(n, h), m(n, 2 * h)]  # noqa: E731
    dec = lambda: [torch.randn(ns, n, z, generator=gen), m(n, h)]  # noqa: E731
    items = enc() + dec()
    if second_decode:
        items += dec()
    if second_encode:
        items += enc()
    return items


@pytest.mark.parametrize("sizes", [[454, 200, 91], [1, 2, 77, 1, 3]])
def test_pool_vae_train_step_and_pool_outputs_match_oracle(sizes):
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import PoolVAE, train_step_loss
    gen = torch.Generator().manual_seed(20)
    ei, n, batch = path_batch(sizes)
    f_in, h, z, ns = 6, 138, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(20)
    ref = OM.PoolVAE(f_in, h, z, ns)
    randomize_bn(ref, gen)
    ref.train()
    tape = pool_tape(n, h, z, ns, gen)
    state = _fresh_state(ref)
    loss_ref = OM.pool_loss(ref, x, ei, batch, 1.0, O.NoiseTape(tape))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    mod = PoolVAE(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, x.to(dev), ei.to(dev), batch.to(dev), 1.0)
    loss.backward()
    _compare_step(ref, mod, loss_ref, loss)
    # the pool/unpool branch of forward (eval mode: no noise needed except eps)
    ref.eval(); mod.eval()
    eps = torch.randn(ns, n, z, generator=gen)
    with torch.no_grad():
        r_rec, r_xu, r_eiu = ref(x, ei, batch, O.NoiseTape([eps]))
        with use_noise(NoiseTape([eps])):
            o_rec, o_xu, o_eiu = mod(x.to(dev), ei.to(dev), batch.to(dev))
    close(o_rec, r_rec, FWD_RTOL, what="recon")
    assert torch.equal(o_eiu.cpu(), r_eiu)
    close(o_xu, r_xu, FWD_RTOL, what="x_unpooled")


@pytest.mark.parametrize("sizes", [[300, 154], [1, 61, 2, 1]])
def test_custom_pool_vae_train_step_matches_oracle(sizes):
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import CustomPoolVAE, custom_pool_step_loss
    gen = torch.Generator().manual_seed(30)
    ei, n, batch = path_batch(sizes)
    f_in, h, z, ns = 6, 138, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(30)
    ref = OM.CustomPoolVAE(f_in, h, z, ns, beta=0.5)
    randomize_bn(ref, gen)
    ref.train()
    tape = pool_tape(n, h, z, ns, gen, second_encode=False, second_decode=True)
    state = _fresh_state(ref)
    loss_ref = OM.custom_pool_loss(ref, x, ei, batch, O.NoiseTape(tape))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    mod = CustomPoolVAE(f_in, h, z, ns, beta=0.5)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = custom_pool_step_loss(mod, x.to(dev), ei.to(dev), batch.to(dev))
    loss.backward()
    _compare_step(ref, mod, loss_ref, loss)
    # coarse graph returned by forward
    ref.eval(); mod.eval()
    eps = torch.randn(ns, n, z, generator=gen)
    with torch.no_grad():
        r = ref(x, ei, batch, O.NoiseTape([eps]))
        with use_noise(NoiseTape([eps])):
            o = mod(x.to(dev), ei.to(dev), batch.to(dev))
    close(o[0], r[0], FWD_RTOL, what="recon")
    assert torch.equal(o[2].cpu(), r[2])
    close(o[1], r[1], FWD_RTOL, what="x_cg")


def test_graphed_train_step_replays_with_fresh_noise_and_learns():
    from hgvae_amd.graphed import GraphedTrainStep
    from hgvae_amd.vae import SkipconAttentionVAE, SkipconVAE
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    ei, n, batch = path_batch([138] * 8)
    x = torch.randn(n, 20, generator=gen).to(dev)
    eid, bd = ei.to(dev), batch.to(dev)
    for cls in (SkipconVAE, SkipconAttentionVAE):
        # lr = 0: parameters stay put, so any change of the loss between replays is fresh noise
        torch.manual_seed(0)
        model = cls(20, 32, 4).to(dev).train()
        frozen = GraphedTrainStep(model, torch.optim.Adam(model.parameters(), lr=0.0, capturable=True), x, eid, bd, seed=5)
        assert frozen.optimizer_in_graph
        before = [p.detach().clone() for p in model.parameters()]
        l1, l2 = float(frozen().item()), float(frozen().item())
        assert l1 == l1 and l2 == l2 and l1 != l2
        assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))
        # optimizer inside the graph, and stepped eagerly after the replay: both train
        for capturable in (True, False):
            torch.manual_seed(0)
            m2 = cls(20, 32, 4).to(dev).train()
            step = GraphedTrainStep(m2, torch.optim.Adam(m2.parameters(), lr=1e-2, capturable=capturable), x, eid, bd,
                                    seed=5)
            assert step.optimizer_in_graph == capturable
            trace = [float(step().item()) for _ in range(150)]
            assert all(l == l for l in trace)
            # medians: robust to the occasional exp(log_std) spike of an untrained VAE
            assert sorted(trace[-30:])[15] < sorted(trace[:30])[15]


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_training_with_torch_adam_tracks_flat_adam(fused):
    """Weights updated by torch.optim.Adam - including fused=True, which does not bump Tensor._version -
    must be seen by the next forward pass: five steps with torch's optimizer and five with FlatAdam from the
    same initial weights and noise streams give the same loss curve (packed projection weights are rebuilt
    at every use, never cached on a version counter)."""
    import hgvae_amd
    from hgvae_amd import synthetic
    from hgvae_amd.distributed import FlatGradBucket
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 8, 60, seed=5, device=dev)

    def run(kind):
        torch.manual_seed(0)
        hgvae_amd.manual_seed(99)
        model = SkipconVAE(60, 138, 28).to(dev).train()
        if kind == "flat":
            opt = FlatAdam(model.parameters(), lr=1e-2)
            bucket = opt.bucket
        else:
            opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=fused)
            bucket = FlatGradBucket(model.parameters())
        losses = []
        for _ in range(5):
            bucket.zero()
            loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
            loss.backward()
            bucket.all_reduce_mean()
            opt.step()
            losses.append(float(loss.detach()))
        return losses

    a, b = run("flat"), run("torch")
    assert a[-1] < a[0]                       # it trains
    for x, y in zip(a, b):
        assert abs(x - y) <= 2e-4 * abs(x), (a, b)


def test_direct_weight_grads_match_autograd_accumulation():
    """Gradients added by the reduction kernel straight into the bucket views equal the ones autograd
    accumulates from returned tensors (same kernels; only the final add differs in rounding order)."""
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd import synthetic
    from hgvae_amd.distributed import FlatGradBucket
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 6 + [77], 60, seed=11, device=dev)
    flats = []
    for direct in (False, True):
        torch.manual_seed(0)
        hgvae_amd.manual_seed(5)
        model = SkipconVAE(60, 138, 28).to(dev).train()
        bucket = FlatGradBucket(model.parameters())
        for _ in range(2):                      # second step: buffers already hold values that must be zeroed, not kept
            bucket.zero()
            loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
            with Fn.direct_weight_grads(direct):
                loss.backward()
            bucket.gather()
        flats.append(bucket.flat.clone())
    a, b = flats
    assert float(a.abs().max()) > 0
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())


@pytest.mark.parametrize("variant", ["skipcon", "attention"])
def test_paired_first_level_equals_two_separate_epilogues(variant, monkeypatch):
    """Inside a train step the first level's relu -> dropout -> batch norm of both encode passes runs as one paired
    operator over the shared convolution output.  With the operator replaced by two ordinary calls that draw the same
    Philox offsets, the loss, every gradient and every BatchNorm buffer must come out bit for bit the same."""
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd import synthetic
    from hgvae_amd.vae import SkipconAttentionVAE, SkipconVAE, train_step_loss
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 5 + [61], 60, seed=4, device=dev)

    def two_calls(y, bn, relu, p):
        ya, yb = Fn.fork(y)
        return Fn.bn_act(ya, bn, relu, p, True, fork=True), Fn.bn_act(yb, bn, relu, p, True, fork=True)

    results = []
    for paired in (True, False):
        if not paired:
            monkeypatch.setattr(Fn, "bn_act_pair", two_calls)
        torch.manual_seed(0)
        hgvae_amd.manual_seed(9)
        model = (SkipconVAE(60, 138, 28) if variant == "skipcon" else SkipconAttentionVAE(60, 64, 28)).to(dev).train()
        losses = []
        for _ in range(2):
            model.zero_grad()
            loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
            loss.backward()
            losses.append(loss.detach().clone())
        results.append((losses, [p.grad.clone() for p in model.parameters()], [b.clone() for b in model.buffers()]))
    (la, ga, ba), (lb, gb, bb) = results
    assert all(torch.equal(x, y) for x, y in zip(la, lb))
    assert all(torch.equal(x, y) for x, y in zip(ga, gb))
    assert all(torch.equal(x, y) for x, y in zip(ba, bb))
    assert int(ba[2]) == 4          # encoder1's BatchNorm saw two batches per step


def test_planned_weight_images_follow_every_parameter_change():
    """Inside a loss evaluation the projection weights' images are packed by one launch from a plan recorded in the
    previous evaluation.  The images must be those of the CURRENT weights however the weights were changed in between
    (optimizer step, in-place edit through .data, load_state_dict), and the first evaluation (no plan yet) must give
    the same numbers as later ones given the same weights and noise."""
    import hgvae_amd
    from hgvae_amd import functional as Fn
    from hgvae_amd import synthetic
    from hgvae_amd.vae import SkipconVAE, train_step_loss
    if Fn.default_projection() == "fp32":
        pytest.skip("HGVAE_PROJECTION=fp32: the fp32 kernels pack their images per use, there is no plan")
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 4, 60, seed=3, device=dev)

    def evaluate(model):
        hgvae_amd.manual_seed(21)
        model.zero_grad()
        loss = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        loss.backward()
        return loss.detach().clone(), [p.grad.clone() for p in model.parameters()]

    torch.manual_seed(0)
    model = SkipconVAE(60, 138, 28).to(dev).train()
    first = evaluate(model)                                   # records the plan
    assert getattr(model, "_hgv_image_plan_grad", None)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    # change the weights in ways that leave Tensor._version alone or replace contents wholesale
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(1.01)
    changed = evaluate(model)                                 # planned images, new weights
    torch.manual_seed(0)
    fresh = SkipconVAE(60, 138, 28).to(dev).train()
    fresh.load_state_dict(model.state_dict())                 # same weights, no plan
    want = evaluate(fresh)
    assert torch.equal(changed[0], want[0]) and all(torch.equal(a, b) for a, b in zip(changed[1], want[1]))
    assert not torch.equal(changed[0], first[0])
    model.load_state_dict(state)                              # back to the first weights (BatchNorm buffers included)
    fresh2 = SkipconVAE(60, 138, 28).to(dev).train()
    fresh2.load_state_dict(state)
    again, want2 = evaluate(model), evaluate(fresh2)
    assert torch.equal(again[0], want2[0]) and all(torch.equal(a, b) for a, b in zip(again[1], want2[1]))
    # evaluation mode keeps a plan of its own (no transposed images)
    model.eval()
    with torch.no_grad():
        train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
        a = train_step_loss(model, batch.x, batch.edge_index, batch.batch, 1.0)
    with Fn.direct_weight_grads(False):
        fresh2.eval()
        with torch.no_grad():
            b = train_step_loss(fresh2, batch.x, batch.edge_index, batch.batch, 1.0)
    assert len(model._hgv_image_plan_nograd) < len(model._hgv_image_plan_grad)
    assert torch.isfinite(a) and torch.isfinite(b)


@pytest.mark.parametrize("variant", ["skipcon", "attention"])
def test_captured_step_with_the_second_pass_on_a_branch_stream(variant):
    """Under capture the second encode pass is a parallel branch of the graph.  Same noise, same kernels, same
    per-stream workspaces, running statistics updated in program order: losses, parameters and BatchNorm buffers after
    three replays equal those of a capture without the branch bit for bit."""
    import hgvae_amd
    from hgvae_amd import synthetic, vae
    from hgvae_amd.graphed import GraphedTrainStep
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import SkipconAttentionVAE, SkipconVAE
    dev = torch.device("cuda:0")
    batch = synthetic.path_graph_batch([138] * 8, 60, seed=5, device=dev)
    finals = []
    try:
        for mode in ("never", "capture"):
            vae.set_branch_streams(mode)
            torch.manual_seed(0)
            model = (SkipconVAE(60, 138, 28) if variant == "skipcon" else SkipconAttentionVAE(60, 64, 28)).to(dev).train()
            opt = FlatAdam(model.parameters(), lr=1e-3)
            step = GraphedTrainStep(model, opt, batch.x, batch.edge_index, batch.batch, bucket=opt.bucket, seed=31)
            losses = [float(step()) for _ in range(3)]
            torch.cuda.synchronize()
            finals.append((losses, torch.cat([p.detach().flatten() for p in model.parameters()]).clone(),
                           [b.clone() for b in model.buffers()]))
    finally:
        vae.set_branch_streams("capture")
    (la, pa, ba), (lb, pb, bb) = finals
    # the same kernels on the same inputs; gradient sums and running-statistics updates in the same order
    assert la == lb, (la, lb)
    assert torch.equal(pa, pb)
    for x, y in zip(ba, bb):
        assert torch.equal(x, y)
