"""Values of the reference's evaluation loops, the stated config-1 batch, and regressions of host-side state handling.

Reference functions: validate_vae / compute_rmsd / test_vae_rmsd (gvae/gvae_skipcon.py:190-233) and the train step on
"one bundled small PDB, batch 32" (chignolin, 138 atoms, 60 features).
"""
import pytest
import torch

from graphs import path_batch
from oracle import models as OM
from oracle import pyg_ops as O
from tapes import chig_batch, skipcon_tape
from test_gpu_model import BWD_RTOL, FWD_RTOL, _compare_step, _fresh_state, close, randomize_bn

pytestmark = pytest.mark.gpu


def _loader(gen, f_in=12, sizes=((40, 57), (33, 80, 21))):
    from hgvae_amd import Batch, Data
    out = []
    for group in sizes:
        items = []
        for n in group:
            ei, _, _ = path_batch([n])
            items.append(Data(x=torch.randn(n, f_in, generator=gen), edge_index=ei))
        out.append(Batch.from_data_list(items))
    return out


def _eval_model_pair(gen, f_in=12, h=32, z=4):
    """(oracle, HIP model) with the same weights and non-trivial BatchNorm buffers, both in eval mode."""
    from hgvae_amd.gvae_skipcon import VAE
    torch.manual_seed(5)
    ref = OM.SkipconVAE(f_in, h, z)
    randomize_bn(ref, gen)
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    mod = VAE(f_in, h, z)
    mod.load_state_dict(ref.state_dict())
    return ref.eval(), mod.to(torch.device("cuda:0")).eval()


def test_validate_vae_value_matches_oracle():
    """validate_vae = mean over batches of (mse + kl) from the eval-mode double pass (gvae_skipcon.py:190-204)."""
    from hgvae_amd.gvae_skipcon import validate_vae
    from hgvae_amd.noise import NoiseTape, use_noise
    gen = torch.Generator().manual_seed(7)
    loader = _loader(gen)
    ref, mod = _eval_model_pair(gen)
    eps = [torch.randn(10, b.x.size(0), 4, generator=gen) for b in loader]
    want = 0.0
    with torch.no_grad():
        for b, e in zip(loader, eps):
            loss, _, _ = OM.skipcon_loss(ref, b.x, b.edge_index, b.batch, 1.0, O.NoiseTape([e]))
            want += loss.item()
    want /= len(loader)
    with use_noise(NoiseTape(eps)):
        got = validate_vae(mod, loader)
    assert abs(got - want) <= FWD_RTOL * abs(want) + 1e-6, (got, want)
    assert not mod.training


def test_test_vae_rmsd_value_matches_oracle():
    """test_vae_rmsd = mean over batches of compute_rmsd(x[:, :3], recon[:, :3]) (gvae_skipcon.py:207-233)."""
    from hgvae_amd.gvae_skipcon import compute_rmsd, test_vae_rmsd
    from hgvae_amd.noise import NoiseTape, use_noise
    gen = torch.Generator().manual_seed(8)
    loader = _loader(gen)
    ref, mod = _eval_model_pair(gen)
    eps = [torch.randn(10, b.x.size(0), 4, generator=gen) for b in loader]
    want = 0.0
    with torch.no_grad():
        for b, e in zip(loader, eps):
            recon = ref(b.x, b.edge_index, b.batch, O.NoiseTape([e]))[0]
            want += OM.compute_rmsd(b.x[:, :3], recon[:, :3]).item()
    want /= len(loader)
    with use_noise(NoiseTape(eps)):
        got = test_vae_rmsd(mod, loader)
    assert abs(got - want) <= FWD_RTOL * abs(want) + 1e-6, (got, want)
    a, b_ = torch.randn(9, 3, generator=gen), torch.randn(9, 3, generator=gen)
    close(compute_rmsd(a.cuda(), b_.cuda()), OM.compute_rmsd(a, b_), 1e-6, what="compute_rmsd")
    with pytest.raises(AssertionError):
        compute_rmsd(a.cuda(), b_[:4].cuda())


def test_chignolin_batch32_train_step_matches_oracle():
    """The stated configuration 1: 32 chignolin graphs (N = 4416, E = 4384, 60 features), VAE(60, 138, 28)."""
    from hgvae_amd.gvae_skipcon import VAE
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import train_step_loss
    b = chig_batch(32)
    n = b.x.size(0)
    assert (n, b.edge_index.size(1), b.x.size(1)) == (4416, 4384, 60)
    gen = torch.Generator().manual_seed(11)
    torch.manual_seed(0)
    ref = OM.SkipconVAE(60, 138, 28, 10)
    randomize_bn(ref, gen)
    ref.train()
    state = _fresh_state(ref)
    tape = skipcon_tape(n, 138, 28, 10, 0.5, gen)
    loss_ref = OM.skipcon_loss(ref, b.x, b.edge_index, b.batch, 1.0, O.NoiseTape(tape))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    mod = VAE(60, 138, 28, 10)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, b.x.to(dev), b.edge_index.to(dev), b.batch.to(dev), 1.0)
    loss.backward()
    _compare_step(ref, mod, loss_ref, loss)


def test_first_conv_sharing_never_outlives_a_loss_evaluation():
    """An odd number of encode passes (test_vae_rmsd: one forward per batch) followed by a weight update that does
    not bump Tensor._version (fused Adam) must not leak the first convolution's output into later calls: evaluation
    after the update equals a freshly built model with the same weights, bit for bit."""
    from hgvae_amd.gvae_skipcon import VAE, test_vae_rmsd, train_vae, validate_vae
    from hgvae_amd.noise import NoiseTape, use_noise
    import hgvae_amd
    gen = torch.Generator().manual_seed(12)
    loader = _loader(gen)
    dev = torch.device("cuda:0")
    loader = [b.to(dev) for b in loader]
    torch.manual_seed(1)
    hgvae_amd.manual_seed(1)
    model = VAE(12, 32, 4).to(dev)
    test_vae_rmsd(model, loader)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, fused=True)
    train_vae(model, loader, opt)
    eps = [torch.randn(10, b.x.size(0), 4, generator=gen) for b in loader]
    with use_noise(NoiseTape(eps)):
        got = validate_vae(model, loader)          # same Batch objects as before the update
    fresh = VAE(12, 32, 4)
    fresh.load_state_dict({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    fresh = fresh.to(dev)
    with use_noise(NoiseTape(eps)):
        want = validate_vae(fresh, loader)
    assert got == want, (got, want)
    # forward-only call, weight change through .data, forward again: the second call sees the new weights
    model.eval()
    b = loader[0]
    with torch.no_grad(), use_noise(NoiseTape([eps[0]])):
        r0 = model(b.x, b.edge_index, b.batch)[0].clone()
        model.encoder1[0].lin_l.weight.data.mul_(1.5)
    with torch.no_grad(), use_noise(NoiseTape([eps[0]])):
        r1 = model(b.x, b.edge_index, b.batch)[0]
    assert not torch.equal(r0, r1)
    assert model._conv1_memo is None


def test_attention_model_at_a_width_that_is_not_a_multiple_of_four():
    """SkipconAttentionVAE(hidden=138): GATConv hands BatchNorm a dense [N, 138] tensor (rows not 16-byte aligned), the
    path on which the epilogue kernels run their scalar variant."""
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconAttentionVAE, train_step_loss
    gen = torch.Generator().manual_seed(13)
    ei, n, batch = path_batch([138, 77, 41])
    f_in, h, z, ns = 60, 138, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(13)
    ref = OM.SkipconAttentionVAE(f_in, h, z, ns)
    randomize_bn(ref, gen)
    ref.train()
    tape = skipcon_tape(n, h, z, ns, 0.49, gen, gat_edges=ei.size(1) + n)
    state = _fresh_state(ref)
    loss_ref = OM.skipcon_loss(ref, x, ei, batch, 1.0, O.NoiseTape(tape))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    junk = torch.full((64 << 20,), float("nan"), device=dev)     # recycled allocator blocks hold NaNs, not zeros
    del junk
    mod = SkipconAttentionVAE(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    with use_noise(NoiseTape(tape)):
        loss = train_step_loss(mod, x.to(dev), ei.to(dev), batch.to(dev), 1.0)
    loss.backward()
    _compare_step(ref, mod, loss_ref, loss)


@pytest.mark.parametrize("aligned", [False, True])
def test_bn_act_keeps_pad_columns_zero(aligned):
    """Outputs of the fused relu/dropout/BatchNorm (and its input gradient) are consumed with 16-byte loads that also
    read the pad columns [H, ld): they must be zero whatever the recycled memory held, on the scalar path too."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graph import new_padded
    from hgvae_amd.noise import NoiseTape, use_noise
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(14)
    n, h = 3001, 138
    y0, g0 = torch.randn(n, h, generator=gen), torch.randn(n, h, generator=gen)
    mask = torch.bernoulli(torch.full((n, h), 0.5), generator=gen)
    bn = torch.nn.BatchNorm1d(h).to(dev)

    def pads(t):
        ld = t.stride(0)
        assert ld == 140
        return torch.as_strided(t, (t.size(0), ld - h), (ld, 1), t.storage_offset() + h)

    seen = {}

    class Tap(torch.autograd.Function):          # records the gradient tensor exactly as the operator returned it
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, grad):
            seen["gy"] = grad
            return grad

    for _ in range(3):
        junk = torch.full((n * 140 * 8,), float("nan"), device=dev)
        del junk
        if aligned:
            y = new_padded(n, h, dev)
            y.copy_(y0.to(dev))
            g = new_padded(n, h, dev)
            g.copy_(g0.to(dev))
        else:
            y, g = y0.to(dev), g0.to(dev)          # dense rows of 552 bytes
        y.requires_grad_(True)
        with use_noise(NoiseTape([mask])):
            out = Fn.bn_act(Tap.apply(y), bn, True, 0.5, True)
        assert float(pads(out.detach()).abs().max()) == 0.0
        torch.autograd.grad(out, y, g)
        gy = seen.pop("gy")
        assert float(pads(gy).abs().max()) == 0.0
        assert torch.isfinite(gy).all()
        with torch.no_grad():
            ev = Fn.bn_act(y.detach(), bn, True, 0.5, False)
        assert float(pads(ev).abs().max()) == 0.0


GRAPHS2_SIZES = [454, 2627, 4697, 11213, 11455, 11940, 13239, 15200]   # the eight graph sizes of the bundled graphs_2 set


@pytest.mark.parametrize("variant", ["pool", "custom_pool"])
def test_pool_variants_at_the_stated_batch_match_float64_oracle(variant):
    """Configuration 3 as stated: one batch of 8 path graphs of the bundled sizes (70 825 nodes, 6 features),
    VAE(6, 138, 28) of vae_with_pool / vae_custom_pool.  Reference = the oracle in float64 (at this size the fp32 CPU
    oracle is itself ~1e-3 away from it); the pooled graph is checked against the oracle's top-k on the device scores."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import CustomPoolVAE, PoolVAE, custom_pool_step_loss, train_step_loss
    from test_gpu_model import pool_tape
    gen = torch.Generator().manual_seed(40)
    ei, n, batch = path_batch(GRAPHS2_SIZES)
    assert n == 70825
    f_in, h, z, ns = 6, 138, 28, 10
    x = torch.randn(n, f_in, generator=gen)
    torch.manual_seed(40)
    custom = variant == "custom_pool"
    ref32 = (OM.CustomPoolVAE if custom else OM.PoolVAE)(f_in, h, z, ns)
    randomize_bn(ref32, gen)
    state = _fresh_state(ref32)
    tape = pool_tape(n, h, z, ns, gen, second_encode=not custom, second_decode=custom)
    ref = (OM.CustomPoolVAE if custom else OM.PoolVAE)(f_in, h, z, ns).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
    ref.train()
    tape64 = [t.double() for t in tape]
    if custom:
        loss_ref = OM.custom_pool_loss(ref, x.double(), ei, batch, O.NoiseTape(tape64))[0]
    else:
        loss_ref = OM.pool_loss(ref, x.double(), ei, batch, 1.0, O.NoiseTape(tape64))[0]
    loss_ref.backward()
    dev = torch.device("cuda:0")
    mod = (CustomPoolVAE if custom else PoolVAE)(f_in, h, z, ns)
    mod.load_state_dict(state)
    mod = mod.to(dev).train()
    xd, eid, bd = x.to(dev), ei.to(dev), batch.to(dev)
    with use_noise(NoiseTape(tape)):
        loss = custom_pool_step_loss(mod, xd, eid, bd) if custom else train_step_loss(mod, xd, eid, bd, 1.0)
    loss.backward()
    close(loss, loss_ref, 2e-6, what="loss")
    rp = dict(ref.named_parameters())
    for name, prm in mod.named_parameters():
        if rp[name].grad is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
        else:
            # atol: bias gradients in front of a BatchNorm are sums over 70 k rows that cancel to ~1e-4 of the summed
            # magnitudes, so fp32 accumulation leaves an absolute error of a few 1e-6 whatever the kernel
            close(prm.grad, rp[name].grad, BWD_RTOL, atol=5e-6, what=name)
    # the pooling branch at this size: selection = the oracle's top-k of the device scores; coarse edges = filter_adj of it
    mod.eval()
    with torch.no_grad():
        x1 = mod.encode(xd, eid)[0]
        pool = mod.pool_unpool_graph.pool1
        x_cg, ei_cg, _, b_cg, perm, sc = pool(x1, eid, batch=bd)
        score = torch.tanh(pool.gnn(x1, eid).reshape(-1)).cpu()
    want_perm = O.topk(score, 0.5, batch)
    assert torch.equal(perm.cpu(), want_perm)
    assert perm.numel() == sum((s + 1) // 2 for s in GRAPHS2_SIZES)
    want_ei, _ = O.filter_adj(ei, None, want_perm, n)
    assert torch.equal(ei_cg.cpu(), want_ei) and torch.equal(b_cg.cpu(), batch[want_perm])
    close(x_cg, x1.cpu()[want_perm] * score[want_perm].view(-1, 1), 1e-6, what="x_cg")
    with torch.no_grad(), Fn.bounded_pooling():
        _, ei_b, _, _, _, _ = pool(x1, eid, batch=bd)
    assert ei_b.shape == (2, ei.size(1)) and int(ei_b.hgv_count) == want_ei.size(1)
    assert torch.equal(ei_b[:, :want_ei.size(1)].cpu(), want_ei)


@pytest.mark.parametrize("variant", ["pool", "custom_pool"])
def test_pool_variants_replay_as_a_captured_step(variant):
    """The pooling variants make no host round trip inside their loss (bounded pooling, cached batch plans), so the
    whole train step can be captured and replayed; replays draw fresh noise and train."""
    from hgvae_amd.graphed import GraphedTrainStep
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import CustomPoolVAE, PoolVAE, custom_pool_step_loss, train_step_loss
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(50)
    ei, n, batch = path_batch([454, 300, 77, 1, 2])
    x = torch.randn(n, 6, generator=gen).to(dev)
    eid, bd = ei.to(dev), batch.to(dev)
    torch.manual_seed(50)
    if variant == "pool":
        model, loss_fn = PoolVAE(6, 32, 4).to(dev).train(), train_step_loss
    else:
        model = CustomPoolVAE(6, 32, 4).to(dev).train()
        loss_fn = lambda m, a, b, c, _lam: custom_pool_step_loss(m, a, b, c)  # noqa: E731
    opt = FlatAdam(model.parameters(), lr=1e-2, capturable=True)
    w0 = model.encoder1[0].lin_l.weight.detach().clone()
    step = GraphedTrainStep(model, opt, x, eid, bd, 1.0, loss_fn=loss_fn, seed=3)
    assert torch.equal(model.encoder1[0].lin_l.weight.detach(), w0)      # building the graph does not train
    losses = [float(step()) for _ in range(30)]
    assert all(v == v for v in losses) and len(set(losses)) > 1
    assert sum(losses[-5:]) < sum(losses[:5])
    # the reference train step of vae_with_pool encodes twice per step, the custom-pool one once
    assert int(model.encoder1[2].num_batches_tracked) == (60 if variant == "pool" else 30)


def test_evaluation_mode_stays_differentiable():
    """model.eval() only freezes dropout and the batch statistics; gradients still flow (fine-tuning with frozen statistics,
    input gradients of a trained model) exactly as through torch's modules."""
    import torch
    from oracle import models as OM
    from oracle import pyg_ops as O
    from hgvae_amd.noise import NoiseTape, use_noise
    from hgvae_amd.vae import SkipconVAE
    from graphs import path_batch
    gen = torch.Generator().manual_seed(3)
    ei, n, batch = path_batch([40, 77])
    x = torch.randn(n, 60, generator=gen)
    torch.manual_seed(3)
    ref = OM.SkipconVAE(60, 138, 28, 3).double()
    with torch.no_grad():
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen).double() * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen).double() + 0.5)
    ref.eval()
    eps = torch.randn(3, n, 28, generator=gen)
    xr = x.double().requires_grad_(True)
    rec_ref = ref(xr, ei, batch, O.NoiseTape([eps.double()]))[0]
    rec_ref.square().mean().backward()
    dev = torch.device("cuda:0")
    mod = SkipconVAE(60, 138, 28, 3)
    mod.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
    mod = mod.to(dev).eval()
    before = {k: v.clone() for k, v in mod.named_buffers()}
    xd = x.to(dev).requires_grad_(True)
    with use_noise(NoiseTape([eps])):
        rec = mod(xd, ei.to(dev), batch.to(dev))[0]
    rec.square().mean().backward()
    assert xd.grad is not None and float(xd.grad.abs().max()) > 0
    scale = float(xr.grad.abs().max())
    assert float((xd.grad.double().cpu() - xr.grad).abs().max()) <= 1e-4 * scale
    rp = dict(ref.named_parameters())
    for name, p in mod.named_parameters():
        assert p.grad is not None, name
        r = rp[name].grad
        assert float((p.grad.double().cpu() - r).abs().max()) <= 1e-4 * max(float(r.abs().max()), 1e-6), name
    for name, b in mod.named_buffers():                      # evaluation leaves the statistics alone
        assert torch.equal(b, before[name]), name


def _frames(n_batches, graphs, nodes, f, seed):
    """Batches of `graphs` path graphs of `nodes` nodes each: one topology, new features per batch (MD frames)."""
    from hgvae_amd.data import Batch
    gen = torch.Generator().manual_seed(seed)
    ei, n, bvec = path_batch([nodes] * graphs)
    dev = torch.device("cuda:0")
    eid, bd = ei.to(dev), bvec.to(dev)
    return [Batch(x=torch.randn(n, f, generator=gen).to(dev), edge_index=eid.clone(), batch=bd.clone()) for _ in range(n_batches)]


def test_captured_training_replays_one_capture_on_new_features():
    """Batches that share (edge_index, batch) are served by ONE captured step whose input is overwritten: with a frozen
    optimizer, the loss of batch k through CapturedTraining is bit-identical to a step captured directly on batch k and
    replayed from the same noise counter."""
    from hgvae_amd.graphed import CapturedTraining, GraphedTrainStep
    from hgvae_amd.graph import new_padded
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    batches = _frames(4, 6, 23, 10, seed=60)          # 10 features: the captured input lives in the padded layout (ld 12)
    torch.manual_seed(60)
    model = SkipconVAE(10, 32, 4).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=0.0, capturable=True)
    cap = CapturedTraining(model, opt, seed=11)
    cap.step(batches[0])
    assert len(cap._entries) == 1 and cap.captured_steps == 1
    entry = cap._entries[0][3]
    for data in batches[1:]:
        counter = entry.counter.clone()
        got = cap.step(data).clone()
        assert len(cap._entries) == 1                   # recognised by comparing edge_index / batch on the device
        x = new_padded(data.x.size(0), data.x.size(1), dev)
        x.copy_(data.x)
        ref = GraphedTrainStep(model, opt, x, data.edge_index, data.batch, seed=11)
        ref.counter.copy_(counter)
        assert torch.equal(ref(), got)
    assert cap.captured_steps == 4 and cap.eager_steps == 0


def test_captured_training_epochs_learn_and_fall_back():
    """Epochs over a device-resident data set of equal graphs: the shard tags its batches with a structure id, the full
    batches and the smaller last one get one capture each, the loss falls; a data set of unequal graphs runs eagerly
    once the capture budget is used up, with the same optimizer."""
    from hgvae_amd.data import Data
    from hgvae_amd.graphed import CapturedTraining
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.shard import GraphShard
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(61)
    ei, _, _ = path_batch([30])
    base = torch.randn(30, 12, generator=gen)
    frames = [Data(x=base + 0.1 * torch.randn(30, 12, generator=gen), edge_index=ei.clone()) for _ in range(22)]
    shard = GraphShard.from_data_list(frames, dev)
    assert shard.uniform
    loader = shard.loader(8)                              # 8 + 8 + 6 graphs
    assert next(iter(loader)).structure_id == ("uniform", shard._token(), 8)
    torch.manual_seed(61)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-2, capturable=True)
    cap = CapturedTraining(model, opt, seed=5)
    losses = [cap.train_epoch(loader) for _ in range(25)]
    assert all(v == v for v in losses)
    assert len(cap._entries) == 2 and cap.eager_steps == 0 and cap.captured_steps == 75
    assert sorted(losses[-5:])[2] < sorted(losses[:5])[2]
    assert int(model.encoder1[2].num_batches_tracked) == 150        # two encode passes per step
    # unequal graphs: no structure id, every batch differs; two captures allowed, the rest eager
    ragged = [Data(x=torch.randn(n, 12, generator=gen), edge_index=path_batch([n])[0]) for n in (5, 9, 14, 20, 27, 33)]
    rshard = GraphShard.from_data_list(ragged, dev)
    assert not rshard.uniform
    assert not hasattr(next(iter(rshard.loader(2))), "structure_id")
    cap2 = CapturedTraining(model, opt, seed=6, max_structures=2)
    w0 = model.encoder1[0].lin_l.weight.detach().clone()
    v = cap2.train_epoch(rshard.loader(2))
    assert v == v and cap2.captured_steps == 2 and cap2.eager_steps == 1
    assert not torch.equal(model.encoder1[0].lin_l.weight.detach(), w0)


def test_captured_validation_matches_validate_vae():
    """CapturedTraining.validate_epoch replays a captured evaluation step per batch structure.  In eval mode the only noise
    left is eps of the reparameterisation, which enters the loss through the decoder: with the same trained weights the
    captured epoch must land where the eager validate_vae does (different eps streams: compared within the spread of two
    eager runs), replay after replay, and follow the weights when they change."""
    from hgvae_amd.graphed import CapturedTraining
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import SkipconVAE, validate_vae
    dev = torch.device("cuda:0")
    batches = _frames(6, 8, 30, 12, seed=70)
    torch.manual_seed(70)
    model = SkipconVAE(12, 32, 4, 10).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-2, capturable=True)
    cap = CapturedTraining(model, opt, seed=9)
    for _ in range(20):
        cap.train_epoch(batches)
    eager = [validate_vae(model, batches) for _ in range(4)]
    captured = [cap.validate_epoch(batches) for _ in range(4)]
    assert len(cap._eval_entries) == 1 and not model.training
    spread = max(eager) - min(eager)
    mid = sum(eager) / len(eager)
    assert len(set(captured)) > 1                                     # fresh eps per replay
    assert all(abs(v - mid) <= 3 * spread + 1e-3 * abs(mid) for v in captured), (eager, captured)
    # the captured step reads the weights at replay time: more training moves the validation loss
    model.train()
    for _ in range(40):
        cap.train_epoch(batches)
    after = cap.validate_epoch(batches)
    assert abs(after - validate_vae(model, batches)) <= 3 * spread + 1e-3 * abs(after) + 0.05 * abs(after)
    assert after < mid


def test_second_capture_keeps_the_optimizer_state():
    """A new batch structure is captured in the middle of training (the smaller last batch of an epoch): the throw-away
    steps of that capture must leave Adam's moments and step count as they were, not reset them."""
    from hgvae_amd.graphed import CapturedTraining
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    full, last = _frames(3, 6, 23, 12, seed=80), _frames(1, 4, 23, 12, seed=81)
    torch.manual_seed(80)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)
    cap = CapturedTraining(model, opt, seed=3)
    for data in full:
        cap.step(data)
    w = model.encoder1[0].lin_l.weight
    st = opt.state[w]
    assert float(st["step"]) == 3.0
    before = {k: v.clone() for k, v in st.items()}
    params_before = [p.detach().clone() for p in model.parameters()]
    from hgvae_amd.graphed import GraphedTrainStep
    x = last[0].x
    GraphedTrainStep(model, opt, x, last[0].edge_index, last[0].batch, bucket=cap._bucket, seed=9)     # capture only
    for k, v in opt.state[w].items():
        assert torch.equal(v, before[k]), k
    for p, q in zip(model.parameters(), params_before):
        assert torch.equal(p.detach(), q)
    cap.step(last[0])                                  # second structure through the epoch driver: one real step
    assert len(cap._entries) == 2 and float(opt.state[w]["step"]) == 4.0
    assert float(opt.state[w]["exp_avg"].abs().max()) > 0


def test_eager_fallback_keeps_gradients_in_the_bucket():
    """An eager step between captured ones, with an optimizer that is stepped outside the graph (plain torch Adam): the
    fallback must not detach p.grad from the bucket's views, or the next replay's gradients would be overwritten by the
    stale eager ones when the bucket is gathered."""
    from hgvae_amd.graphed import CapturedTraining
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    a, b = _frames(2, 5, 19, 12, seed=82), _frames(1, 3, 31, 12, seed=83)
    torch.manual_seed(82)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)          # not capturable: stepped after the replay
    cap = CapturedTraining(model, opt, seed=4, max_structures=1)
    cap.step(a[0])
    cap.step(b[0])                                      # second structure, budget used up: eager
    assert cap.eager_steps == 1
    bucket = cap._bucket
    for p, v in zip(bucket.params, bucket.views):
        assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
    stale = bucket.flat.clone()
    entry = cap._entries[0][3]
    entry.x.copy_(a[1].x)
    entry.graph.replay()                                # gradients of the new batch, written into the bucket
    fresh = bucket.flat.clone()
    assert not torch.equal(fresh, stale)
    bucket.all_reduce_mean()                            # what __call__ does next: must leave the fresh gradients alone
    assert torch.equal(bucket.flat, fresh)
    for p, v in zip(bucket.params, bucket.views):
        assert p.grad.data_ptr() == v.data_ptr()


def test_branch_membership_ends_with_the_captured_step():
    """A captured step runs its second encode pass as a branch; afterwards an ordinary training-mode forward on any
    stream updates the BatchNorm running statistics itself (nothing is left parked in stand-ins)."""
    from hgvae_amd import functional as Fn
    from hgvae_amd.graphed import GraphedTrainStep
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    data = _frames(1, 6, 23, 12, seed=84)[0]
    torch.manual_seed(84)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=1e-3, capturable=True)
    step = GraphedTrainStep(model, opt, data.x, data.edge_index, data.batch, seed=2)
    step()
    torch.cuda.synchronize()
    assert Fn._trunk_stream is None and not Fn.on_branch_stream()
    bn = model.encoder2[2]
    rm, count = bn.running_mean.clone(), int(bn.num_batches_tracked)
    model(data.x, data.edge_index, data.batch)          # eager, default stream, train mode
    torch.cuda.synchronize()
    assert int(bn.num_batches_tracked) == count + 1
    assert not torch.equal(bn.running_mean, rm)
    assert not Fn._branch_updates


def test_captured_validation_is_keyed_by_the_kl_weight():
    from hgvae_amd.graphed import CapturedTraining
    from hgvae_amd.optim import FlatAdam
    from hgvae_amd.vae import SkipconVAE
    dev = torch.device("cuda:0")
    batches = _frames(2, 4, 17, 12, seed=85)
    torch.manual_seed(85)
    model = SkipconVAE(12, 32, 4).to(dev).train()
    cap = CapturedTraining(model, FlatAdam(model.parameters(), lr=1e-3, capturable=True), seed=1)
    one = cap.validate_epoch(batches, lambda_kl=1.0)
    zero = cap.validate_epoch(batches, lambda_kl=0.0)
    assert len(cap._eval_entries) == 2
    assert zero < one                                   # the KL term is positive and was dropped
