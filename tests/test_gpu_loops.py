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
