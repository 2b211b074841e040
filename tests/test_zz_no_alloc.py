"""hgv_bn_act_fwd_static (the entry point that allocates nothing) against hgv_bn_act_fwd, through the C ABI."""
import pytest
import torch

from hgvae_amd import _lib
from hgvae_amd.graph import new_padded

pytestmark = pytest.mark.gpu


def _run(name, y, mask, p, relu, gamma, beta, seed, offset):
    """One forward through entry point ``name``; returns (out, save_mean, save_rstd, running_mean, running_var, tracked)."""
    n, h = y.shape
    dev = y.device
    out = new_padded(n, h, dev)
    save_mean = torch.empty(h, device=dev)
    save_rstd = torch.empty(h, device=dev)
    rm, rv = torch.zeros(h, device=dev), torch.ones(h, device=dev)
    tracked = torch.zeros(1, dtype=torch.int64, device=dev)
    args = [y.data_ptr(), y.stride(0), n, h, int(relu), float(p), _lib.ptr(mask), seed, offset, None, gamma.data_ptr(),
            beta.data_ptr(), 1e-5, rm.data_ptr(), rv.data_ptr(), 0.1, tracked.data_ptr(), save_mean.data_ptr(),
            save_rstd.data_ptr(), out.data_ptr(), out.stride(0)]
    if name == "hgv_bn_act_fwd":
        ws = torch.empty(_lib.load().hgv_bn_act_ws(n, h), dtype=torch.uint8, device=dev)
        args += [ws.data_ptr(), ws.numel()]
    _lib.call(name, *args, _lib.stream())
    torch.cuda.synchronize()
    return out, save_mean, save_rstd, rm, rv, tracked


def _case(n, h, p, relu, explicit_mask, seed):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(seed)
    y = new_padded(n, h, dev)
    y.copy_(torch.randn(n, h, generator=gen, device=dev) * 2 + 0.3)
    gamma = torch.rand(h, generator=gen, device=dev) + 0.5
    beta = torch.randn(h, generator=gen, device=dev)
    mask = (torch.rand(n, h, generator=gen, device=dev) >= p).to(torch.uint8).contiguous() if explicit_mask else None
    ref = _run("hgv_bn_act_fwd", y, mask, p, relu, gamma, beta, 77, 1 << 20)
    got = _run("hgv_bn_act_fwd_static", y, mask, p, relu, gamma, beta, 77, 1 << 20)
    for a, b, what in zip(got, ref, ("out", "save_mean", "save_rstd", "running_mean", "running_var", "tracked")):
        assert torch.equal(a, b), f"{what} differs at N={n} H={h}"
    # independent check of the statistics against torch (the two entry points share kernels)
    d = torch.relu(y) if relu else y.clone()
    if mask is not None:
        d = d * mask / (1 - p)
        assert torch.allclose(got[1], d.mean(0), rtol=1e-5, atol=1e-6)
    assert int(got[5]) == 1


def test_static_entry_matches_workspace_entry_two_calls_in_a_row():
    # the second call is smaller: it must not be disturbed by what the first left in the library's scratch array
    _case(50_000, 138, 0.5, True, True, 1)
    _case(777, 56, 0.5, True, True, 2)
    _case(3, 5, 0.0, False, False, 3)
    _case(20_011, 276, 0.49, True, False, 4)      # Philox masks, H above one 256-thread row slot


def test_static_entry_rejects_widths_beyond_its_scratch():
    dev = torch.device("cuda:0")
    y = new_padded(8, 292, dev)
    y.zero_()
    g = torch.ones(292, device=dev)
    with pytest.raises(_lib.HgvError, match="exceeds the built-in scratch"):
        _run("hgv_bn_act_fwd_static", y, None, 0.0, False, g, g, 0, 0)
