"""Dense-Laplacian spectral terms (autoencoder/vae_laplacian.py:24-52,170-191; vae_custom_pool.py:50-53) on the device
against the CPU restatement and against answers known in closed form."""
import math

import pytest
import torch

from oracle import spectral as OS

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _graphs():
    gen = torch.Generator().manual_seed(5)
    n = 300
    sym = torch.randint(0, n, (2, 900), generator=gen)
    sym = torch.cat([sym, sym.flip(0)], dim=1)                      # symmetric with repeats and self loops
    path = torch.stack([torch.arange(0, 453), torch.arange(1, 454)])  # the topology of the bundled graphs (directed path)
    yield "symmetric", sym, n
    yield "directed path", path, 454
    yield "isolated nodes", torch.tensor([[0, 1, 1, 5], [1, 0, 5, 1]]), 9
    yield "no edges", torch.zeros(2, 0, dtype=torch.int64), 4


@pytest.mark.parametrize("name,ei,n", list(_graphs()), ids=lambda v: v if isinstance(v, str) else None)
def test_normalized_laplacian_equals_the_reference_expression(name, ei, n):
    from hgvae_amd import spectral
    want = OS.normalized_laplacian(ei, n)
    got = spectral.normalized_laplacian(ei.to(DEV), n).cpu()
    # the only step not formed by the kernel is deg^-1/2 (torch on either device); everything after it is exact
    assert torch.allclose(got, want, rtol=0, atol=1.2e-7)
    strided = ei.t().contiguous().to(DEV).t()                        # [2, E] view of an [E, 2] buffer
    assert torch.equal(spectral.normalized_laplacian(strided, n).cpu(), got)


def test_normalized_laplacian_is_bit_exact_given_the_same_scaling():
    """Same dis on both sides: element = (r == c) - (dis[r] * count) * dis[c], rounded in that order."""
    from hgvae_amd import _lib
    gen = torch.Generator().manual_seed(6)
    n = 257
    ei = torch.randint(0, n, (2, 4000), generator=gen)
    deg = OS.compute_degree(ei, n)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    want = OS.normalized_laplacian(ei, n)
    d_ei, d_dis = ei.to(DEV), dis.to(DEV)
    ld = 260
    out = torch.full((n, ld), float("nan"), device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.call("hgv_normalized_laplacian", d_ei.data_ptr(), d_ei.stride(0), d_ei.stride(1), ei.size(1), n, d_dis.data_ptr(),
              out.data_ptr(), ld, status.data_ptr(), _lib.stream())
    assert torch.equal(out[:, :n].cpu(), want)
    assert float(out[:, n:].abs().max()) == 0.0 and int(status.item()) == 0


def test_out_of_range_edges_raise_like_the_reference():
    from hgvae_amd import spectral
    ei = torch.tensor([[0, 1, 7], [1, 0, 2]])
    with pytest.raises(ValueError):
        OS.normalized_laplacian(ei, 5)
    with pytest.raises(ValueError):
        spectral.normalized_laplacian(ei.to(DEV), 5)
    with pytest.raises(ValueError):
        spectral.normalized_laplacian(torch.zeros(3, 4, dtype=torch.int64, device=DEV), 5)


def test_eigenvalues_of_a_cycle_and_fiedler_value():
    """Undirected cycle C_n: eigenvalues 1 - cos(2 pi k / n); Fiedler value 1 - cos(2 pi / n)."""
    from hgvae_amd import spectral
    n = 64
    a = torch.arange(n)
    ei = torch.cat([torch.stack([a, (a + 1) % n]), torch.stack([(a + 1) % n, a])], dim=1)
    ev = spectral.laplacian_eigenvalues(ei.to(DEV), n).double().cpu()
    want = torch.sort(1 - torch.cos(2 * math.pi * torch.arange(n, dtype=torch.float64) / n)).values
    assert torch.allclose(ev, want, atol=2e-6)
    f = float(spectral.fiedler_value(ei.to(DEV), n))
    assert abs(f - (1 - math.cos(2 * math.pi / n))) < 2e-6
    assert abs(f - float(OS.compute_fiedler_value(ei, n))) < 2e-6


def test_l_frequency_reference_form_is_nan_and_the_wider_form_matches_the_restatement():
    from hgvae_amd import spectral
    gen = torch.Generator().manual_seed(7)
    n, m = 200, 100
    e1 = torch.randint(0, n, (2, 700), generator=gen)
    e1 = torch.cat([e1, e1.flip(0)], dim=1)
    e2 = torch.randint(0, m, (2, 300), generator=gen)
    e2 = torch.cat([e2, e2.flip(0)], dim=1)
    ref = OS.l_frequency(e1, e2, n, m)
    got = spectral.l_frequency(e1.to(DEV), e2.to(DEV), n, m)
    assert math.isnan(float(ref)) and math.isnan(float(got))          # one eigenvalue, min-max normalised: 0 / 0
    ref8 = OS.l_frequency(e1, e2, n, m, num_eigs=8)
    got8 = spectral.l_frequency(e1.to(DEV), e2.to(DEV), n, m, num_eigs=8)
    assert math.isfinite(float(got8)) and abs(float(got8) - float(ref8)) < 1e-4
    with pytest.raises(ValueError):
        spectral.l_frequency(e1.to(DEV), e2.to(DEV), None, m)


def test_coarsen_graph_spectral_probe_uses_the_device_laplacian():
    from hgvae_amd.vae import CoarsenGraph
    n = 48
    a = torch.arange(n)
    ei = torch.cat([torch.stack([a, (a + 1) % n]), torch.stack([(a + 1) % n, a])], dim=1).to(DEV)
    f = float(CoarsenGraph.fiedler_value(ei, n))
    assert abs(f - (1 - math.cos(2 * math.pi / n))) < 2e-6
