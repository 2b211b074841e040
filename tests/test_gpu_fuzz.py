"""Random shapes (row counts around the tile sizes, widths 1 ... 288, one or two sources, empty and ragged graphs) through the
projection, aggregation and level-epilogue operators against float64 torch on the device; tests/fuzz_ops.py holds the cases."""
import pytest

pytestmark = pytest.mark.gpu


# (seed, cases): seed 5 reaches a whole-step case that roundings decide (case 312: the harness asks the float64 restatement
# how far its own gradients move under a 2e-7 input perturbation), seed 11 a 1 x 1 attention layer whose weight gradient is one
# cancelling sum (case 320), seed 15 the KL of a one-element batch (case 54)
@pytest.mark.parametrize("seed,cases", [(11, 330), (12, 250), (5, 320), (15, 80)])
def test_random_shapes_against_float64(seed, cases):
    import fuzz_ops
    fail, _ = fuzz_ops.run(cases, seed)
    assert fail is None, fail


@pytest.mark.parametrize("seed,cases", [(21, 45), (22, 45), (23, 45)])
def test_random_whole_steps_with_levels_deferred_at_every_size(seed, cases, monkeypatch):
    """The whole-step cases (four model variants, random graphs / widths / frozen parameters) with the large-batch dataflow
    forced on at every size: levels kept as (d, s, t), consumers folding the normalisation, backward sums from the consumers'
    products, and - at hidden width 138 - projections that store the activated level and its column sums."""
    import fuzz_ops
    from hgvae_amd import functional as Fn
    monkeypatch.setattr(Fn, "DEFER_MIN_ROWS", 0)
    names = set()
    real_call = Fn._lib.call

    def spy_call(name, *a, _real=real_call):
        names.add(name)
        return _real(name, *a)
    monkeypatch.setattr(Fn._lib, "call", spy_call)
    fail, worst, widened = fuzz_ops.run_models(cases, seed)
    assert fail is None, fail
    assert widened <= max(2, cases // 10), (widened, cases)
    for expected in ("hgv_bn_stats_act", "hgv_pack16h_weights_affine", "hgv_wgrad_finish", "hgv_bn_act_bwd_pre_alg",
                     "hgv_gemm_nt16h_act_stats", "hgv_bn_stats_merge", "hgv_agg_fwd_ind"):
        assert expected in names, expected
