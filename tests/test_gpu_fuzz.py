"""Random shapes (row counts around the tile sizes, widths 1 ... 288, one or two sources, empty and ragged graphs) through the
projection, aggregation and level-epilogue operators against float64 torch on the device; tests/fuzz_ops.py holds the cases."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_against_float64(seed):
    import fuzz_ops
    fail, _ = fuzz_ops.run(250, seed)
    assert fail is None, fail
