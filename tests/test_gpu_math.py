"""The short square root of band_mag2() (minimodem_amd/csrc/mifsk_devmath.h) against the exact,
correctly rounded sequence it stands in for, on the device: 2^32 sums of squares, half of them
placed within a few hundred units in the last place of a float rounding boundary -- where the two
could part.  What the magnitudes feed (reference: hypotf in band_mag, /root/reference/src/fsk.c:
107-114) is compared bit for bit with the oracle by every parity test; this pins the guard itself."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import minimodem_amd as M
    c = M.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", [1, 2026])
def test_short_sqrt_equals_the_exact_sequence_wherever_the_guard_lets_it_through(ctx, seed):
    n = 1 << 31
    bad, guarded, raw_bad, total = ctx.selftest_sqrt(n, seed=seed)
    assert total >= n
    assert bad == 0, "%d of %d values: short path accepted, result differs" % (bad, total)
    # the guard is there for something (the boundary cases do part without it) ...
    assert raw_bad > 0
    # ... and costs next to nothing on ordinary values: the boundary half is guarded by design
    # (1201 placements around the midpoint, 2 * 4096 wide guard: all of them), the random half
    # at 2^13 / 2^29 plus the non-finite / tiny inputs
    assert guarded <= 0.56 * total
