"""The claim behind `div_by_rcp` (minimodem_amd/csrc/mifsk_devlib.h): for floats x, c with a
normal quotient, (float)((double)x * rc) is the IEEE float quotient x / c whenever rc is 1 / c in
double to within a few units in the last place -- the exact quotient of two 24-bit significands
stays at least 2^-48 (relative) away from every float rounding boundary, the product is within
2^-51 of it.  A numpy model of that arithmetic (float64 multiply, float32 rounding) over random
and adversarial operands, with the reciprocal perturbed by up to +-2 ulp (what v_rcp_f64 plus two
Newton steps can be off by).  The device code itself is pinned by the parity tests: every frame's
confidence is compared with the oracle's bit for bit."""
import numpy as np


def _check(x, c):
    x = x.astype(np.float32)
    c = c.astype(np.float32)
    want = x / c                                            # IEEE float division
    normal = np.isfinite(want) & ((np.abs(want) >= np.finfo(np.float32).tiny) | (want == 0))
    rc = np.float64(1.0) / c.astype(np.float64)
    for k in (-2, -1, 0, 1, 2):
        r = rc
        for _ in range(abs(k)):
            r = np.nextafter(r, np.inf if k > 0 else -np.inf)
        got = (x.astype(np.float64) * r).astype(np.float32)
        bad = normal & (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        assert not bad.any(), (k, x[bad][:4], c[bad][:4], got[bad][:4], want[bad][:4])


def test_random_operands():
    rng = np.random.default_rng(1)
    n = 2_000_000
    with np.errstate(all="ignore"):
        # magnitudes as the confidence pass sees them, and the whole float range
        _check(rng.uniform(0, 2, n), rng.uniform(1e-3, 2, n))
        bits = rng.integers(0, 0x7F7FFFFF, size=n, dtype=np.uint32)
        bits2 = rng.integers(0x00800000, 0x7F7FFFFF, size=n, dtype=np.uint32)
        _check(bits.view(np.float32), bits2.view(np.float32))


def test_quotients_next_to_rounding_boundaries():
    """x = q * c for q one step either side of a float midpoint: the hardest quotients there are."""
    rng = np.random.default_rng(2)
    n = 500_000
    c = rng.integers(0x3F800000, 0x40000000, size=n, dtype=np.uint32).view(np.float32)     # [1, 2)
    q = rng.integers(0x3F800000, 0x40000000, size=n, dtype=np.uint32).view(np.float32)
    mid = q.astype(np.float64) + 2.0 ** -24                                                # a midpoint
    with np.errstate(all="ignore"):
        for eps in (-2.0 ** -47, 2.0 ** -47, -2.0 ** -40, 2.0 ** -40):
            x = (mid * (1 + eps) * c.astype(np.float64)).astype(np.float32)               # rounded: lands near
            _check(x, c)
        # small integers over small integers (the frame length, the class counts)
        a = rng.integers(1, 1 << 24, size=n).astype(np.float32)
        b = rng.integers(1, 64, size=n).astype(np.float32)
        _check(a, b)
        _check(a * np.float32(1e-7), np.full(n, 11, np.float32))
