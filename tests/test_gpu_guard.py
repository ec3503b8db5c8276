"""The shared-segment exactness guard, forced.

A long-window search sums every window from shared segments and accepts the assembled value X
only where (float)(X - d) == (float)(X + d) for the rounding bound d = c*u*A; a window that
fails is summed again in index order (the oracle's operation sequence).  These tests BUILD
inputs whose index-order sum sits on a float rounding boundary -- the midpoint of two adjacent
floats, to within 1e-11, where d is about 1e-10 -- for chosen windows of chosen frames, and
check (a) through the work counters that the index-order fallback fired for them (the same
stream tuned only coarsely, 1e-7 off the boundary, runs without a single fallback), and (b)
that every frame record -- confidence and amplitude bit patterns included -- is the
oracle's.  Without the guard the assembled sums (which differ from the index-order ones by
up to d) would round to the other float about every second time.

Construction: for window [a, a+B) and one of the four sums (mark / space, re / im), two
samples with a large twiddle in that sum are replaced -- the first moves the sum to within
the float granularity of the sample (1e-7) of the boundary, the second, a tiny value with a
fine granularity, the rest of the way; the index-order sum is re-evaluated by the oracle's own
chain (ofsk_bit_dft_f64) after each step.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

CNT_SEG_SCANS, CNT_FALLBACK = 20, 21


@pytest.fixture(scope="module")
def gpu():
    import torch
    import minimodem_amd as M
    ctx = M.Context()
    yield M, torch, ctx
    ctx.close()


def _dft64(lib, plan, x, a, B):
    out = (C.c_double * 4)()
    w = np.ascontiguousarray(x[a:a + B], dtype=np.float32)
    lib.ofsk_bit_dft_f64(plan, w.ctypes.data, B, out)
    return [out[0], out[1], out[2], out[3]]


def _boundary_near(v):
    """the rounding boundary (midpoint of two adjacent floats) nearest the double v"""
    f = np.float32(v)
    up = np.nextafter(f, np.float32(np.inf))
    dn = np.nextafter(f, np.float32(-np.inf))
    m_up = (float(f) + float(up)) / 2.0
    m_dn = (float(f) + float(dn)) / 2.0
    return m_up if abs(m_up - v) < abs(m_dn - v) else m_dn


def _tune(lib, plan, cfg, x, a, comp, fine=True):
    """Put sum `comp` (0 mark re, 1 mark im, 2 space re, 3 space im) of window [a, a + B) on a
    float rounding boundary.  Returns the distance left (double)."""
    B = int(cfg.bit_nsamples)
    b = int(cfg.b_mark) if comp < 2 else int(cfg.b_space)
    tw = np.zeros((B, 2))
    w2 = (C.c_double * 2)()
    for n in range(B):
        lib.ofsk_twiddle(b, n, int(cfg.fftsize), w2)
        tw[n] = (w2[0], w2[1])
    coef = tw[:, comp & 1]
    # two samples in the middle of the window with a twiddle near +-1 in this sum
    cand = [n for n in range(B // 4, 3 * B // 4) if abs(coef[n]) > 0.95]
    n1, n2 = cand[0], cand[len(cand) // 2]
    x[a + n1] = 0.0
    x[a + n2] = 0.0
    X = _dft64(lib, plan, x, a, B)[comp]
    assert abs(X) > 50.0                 # (the caller picks the band this bit carries)
    target = _boundary_near(X)
    if not fine:
        # stop a hair short of the boundary: 1e-6 of it, thousands of rounding bounds away
        x[a + n1] = np.float32((target - X + 1e-6) / coef[n1])
        return abs(target - _dft64(lib, plan, x, a, B)[comp])
    x[a + n1] = np.float32((target - X) / coef[n1])
    for _ in range(4):
        X = _dft64(lib, plan, x, a, B)[comp]
        r = target - X
        if abs(r) < 2e-12:
            break
        x[a + n2] = np.float32(float(x[a + n2]) + r / coef[n2])
    return abs(target - _dft64(lib, plan, x, a, B)[comp])


def _run(M, torch, ctx, cfg, x):
    n = (len(x) + 3) & ~3
    host = np.zeros((1, n), np.float32)
    host[0, :len(x)] = x
    out = M.demod_batch(ctx, cfg, torch.from_numpy(host).cuda(),
                        nsamples=torch.tensor([len(x)], dtype=torch.int32).cuda(),
                        want=("bytes", "frames", "episodes", "counters"), episodes_cap=8)
    torch.cuda.synchronize()
    return M.results_to_host(out)


def _equal_oracle(res, ocfg, x):
    ref = O.oracle_rx_stream(ocfg, x)
    nf = int(res["nframes"][0])
    return nf == len(ref["frames"]) and res["frames"][0, :nf].tobytes() == ref["frames"].tobytes() \
        and res["bytes"][0, :int(res["nbytes"][0])].tobytes() == ref["bytes"]


@pytest.mark.parametrize("mode,kw", [("rtty", {}), ("rtty", dict(sample_rate=44100))], ids=["rtty48k", "rtty44k1"])
def test_windows_on_a_rounding_boundary_take_the_index_order_fallback(gpu, mode, kw):
    M, torch, ctx = gpu
    cfg = M.rx_config(mode, **kw)
    ocfg = O.oracle_config(mode, **kw)
    lib = O.oracle_lib()
    plan = lib.ofsk_plan_new(float(cfg.sample_rate), cfg.mark_f, cfg.space_f, cfg.band_width)
    assert M.demod_plan(ctx, cfg, 1)["engine"] == "wave"
    B = int(cfg.bit_nsamples)
    tuned = 0
    for seed in range(40):
        rng = np.random.default_rng(900 + seed)
        words = rng.integers(0, 32, size=24, dtype=np.uint8)
        x0 = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 40)), amplitude=0.8)
        x0 = np.concatenate([x0, np.zeros(4 * B, np.float32)])
        r0 = _run(M, torch, ctx, cfg, x0)
        assert _equal_oracle(r0, ocfg, x0)
        c0 = r0["counters"][0]
        assert int(c0[CNT_SEG_SCANS]) > 10            # the searches do go through shared segments
        if int(c0[CNT_FALLBACK]) != 0:
            continue                                   # (a natural fallback: not a clean baseline)
        nf = int(r0["nframes"][0])
        fr = r0["frames"][0, :nf]
        # four windows of four different frames: bit k of frame f starts at frame.start +
        # bit_offset[k] (fsk.c:204,249); bits 0 and 7 of "10ddddd1" are mark, bit 1 (the start
        # bit) is space; of the band's two sums the larger one is tuned
        picks = [(6, 0, 0), (9, 7, 0), (12, 1, 2), (15, 1, 2)]
        coarse, fine = x0.copy(), x0.copy()
        dist = []
        for f, k, band in picks:
            a = int(fr[f]["start"]) + int(cfg.bit_offset[k])
            X = _dft64(lib, plan, x0, a, B)
            comp = band + (1 if abs(X[band + 1]) > abs(X[band]) else 0)
            dc = _tune(lib, plan, cfg, coarse, a, comp, fine=False)
            df = _tune(lib, plan, cfg, fine, a, comp, fine=True)
            assert 5e-7 < dc < 2e-6, dc
            assert df < 1e-11, df
            dist.append(df)
        rc = _run(M, torch, ctx, cfg, coarse)
        rf = _run(M, torch, ctx, cfg, fine)
        # bit-exact either way ...
        assert _equal_oracle(rc, ocfg, coarse)
        assert _equal_oracle(rf, ocfg, fine)
        # ... the frames are where they were (two samples of a 1056-sample window changed)
        assert int(rf["nframes"][0]) == nf and np.array_equal(rf["frames"][0, :nf]["start"], fr["start"])
        if int(rc["counters"][0][CNT_FALLBACK]) != 0:
            continue                                   # (the coarse edit happened to create one)
        # ... and the windows ON the boundary went back to the index-order sum: at least one
        # fallback pass per tuned frame's search
        assert int(rf["counters"][0][CNT_FALLBACK]) >= len(picks), \
            (rf["counters"][0][CNT_FALLBACK], dist)
        tuned += 1
        if tuned >= 3:
            break
    assert tuned >= 1, "no stream with a fallback-free baseline among the seeds"
    lib.ofsk_plan_destroy(plan)


@pytest.mark.parametrize("mode,kw", [("rtty", {}), ("rtty", dict(sample_rate=44100))], ids=["rtty48k", "rtty44k1"])
def test_guard_where_the_energy_sums_underflow(gpu, mode, kw):
    """The guard's A = sum |x| comes from FLOAT sums of FLOAT squares: below 1e-19 of full scale a
    square is a subnormal, below 1e-23 it is zero -- and a bound of zero would accept every
    assembled sum.  What those roundings can lose is part of the bound (csrc/mifsk_wave.hip:
    dfloor, and the second look's A_s): streams at 1e-18 ... 1e-36 of full scale go through the
    shared segments and give the oracle's frames, confidence and amplitude bit patterns included."""
    M, torch, ctx = gpu
    cfg = M.rx_config(mode, **kw)
    ocfg = O.oracle_config(mode, **kw)
    rng = np.random.default_rng(77)
    n_scans = 0
    for amp in (1e-18, 1e-20, 3e-22, 1e-23, 1e-25, 1e-30, 1e-36):
        words = rng.integers(0, 32, size=20, dtype=np.uint8)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 40)), amplitude=1.0)
        x = (x.astype(np.float64) * amp).astype(np.float32)
        x = np.concatenate([x, np.zeros(4 * int(cfg.bit_nsamples), np.float32)])
        r = _run(M, torch, ctx, cfg, x)
        assert int(r["nframes"][0]) >= 20, (amp, int(r["nframes"][0]))
        assert _equal_oracle(r, ocfg, x), amp
        n_scans += int(r["counters"][0][CNT_SEG_SCANS])
    assert n_scans > 100
