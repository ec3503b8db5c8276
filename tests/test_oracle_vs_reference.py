"""Live comparison of the restatement with the reference built from its own
sources (oracle/_ref) on the reference's OWN test inputs (tests/*.test in
/root/reference).  Runs only where both exist, i.e. in the build container;
the committed goldens (test_oracle_golden.py) cover the same ground elsewhere."""
import os

import numpy as np
import pytest

import _oracle as O

REF_TESTS = "/root/reference/tests"
pytestmark = pytest.mark.skipif(
    not (O.have_ref() and os.path.isdir(REF_TESTS)),
    reason="needs oracle/_ref and /root/reference/tests (build container only)")

P24 = "--samplerate 24000 -M 1200 -S 2400"
K24 = dict(sample_rate=24000, mark_f=1200, space_f=2400)
# (reference test, payload file, tx args, rx args, config kwargs, expect "perfect")
CASES = [
    ("01-self-test-1200", "testdata-ascii.txt", "1200", "1200", dict(baudmode="1200"), False),
    ("02-self-test-300", "testdata-ascii.txt", "300", "300", dict(baudmode="300"), False),
    ("05-self-test-12000", "testdata-ascii.txt", "12000", "12000", dict(baudmode="12000"), False),
    ("06-float-samples", "testdata-ascii.txt", "12000 --float-samples", "12000",
     dict(baudmode="12000"), False),
    ("07-no-lut", "testdata-ascii.txt", "1200 --lut=0", "1200", dict(baudmode="1200"), False),
    ("08-lut16", "testdata-ascii.txt", "1200 --lut=16", "1200", dict(baudmode="1200"), False),
    ("09-lut16-float", "testdata-ascii.txt", "1200 --lut=16 --float-samples", "1200",
     dict(baudmode="1200"), False),
    ("10-verify-perfect", "testdata-ascii.txt", "1200 " + P24, "1200 " + P24,
     dict(baudmode="1200", **K24), True),
    ("11-perfect-nolut", "testdata-ascii.txt", "1200 --lut=0 " + P24, "1200 " + P24,
     dict(baudmode="1200", **K24), True),
    ("12-perfect-lut16", "testdata-ascii.txt", "1200 --lut=16 " + P24, "1200 " + P24,
     dict(baudmode="1200", **K24), True),
    ("13-perfect-nolut-float", "testdata-ascii.txt", "1200 --lut=0 --float-samples " + P24,
     "1200 " + P24, dict(baudmode="1200", **K24), True),
    ("14-perfect-lut16-float", "testdata-ascii.txt", "1200 --lut=16 --float-samples " + P24,
     "1200 " + P24, dict(baudmode="1200", **K24), True),
    ("15-perfect-float", "testdata-ascii.txt", "1200 --float-samples " + P24, "1200 " + P24,
     dict(baudmode="1200", **K24), True),
    ("21-rate-slop-292", "testdata-ascii.txt", "292", "300", dict(baudmode="300"), False),
    ("21-rate-slop-308", "testdata-ascii.txt", "308", "300", dict(baudmode="300"), False),
    ("30-amplitude-0.01", "testdata-ascii.txt", "--volume 0.01 1200", "1200",
     dict(baudmode="1200"), False),
    ("30-amplitude-E", "testdata-ascii.txt", "--volume E 1200", "1200",
     dict(baudmode="1200"), False),
    ("60-multibyte", "testdata-multibyte.txt", "1200", "1200", dict(baudmode="1200"), False),
    ("80-SAME", "testdata-ascii.txt", "same", "same", dict(baudmode="same"), False),
    ("81-ascii7", "testdata-ascii.txt", "1200 -7", "1200 -7",
     dict(baudmode="1200", n_data_bits=7), False),
    ("03-rtty", "testdata-baudot.txt", "rtty", "rtty", dict(baudmode="rtty"), False),
    ("81-tdd", "testdata-baudot.txt", "tdd", "tdd", dict(baudmode="tdd"), False),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_restatement_equals_reference_program(case, tmp_path):
    name, payload_file, tx, rx, kw, perfect = case
    payload = open(os.path.join(REF_TESTS, payload_file), "rb").read()
    wav = str(tmp_path / "x.wav")
    O.ref_tx(payload, tx.split(), wav)
    out, err = O.ref_rx(wav, rx.split())
    ref_lines = [l for l in err.splitlines() if l.startswith("### NOCARRIER")]
    assert ref_lines
    if perfect:
        assert "confidence=inf" in ref_lines[-1] and "(rate perfect)" in ref_lines[-1]
    sr, x = O.read_wav(wav)
    cfg = O.oracle_config(**kw)
    assert sr == cfg.sample_rate
    for ring in (True, False):
        r = O.oracle_rx_stream(cfg, x, ring_mode=ring)
        if cfg.decoder == 0:
            assert r["bytes"] == out
            if "-7" not in tx and "E" not in tx:
                assert out == payload
        assert [O.format_nocarrier(cfg, e) for e in r["episodes"]] == ref_lines


def test_dc_offset_sweep_tests_40_41(tmp_path):
    """--Xrxnoise adds a DC offset of -factor to every sample (rand()/RAND_MAX is
    an integer division, src/simpleaudio-sndfile.c:64-70); tests 40/41 sweep it."""
    payload = open(os.path.join(REF_TESTS, "testdata-ascii.txt"), "rb").read()
    for flags, kw in (("1200", dict(baudmode="1200")),
                      ("1200 " + P24, dict(baudmode="1200", **K24))):
        wav = str(tmp_path / "n.wav")
        O.ref_tx(payload, (flags + " --volume 0.5").split(), wav)
        sr, x = O.read_wav(wav)
        for noise in (0.0, 0.05, 0.10, 0.50):
            out, err = O.ref_rx(wav, (flags + " --Xrxnoise %g --rx-one" % noise).split())
            ref_lines = [l for l in err.splitlines() if l.startswith("### NOCARRIER")]
            cfg = O.oracle_config(rx_one=1, **kw)
            xn = x + np.float32((0.0 - 0.5) * (noise * 2)) if noise else x
            r = O.oracle_rx_stream(cfg, xn.astype(np.float32), ring_mode=False)
            assert r["bytes"] == out == payload
            assert [O.format_nocarrier(cfg, e) for e in r["episodes"]] == ref_lines


def test_find_frame_against_reference_fsk_c():
    """Call level: reference src/fsk.c (+FFT shim) vs restatement on random windows of
    noisy FSK; integers exact, magnitudes to float tolerance."""
    rng = np.random.default_rng(7)
    for mode in ("1200", "300", "same", "12000"):
        cfg = O.oracle_config(mode)
        payload = bytes(rng.integers(32, 127, size=40, dtype=np.uint8))
        import tempfile
        wav = O.tmp_wav()
        O.ref_tx(payload, [mode, "--float-samples"], wav)
        sr, x = O.read_wav(wav)
        os.unlink(wav)
        x = (x + rng.normal(0, 0.2, x.shape)).astype(np.float32)
        xp = np.concatenate([x, np.zeros(4 * int(cfg.expect_nsamples), np.float32)])
        rl, ol = O.ref_lib(), O.oracle_lib()
        rp = rl.fsk_plan_new(float(sr), cfg.mark_f, cfg.space_f, cfg.band_width)
        op = ol.ofsk_plan_new(float(sr), cfg.mark_f, cfg.space_f, cfg.band_width)
        assert (rp.contents.fftsize, rp.contents.b_mark, rp.contents.b_space) == \
               (op.contents.fftsize, op.contents.b_mark, op.contents.b_space)
        for off in rng.integers(0, len(x) - 1, size=200):
            for ci, step, lim, exp in ((0, cfg.try_step[0], cfg.search_limit, cfg.expect_sync),
                                       (1, cfg.try_step_fine[1], float("inf"), cfg.expect_data)):
                args = (xp[int(off):], int(cfg.expect_nsamples), int(cfg.try_first[ci]),
                        int(cfg.try_max[ci]), int(step), lim, exp)
                rc, rb, ra, rs = O.ref_find_frame(rp, *args)
                oc, ob, oa, os_ = O.oracle_find_frame(op, *args)
                assert (rb, rs) == (ob, os_)
                assert oc == pytest.approx(rc, rel=1e-5, abs=1e-6)
                assert oa == pytest.approx(ra, rel=1e-5, abs=1e-6)
        rl.fsk_plan_destroy(rp)
        ol.ofsk_plan_destroy(op)
