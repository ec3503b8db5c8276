"""The host-side synthetic-input generator (csrc/mifsk_tx.cpp) must write the
very samples `minimodem --tx --file` writes.  Checked against the committed
goldens (made by the reference program) and, where oracle/_ref exists, live."""
import os

import numpy as np
import pytest

import _golden as G
import _oracle as O
import minimodem_amd as M

TX_OPTS = {   # golden name -> synthesize() options
    "t01_1200": dict(s16=True),
    "t02_300": dict(s16=True),
    "t05_12000": dict(s16=True),
    "t06_1200_float": dict(),
    "t07_1200_nolut": dict(s16=True, lut=0),
    "t08_1200_lut16": dict(s16=True, lut=16),
    "t10_perfect": dict(s16=True),
    "t13_perfect_nolut_float": dict(lut=0),
    "t14_perfect_lut16_float": dict(lut=16),
    "t30_ampl_0p3": dict(s16=True, amplitude=0.30),
    "t60_7bit": dict(s16=True),
    "t80_same": dict(s16=True),
    "t09_1200_lut16_float": dict(lut=16),
    "t11_perfect_nolut": dict(s16=True, lut=0),
    "t12_perfect_lut16": dict(s16=True, lut=16),
    "t15_perfect_float": dict(),
    "t30_ampl_3p50": dict(s16=True, amplitude=3.5),
    "t30_ampl_0p01": dict(s16=True, amplitude=0.01),
    "t31_ampl_float_3p50": dict(amplitude=3.5),
    "t31_ampl_float_0p01": dict(amplitude=0.01),
}


@pytest.mark.parametrize("name", sorted(TX_OPTS))
def test_synth_equals_reference_tx_golden(name):
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    x = M.synthesize(cfg, g["payload"], **TX_OPTS[name])
    assert x.shape == g["samples"].shape
    assert np.array_equal(x, g["samples"])


@pytest.mark.skipif(not O.have_ref(), reason="needs oracle/_ref")
def test_synth_equals_reference_tx_live(tmp_path):
    rng = np.random.default_rng(3)
    payload = bytes(rng.integers(0, 256, size=300, dtype=np.uint8))
    for mode, extra, opts in (("1200", ["--float-samples"], dict()),
                              ("300", ["--float-samples", "--volume", "0.5"], dict(amplitude=0.5)),
                              ("same", [], dict(s16=True)),
                              ("12000", ["--lut=0"], dict(s16=True, lut=0))):
        wav = str(tmp_path / "a.wav")
        O.ref_tx(payload, [mode] + extra, wav)
        sr, ref = O.read_wav(wav)
        cfg = M.rx_config(mode)
        x = M.synthesize(cfg, payload, **opts)
        assert np.array_equal(x, ref), mode
