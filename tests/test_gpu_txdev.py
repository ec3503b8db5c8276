"""TX side on the device: mifsk_tx_synthesize_batch must write the very
samples the host generator (csrc/mifsk_tx.cpp) writes -- which tests/test_tx_synth.py
pins to the WAV files of the reference's own transmitter -- and, directly, the samples
of the golden recordings made by the reference."""
import numpy as np
import pytest

import _golden as G
import minimodem_amd as M
from test_tx_synth import TX_OPTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU: the device transmitter has no CPU fallback")
    return torch, M.Context(0)


# every recording, the three made with --lut=0 included (t07, t11, t13: a sinf per sample,
# glibc's algorithm restated on the device -- csrc/mifsk_sinf.h)
@pytest.mark.parametrize("name", sorted(TX_OPTS))
def test_device_tx_equals_reference_tx_golden(gpu, name):
    torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    opts = dict(TX_OPTS[name])
    words = torch.from_numpy(np.frombuffer(g["payload"], np.uint8).copy()[None, :]).cuda()
    x, n = M.synthesize_batch(ctx, cfg, words, **opts)
    n = int(n[0])
    assert n == g["samples"].shape[0]
    assert x.cpu().numpy()[0, :n].tobytes() == g["samples"].tobytes()


@pytest.mark.parametrize("mode,kw", [
    ("1200", {}), ("300", {}), ("12000", {}), ("rtty", {}), ("tdd", {}), ("same", {}),
    ("1200", dict(msb_first=1)), ("1200", dict(invert_start_stop=1)),
    ("1200", dict(n_data_bits=7)), ("1200", dict(nstopbits=2.0)), ("600", dict(nstartbits=0, nstopbits=0.0)),
])
@pytest.mark.parametrize("s16", [False, True])
def test_device_tx_equals_host_generator_ragged_batch(gpu, mode, kw, s16):
    torch, ctx = gpu
    cfg = M.rx_config(mode, **kw)
    rng = np.random.default_rng(11)
    nstreams, maxw = 9, 300
    hi = 1 << min(8, int(cfg.n_data_bits))
    words = rng.integers(0, hi, size=(nstreams, maxw), dtype=np.uint8)
    nwords = np.array([maxw, 1, 0, 17, 255, 256, 257, 64, 299], np.int32)
    lead = np.array([0, 5, 100, 0, 3333, 1, 40, 41, 7], np.int32)
    for lut, amp in ((4096, 1.0), (16, 0.37), (1024, 1.7), (0, 1.0), (0, 0.37), (0, 1.7)):
        x, n = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda(),
                                  nwords=torch.from_numpy(nwords).cuda(), lut=lut, amplitude=amp,
                                  leading_silence=torch.from_numpy(lead).cuda(), s16=s16)
        x, n = x.cpu().numpy(), n.cpu().numpy()
        for i in range(nstreams):
            ref = M.synthesize(cfg, words[i, :nwords[i]], lut=lut, amplitude=amp,
                               leading_silence=int(lead[i]), s16=s16)
            assert int(n[i]) == ref.shape[0], (mode, i)
            assert x[i, :n[i]].tobytes() == ref.tobytes(), (mode, kw, lut, i)
            assert not x[i, n[i]:].any()


def test_device_sinf_equals_the_c_library_on_long_tones(gpu):
    """--lut=0 at 0.5 baud: 96000-sample tones drive the sinf argument to ~12 000 rad (the
    large-argument reduction of glibc's sinf); device == host generator (which calls libm)."""
    torch, ctx = gpu
    cfg = M.rx_config("0.5")
    words = torch.from_numpy(np.frombuffer(b"K", np.uint8).copy()[None, :]).cuda()
    for s16 in (False, True):
        x, n = M.synthesize_batch(ctx, cfg, words, lut=0, s16=s16)
        ref = M.synthesize(cfg, b"K", lut=0, s16=s16)
        assert int(n[0]) == ref.shape[0] > 1000000
        assert x.cpu().numpy()[0, :int(n[0])].tobytes() == ref.tobytes()


def test_device_tx_cuts_rows_that_are_too_short(gpu):
    torch, ctx = gpu
    cfg = M.rx_config("1200")
    words = torch.zeros((1, 4), dtype=torch.uint8).cuda()
    x, n = M.synthesize_batch(ctx, cfg, words, stride=64)    # row shorter than the stream: cut
    assert int(n[0]) > 64 and x.shape == (1, 64)


def test_device_tx_feeds_device_rx(gpu):
    """Generator -> demodulator without leaving the device."""
    torch, ctx = gpu
    cfg = M.rx_config("1200")
    rng = np.random.default_rng(5)
    words = rng.integers(32, 127, size=(64, 200), dtype=np.uint8)
    x, n = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda())
    res = M.results_to_host(M.demod_batch(ctx, cfg, x, nsamples=n, want=("bytes",)))
    for i in range(64):
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == words[i].tobytes()
