"""BASELINE.json-sized GPU runs checked through size-independent properties:
encode -> decode round trip of every stream (the decoded bytes must be the
transmitted payload), frame/byte accounting, and a sample of streams compared
bit-for-bit with the oracle.  (Whole-batch oracle comparison at the benchmark
size is done inside bench.py's cpu_port leg on every default run.)"""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import minimodem_amd as M
    ctx = M.Context()
    yield M, torch, ctx
    ctx.close()


def _batch(M, cfg, nstreams, nwords, seed, lo=0x20, hi=0x7F, nsamples=None):
    rng = np.random.default_rng(seed)
    streams, payloads = [], []
    for i in range(nstreams):
        w = rng.integers(lo, hi, size=nwords, dtype=np.uint8)
        streams.append(M.synthesize(cfg, w, leading_silence=int(rng.integers(0, 41))))
        payloads.append(w)
    n = nsamples or max(len(s) for s in streams)
    n = (n + 3) & ~3
    host = np.zeros((nstreams, n), np.float32)
    for i, s in enumerate(streams):
        host[i, :len(s)] = s
    return host, payloads


def _run(M, torch, ctx, cfg, host, want=("bytes", "episodes")):
    out = M.demod_batch(ctx, cfg, torch.from_numpy(host).cuda(), want=want, episodes_cap=4)
    torch.cuda.synchronize()
    return M.results_to_host(out)


def test_config2_bell202_1024_streams_x_10s(gpu):
    """BASELINE configs[1]: 1024 streams x 480000 samples, 1200 baud."""
    M, torch, ctx = gpu
    cfg = M.rx_config("1200")
    host, payloads = _batch(M, cfg, 1024, 1199, seed=42, nsamples=480000)
    res = _run(M, torch, ctx, cfg, host, want=("bytes", "episodes", "frames"))
    for i in range(1024):
        nb = int(res["nbytes"][i])
        assert res["bytes"][i, :nb].tobytes() == payloads[i].tobytes(), i
        assert int(res["nframes"][i]) == 1199 and int(res["nepisodes"][i]) == 1
        ep = res["episodes"][i, 0]
        assert int(ep["nframes"]) == 1199 and int(ep["end_reason"]) == 2
        # frames advance by one frame length, give or take the tracker's search range
        st = res["frames"][i, :1199]["start"].astype(np.int64)
        d = np.diff(st)
        assert d.min() >= 400 - 20 and d.max() <= 400 + 30 and np.median(d) == 400
    ocfg = O.oracle_config("1200")
    for i in (0, 511, 1023):
        ref = O.oracle_rx_stream(ocfg, host[i])
        assert res["frames"][i, :1199].tobytes() == ref["frames"].tobytes()
        assert res["episodes"][i, :1].tobytes() == ref["episodes"].tobytes()


def test_config4_12000_baud_8192_streams(gpu):
    """BASELINE configs[3], one GPU's shard: 8192 streams x 2 s at 12000 baud."""
    M, torch, ctx = gpu
    cfg = M.rx_config("12000")
    host, payloads = _batch(M, cfg, 8192, 2395, seed=7, nsamples=96000)
    res = _run(M, torch, ctx, cfg, host)
    bad = [i for i in range(8192)
           if res["bytes"][i, :int(res["nbytes"][i])].tobytes() != payloads[i].tobytes()]
    assert not bad, bad[:10]
    ocfg = O.oracle_config("12000")
    for i in (0, 4095, 8191):
        assert O.oracle_rx_stream(ocfg, host[i])["bytes"] == payloads[i].tobytes()


def test_config3_rtty_long_windows(gpu):
    """BASELINE configs[2] shape (1056-sample bit windows), reduced stream count."""
    M, torch, ctx = gpu
    cfg = M.rx_config("rtty")
    host, payloads = _batch(M, cfg, 128, 60, seed=3, lo=0, hi=32)
    res = _run(M, torch, ctx, cfg, host, want=("bits", "episodes"))
    ocfg = O.oracle_config("rtty")
    for i in range(0, 128, 9):
        ref = O.oracle_rx_stream(ocfg, host[i])
        nf = int(res["nframes"][i])
        assert nf == len(ref["frames"]) >= 58
        assert np.array_equal(res["bits"][i, :nf], ref["frames"]["bits"])
    # 5-bit words come back as transmitted (the frame before the first one may be the leader)
    for i in range(128):
        nf = int(res["nframes"][i])
        got = res["bits"][i, :nf].astype(np.uint8)
        assert bytes(payloads[i]) in bytes(got), i


def test_config5_same_with_noise_sweep(gpu):
    """BASELINE configs[4] shape: NOAA SAME, amplitude 0.5, AWGN SNR sweep; decode
    must equal the oracle's on identical buffers at every SNR (whether or not the
    payload survives), plus the reference's DC-offset sweep (tests/40-noise.test)."""
    M, torch, ctx = gpu
    cfg = M.rx_config("same")
    ocfg = O.oracle_config("same")
    rng = np.random.default_rng(11)
    host, payloads = _batch(M, cfg, 48, 40, seed=5)
    host *= np.float32(0.5)
    p_sig = 0.5 ** 2 / 2
    noisy = host.copy()
    labels = []
    for i in range(48):
        kind = i % 8
        if kind < 6:
            snr_db = [None, 20, 12, 9, 6, 3][kind]
            if snr_db is not None:
                sigma = np.sqrt(p_sig / 10 ** (snr_db / 10))
                noisy[i] += rng.normal(0, sigma, noisy.shape[1]).astype(np.float32)
            labels.append(("snr", snr_db))
        else:
            dc = [0.05, 0.50][kind - 6]
            noisy[i] -= np.float32(dc)
            labels.append(("dc", dc))
    res = _run(M, torch, ctx, cfg, noisy)
    decoded_ok = 0
    for i in range(48):
        ref = O.oracle_rx_stream(ocfg, noisy[i])
        nb = int(res["nbytes"][i])
        assert res["bytes"][i, :nb].tobytes() == ref["bytes"], (i, labels[i])
        decoded_ok += ref["bytes"] == payloads[i].tobytes()
    assert decoded_ok >= 12		# at least the clean and the 20 dB streams
