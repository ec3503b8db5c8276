"""mifsk_session_* (csrc/mifsk_session.cpp): a batch of streams fed in pieces from HOST memory
through the C ABI -- the unconsumed tails, the origins, the loop state, the RING cells and the
output arrays are the library's (reference: the loop that reads its stream half a samplebuf at a
time, src/minimodem.c:1144-1174).  Any cut must give, concatenated, the oracle's frames, bytes
and episodes, bit for bit."""
import ctypes as C

import numpy as np
import pytest

import _golden as G
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import minimodem_amd as M
    ctx = M.Context()
    yield M, torch, ctx
    ctx.close()


def _feed(M, ctx, cfg, streams, cuts, **kw):
    """cuts[i]: sorted cut positions of stream i (the same number for every stream)"""
    n, ncalls = len(streams), len(cuts[0]) + 1
    sess = M.Session(ctx, cfg, n, **kw)
    acc = [dict(frames=[], bytes=b"", episodes=[]) for _ in range(n)]
    for k in range(ncalls):
        new = []
        for i, x in enumerate(streams):
            e = [0] + list(cuts[i]) + [len(x)]
            new.append(x[e[k]:e[k + 1]])
        res = sess.feed(new, final=(k == ncalls - 1))
        for i, r in enumerate(res):
            assert r["status"] == 0
            acc[i]["frames"].append(r["frames"])
            acc[i]["bytes"] += r["bytes"]
            acc[i]["episodes"].append(r["episodes"])
            assert r["finished"] == (k == ncalls - 1)
            assert r["consumed"] + r["pending"] >= min(e[k + 1], r["consumed"])	# nothing fed is lost
    sess.close()
    for a in acc:
        a["frames"] = np.concatenate(a["frames"])
        a["episodes"] = np.concatenate(a["episodes"])
    return acc


@pytest.mark.parametrize("variant", ["library", "wave", "workgroup", "ring"])
@pytest.mark.parametrize("name", G.names())
def test_any_golden_fed_in_pieces_equals_the_oracle(gpu, name, variant):
    M, torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    x = g["samples"]
    if len(x) > 2000000:
        pytest.skip("0.5 baud: one samplebuf is longer than the recording's pieces")
    if variant == "workgroup" and cfg.auto_carrier_threshold > 0:
        pytest.skip("--auto-carrier runs on the wavefront engine")
    ring = variant == "ring"
    ref = O.oracle_rx_stream(ocfg, x, ring_mode=ring)
    rng = np.random.default_rng(len(x) + 7)
    for trial in range(2):
        cuts = sorted(int(c) for c in rng.integers(0, len(x) + 1, size=3))
        if trial == 1:
            cuts = [1, 2, len(x) // 2, len(x)]                 # tiny pieces, and an empty final one
        got = _feed(M, ctx, cfg, [x], [cuts], ring_exact=ring,
                    engine=None if variant in ("library", "ring") else variant)[0]
        assert got["frames"].tobytes() == ref["frames"].tobytes(), (name, cuts)
        assert got["bytes"] == ref["bytes"], (name, cuts)
        assert got["episodes"].tobytes() == ref["episodes"].tobytes(), (name, cuts)


@pytest.mark.parametrize("mode", ["1200", "300", "12000", "same", "rtty"])
def test_ragged_noisy_batch_cut_differently_per_stream(gpu, mode):
    M, torch, ctx = gpu
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    rng = np.random.default_rng(hash(mode) % 1000)
    nwords = {"1200": 60, "300": 24, "12000": 200, "same": 40, "rtty": 6}[mode]
    streams = []
    for i in range(9):
        words = rng.integers(0 if mode == "rtty" else 32, 32 if mode == "rtty" else 127, size=nwords + i, dtype=np.uint8)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 300)))
        if i % 3 == 1:                                         # a second burst behind a gap
            x = np.concatenate([x, np.zeros(int(rng.integers(100, 3000)), np.float32),
                                M.synthesize(cfg, words[: nwords // 2])])
        streams.append((x + rng.normal(0, 0.08, x.shape)).astype(np.float32))
    cuts = [sorted(int(c) for c in rng.integers(0, len(x) + 1, size=5)) for x in streams]
    got = _feed(M, ctx, cfg, streams, cuts)
    for i, x in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, x)
        assert got[i]["frames"].tobytes() == ref["frames"].tobytes(), (mode, i)
        assert got[i]["bytes"] == ref["bytes"] and got[i]["episodes"].tobytes() == ref["episodes"].tobytes()
    # and without the per-frame records: bytes and episodes all the same
    lean = _feed_lean(M, ctx, cfg, streams, cuts)
    for i in range(len(streams)):
        assert lean[i] == got[i]["bytes"]


def _feed_lean(M, ctx, cfg, streams, cuts):
    sess = M.Session(ctx, cfg, len(streams), want_frames=False)
    out = [b""] * len(streams)
    for k in range(len(cuts[0]) + 1):
        new = []
        for i, x in enumerate(streams):
            e = [0] + list(cuts[i]) + [len(x)]
            new.append(x[e[k]:e[k + 1]])
        for i, r in enumerate(sess.feed(new, final=(k == len(cuts[0])))):
            assert len(r["frames"]) == 0
            out[i] += r["bytes"]
    sess.close()
    return out


def test_argument_errors(gpu):
    M, torch, ctx = gpu
    from minimodem_amd import _lib
    lib = _lib.load()
    cfg = M.rx_config("1200")
    h = C.c_void_p()
    assert lib.mifsk_session_create(C.byref(h), ctx.handle, C.byref(cfg), 0, 0) == -22
    assert lib.mifsk_session_create(C.byref(h), None, C.byref(cfg), 1, 0) == -22
    assert lib.mifsk_session_create(C.byref(h), ctx.handle, C.byref(cfg), 1, _lib.IO_ENGINE_WAVE | _lib.IO_ENGINE_WORKGROUP) == -22
    assert lib.mifsk_session_create(C.byref(h), ctx.handle, C.byref(cfg), 1, _lib.IO_RING_EXACT | _lib.IO_ENGINE_WORKGROUP) == -22
    assert lib.mifsk_session_create(C.byref(h), ctx.handle, C.byref(cfg), 1, 0x40) == -22
    s = M.Session(ctx, cfg, 2)
    x = M.synthesize(cfg, np.arange(40, 80, dtype=np.uint8))
    cnt = (C.c_uint32 * 2)(5, 0)
    assert lib.mifsk_session_feed(s.handle, None, cnt, 0) == -22           # a count without its samples
    r = s.feed([x, None], final=True)
    assert r[0]["bytes"] == bytes(range(40, 80)) and r[1]["bytes"] == b"" and r[1]["finished"]
    with pytest.raises(RuntimeError):
        s.feed([None, None], final=True)                                   # the final piece has been fed
    assert not lib.mifsk_session_get(s.handle, 2) and lib.mifsk_session_pending(s.handle, -1) == 0
    s.close()
