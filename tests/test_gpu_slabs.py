"""Streams that arrive in pieces (mifsk_demod_slab, SURVEY 8 e; reference: the half-buffer
refills of src/minimodem.c:1144-1174 -- its loop never sees more than one samplebuf at a time).
Any cut of a stream into slabs must give the frames, bytes and episodes of the single call --
and of the oracle -- bit for bit: the loop's state (cursor, buffer arithmetic, carrier totals,
tracker, --auto-carrier band) is carried in device memory from call to call.  Both engines
(the wavefront kernel's and the workgroup kernel's resumable instantiations) and the two taking
turns on one stream: the state record is common to them."""
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


def _feed_in_slabs(M, ctx, cfg, streams, cuts_per_stream, episodes_cap=32, engine=None, ring=False):
    """cuts_per_stream[i] = sorted cut positions of stream i (same count for all streams).
    Returns per-stream dict(frames, bytes, episodes) concatenated over the calls.
    engine: "wave", "workgroup", None (the library's choice) or "alternate" (the engines take
    turns from slab to slab: one state record serves both)."""
    n = len(streams)
    ncalls = len(cuts_per_stream[0]) + 1
    sess = M.SlabSession(ctx, cfg, n, episodes_cap=episodes_cap,
                         engine=None if engine == "alternate" else engine, ring_exact=ring)
    acc = [dict(frames=[], bytes=b"", episodes=[], bits=[]) for _ in range(n)]
    for k in range(ncalls):
        new = []
        for i, x in enumerate(streams):
            edges = [0] + list(cuts_per_stream[i]) + [len(x)]
            new.append(x[edges[k]:edges[k + 1]])
        if engine == "alternate":
            sess.engine = ("workgroup", "wave")[k & 1]
        res = sess.feed(new, final=(k == ncalls - 1))
        for i in range(n):
            assert int(res["status"][i]) == 0
            nf, nb, ne = int(res["nframes"][i]), int(res["nbytes"][i]), int(res["nepisodes"][i])
            acc[i]["frames"].append(res["frames"][i, :nf].copy())
            acc[i]["bits"].append(res["bits"][i, :nf].copy())
            acc[i]["bytes"] += res["bytes"][i, :nb].tobytes()
            acc[i]["episodes"].append(res["episodes"][i, :ne].copy())
    for a in acc:
        a["frames"] = np.concatenate(a["frames"]) if a["frames"] else np.zeros(0, M.FRAME_DTYPE)
        a["episodes"] = np.concatenate(a["episodes"]) if a["episodes"] else np.zeros(0, M.EPISODE_DTYPE)
    assert all(int(f) & M.STATE_FINISHED for f in res["state"]["flags"])
    return acc


@pytest.mark.parametrize("engine", ["wave", "workgroup", "alternate"])
@pytest.mark.parametrize("name", G.names())
def test_any_golden_fed_in_three_slabs_equals_one_shot_and_oracle(gpu, name, engine):
    M, torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    x = g["samples"]
    if len(x) > 2000000:
        pytest.skip("0.5 baud: one samplebuf is longer than the recording's slabs")
    if engine != "wave" and cfg.auto_carrier_threshold > 0:
        pytest.skip("--auto-carrier runs on the wavefront engine")
    rng = np.random.default_rng(len(x))
    ref = O.oracle_rx_stream(ocfg, x)
    for trial in range(3):
        cuts = sorted(int(c) for c in rng.integers(0, len(x) + 1, size=3))
        if trial == 2:
            cuts = [1, 2, len(x) // 2]                    # slabs far shorter than a samplebuf
        got = _feed_in_slabs(M, ctx, cfg, [x], [cuts], engine=engine)[0]
        assert got["frames"].tobytes() == ref["frames"].tobytes(), (name, cuts)
        assert got["bytes"] == ref["bytes"], (name, cuts)
        assert got["episodes"].tobytes() == ref["episodes"].tobytes(), (name, cuts)


@pytest.mark.parametrize("name", G.names())
def test_any_golden_fed_in_slabs_prints_what_the_reference_prints(gpu, name):
    """VERDICT r3 item 6.  The reference feeds its loop through one samplebuf whose stale cells a
    search may read (minimodem.c:1150-1156).  (a) RING addressing for slabs
    (mifsk_demod_slab_ring: the cells persist in device memory between the calls): any cut
    equals the oracle's cell-for-cell replica of that buffer, frame for frame -- no golden
    exempted, t50_auto_rtty_lead (whose searches read stale cells in mid stream) included.
    (b) Under FLAT addressing (mifsk_demod_slab, both engines) the frames may differ from the
    reference's in the last digits of a confidence where a search saw a stale cell -- and
    still every golden, fed in slabs, prints exactly what `minimodem --rx --file` printed:
    stdout and the CARRIER / NOCARRIER lines with their three-decimal statistics."""
    M, torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    x = g["samples"]
    if len(x) > 2000000:
        pytest.skip("0.5 baud: one samplebuf is longer than the recording's slabs")
    rng = np.random.default_rng(len(x) + 1)
    ring_ref = O.oracle_rx_stream(ocfg, x, ring_mode=True)

    def text_of(got):
        out, err = M.stream_text(cfg, np.concatenate(got["bits"]) if got["bits"] else np.zeros(0, np.int64),
                                 got["episodes"], print_filter="--print-filter" in g["rx_args"],
                                 b_mark=None)
        el = [l for l in err.splitlines() if l]
        return out, [l for l in el if l.startswith("### CARRIER")], [l for l in el if l.startswith("### NOCARRIER")]

    for trial in range(2):
        cuts = sorted(int(c) for c in rng.integers(0, len(x) + 1, size=4))
        if trial == 1:
            cuts = [3, len(x) // 3, len(x) // 3 + 1, 2 * len(x) // 3]
        got = _feed_in_slabs(M, ctx, cfg, [x], [cuts], engine="wave", ring=True)[0]
        assert got["frames"].tobytes() == ring_ref["frames"].tobytes(), (name, cuts)
        assert got["episodes"].tobytes() == ring_ref["episodes"].tobytes(), (name, cuts)
        assert got["bytes"] == ring_ref["bytes"], (name, cuts)
        for engine in ("wave",) if cfg.auto_carrier_threshold > 0 else ("wave", "workgroup"):
            flat = _feed_in_slabs(M, ctx, cfg, [x], [cuts], engine=engine)[0]
            out, car, noc = text_of(flat)
            assert out == g["stdout"], (name, engine, cuts)
            assert car == g["carrier"] and noc == g["nocarrier"], (name, engine, cuts)
        out, car, noc = text_of(got)
        assert out == g["stdout"] and car == g["carrier"] and noc == g["nocarrier"], (name, "ring", cuts)


@pytest.mark.parametrize("mode,kw", [("1200", {}), ("300", {}), ("12000", {}), ("same", {}), ("rtty", {}),
                                     ("1200", dict(auto_carrier_threshold=0.001)), ("uic-ground", {}),
                                     ("5", {})],
                         ids=["1200", "300", "12000", "same", "rtty", "1200-auto", "uic", "5baud"])
@pytest.mark.parametrize("engine", ["wave", "workgroup", "alternate"])
def test_batch_of_streams_fed_in_many_ragged_slabs(gpu, mode, kw, engine):
    """A batch whose streams are cut at different places, seven slabs each (some empty, some a
    few samples): noisy, with gaps between bursts (episodes that end inside one slab and are
    reported by a later call), ragged lengths."""
    M, torch, ctx = gpu
    if engine != "wave" and kw:
        pytest.skip("--auto-carrier runs on the wavefront engine")
    cfg = M.rx_config(mode, **kw)
    ocfg = O.oracle_config(mode, **kw)
    rng = np.random.default_rng(808)
    five = cfg.n_data_bits == 5
    streams = []
    for i in range(3 if mode == "5" else 10):
        parts = []
        for b in range(1 + i % 3):
            if mode.startswith("uic"):
                import test_gpu_parity as T
                y, _ = T._uic_stream(M, cfg, rng, 6)
            else:
                # (5 baud: 9600-sample bit windows -- the workgroup engine's instantiation without
                # an LDS slab, demod_kernel<false, 0, 3, true>)
                nw = {"rtty": 6, "300": 16, "12000": 150, "5": 1}.get(mode, 50)
                words = rng.integers(0 if five else 32, 32 if five else 127, size=nw + i, dtype=np.uint8)
                y = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 80)),
                                 amplitude=float(rng.uniform(0.3, 1.0)))
            parts.append(y)
            parts.append(np.zeros(int(rng.integers(0, (1 if mode == "5" else 3) * int(cfg.samplebuf_size))), np.float32))
        x = np.concatenate(parts).astype(np.float32)
        if i % 2:
            x = (x + rng.normal(0, 0.06, x.shape)).astype(np.float32)
        streams.append(x)
    streams.append(rng.normal(0, 0.3, 40000).astype(np.float32))
    streams.append(np.zeros(0, np.float32))
    cuts = []
    for x in streams:
        c = sorted(int(v) for v in rng.integers(0, len(x) + 1, size=6))
        cuts.append(c)
    got = _feed_in_slabs(M, ctx, cfg, streams, cuts, engine=engine)
    total = 0
    for i, x in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, x)
        assert got[i]["frames"].tobytes() == ref["frames"].tobytes(), (mode, i, cuts[i])
        assert got[i]["bytes"] == ref["bytes"], (mode, i)
        assert got[i]["episodes"].tobytes() == ref["episodes"].tobytes(), (mode, i)
        total += len(ref["frames"])
    assert total > (3 if mode == "5" else 100)


@pytest.mark.parametrize("engine", ["wave", "workgroup"])
def test_stream_longer_than_one_call_keeps_only_a_samplebuf_of_history(gpu, engine):
    """What the state is for: a long stream fed 50 000 samples at a time.  After every call the
    caller may drop everything before state.base -- never more than a samplebuf plus one frame
    behind the newest sample."""
    M, torch, ctx = gpu
    cfg = M.rx_config("1200")
    ocfg = O.oracle_config("1200")
    rng = np.random.default_rng(4)
    x = M.synthesize(cfg, rng.integers(32, 127, size=2000, dtype=np.uint8), leading_silence=777)
    x = (x + rng.normal(0, 0.03, x.shape)).astype(np.float32)
    sess = M.SlabSession(ctx, cfg, 1, engine=engine)
    frames = []
    fed = 0
    while fed < len(x):
        k = min(50000, len(x) - fed)
        res = sess.feed([x[fed:fed + k]], final=(fed + k == len(x)))
        fed += k
        frames.append(res["frames"][0, :int(res["nframes"][0])].copy())
        if fed < len(x):
            assert fed - int(res["state"]["base"][0]) <= int(cfg.samplebuf_size) + 2 * int(cfg.frame_nsamples)
            assert len(sess.tail[0]) == fed - int(res["state"]["base"][0])
    ref = O.oracle_rx_stream(ocfg, x)
    assert np.concatenate(frames).tobytes() == ref["frames"].tobytes()
