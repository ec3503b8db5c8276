"""GPU parity tests proper: the HIP path, called through the C ABI
(libmifsk.so), against the oracle restatement on the same inputs.

Bar: decoded bytes, frame bits, frame starts, flags and episode counters are
bit-exact; confidences and amplitudes are compared as raw f32 bit patterns too
(the kernels perform the oracle's operation sequence), with the documented
tolerance (rel 1e-5 / abs 1e-6) only as the fallback assertion message."""
import numpy as np
import pytest

import _golden as G
import _oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    import minimodem_amd as M
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    ctx = M.Context()
    assert "gfx950" in ctx.device_name
    yield M, torch, ctx
    ctx.close()


def _bits_equal_f32(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32),
                          np.asarray(b, np.float32).view(np.uint32))


# (engine, RING addressing): one wavefront per stream with the default flat addressing and with
# the reference's stale-cell buffer semantics, and the workgroup-per-stream engine
VARIANTS = [("wave", False), ("wave", True), ("workgroup", False)]
VARIANT_IDS = ["wave-flat", "wave-ring", "workgroup-flat"]


def run_gpu_streams(M, torch, ctx, cfg, streams, want=("bytes", "frames", "episodes", "bits"),
                    engine="wave", ring=False):
    """streams: list of float32 numpy arrays (ragged).  Returns host results."""
    n = len(streams)
    maxn = max([len(s) for s in streams] + [4])
    stride = (maxn + 3) & ~3
    host = np.zeros((n, stride), np.float32)
    lens = np.zeros(n, np.int32)
    for i, s in enumerate(streams):
        host[i, :len(s)] = s
        lens[i] = len(s)
    d = torch.from_numpy(host).cuda()
    dl = torch.from_numpy(lens).cuda()
    out = M.demod_batch(ctx, cfg, d, nsamples=dl, want=want, episodes_cap=16,
                        engine=None if engine == "wave" else engine, ring_exact=ring,
                        force_engine=True)
    torch.cuda.synchronize()
    return M.results_to_host(out)


def assert_stream_equal(res, i, ref, cfg_name=""):
    nf = int(res["nframes"][i])
    assert nf == len(ref["frames"]), (cfg_name, i, nf, len(ref["frames"]))
    got = res["frames"][i, :nf]
    exp = ref["frames"]
    for field in ("bits", "start", "flags"):
        assert np.array_equal(got[field], exp[field]), (cfg_name, i, field)
    for field in ("confidence", "amplitude"):
        if not _bits_equal_f32(got[field], exp[field]):
            np.testing.assert_allclose(got[field], exp[field], rtol=1e-5, atol=1e-6,
                                       err_msg="%s stream %d %s" % (cfg_name, i, field))
            raise AssertionError("%s stream %d: %s within tolerance but not bit-identical"
                                 % (cfg_name, i, field))
    nb = int(res["nbytes"][i])
    assert res["bytes"][i, :nb].tobytes() == ref["bytes"]
    assert np.array_equal(res["bits"][i, :nf], exp["bits"])
    ne = int(res["nepisodes"][i])
    assert ne == len(ref["episodes"])
    assert res["episodes"][i, :ne].tobytes() == ref["episodes"].tobytes()
    assert int(res["status"][i]) == 0


@pytest.mark.parametrize("variant", VARIANTS, ids=VARIANT_IDS)
@pytest.mark.parametrize("name", G.names())
def test_demod_matches_oracle_and_reference_on_goldens(gpu, name, variant):
    M, torch, ctx = gpu
    engine, ring = variant
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    if engine == "workgroup" and cfg.auto_carrier_threshold > 0:
        pytest.skip("--auto-carrier runs on the wavefront engine only")
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    res = run_gpu_streams(M, torch, ctx, cfg, [g["samples"]], engine=engine, ring=ring)
    # RING addressing is compared with the oracle's cell-for-cell replica of the reference's
    # buffer: nothing is exempted (t50_auto_rtty_lead reads stale cells in mid stream)
    ref = O.oracle_rx_stream(ocfg, g["samples"], ring_mode=ring)
    assert_stream_equal(res, 0, ref, name)
    # and, transitively, the reference program's own output for this input
    if cfg.decoder == 0:
        assert res["bytes"][0, :int(res["nbytes"][0])].tobytes() == G.raw_stdout(g)
    lines = [O.format_nocarrier(ocfg, e) for e in res["episodes"][0, :int(res["nepisodes"][0])]]
    assert lines == g["nocarrier"]
    # device frame bits + episodes -> host post-pass (mifsk_stream_text) == everything the
    # reference printed: stdout through its databits decoder (ascii, baudot, caller-ID,
    # binary, print filter) and the CARRIER / NOCARRIER lines on stderr
    if cfg.auto_carrier_threshold > 0:          # --auto-carrier: the band found, and the space band with it
        assert int(res["carrier_band"][0]) == ref["carrier_band"] >= 0
    out, err = M.stream_text(cfg, res["bits"][0, :int(res["nframes"][0])],
                             res["episodes"][0, :int(res["nepisodes"][0])],
                             print_filter="--print-filter" in g["rx_args"],
                             b_mark=int(res["carrier_band"][0]) if "carrier_band" in res else None)
    assert out == g["stdout"]
    elines = [l for l in err.splitlines() if l]
    assert [l for l in elines if l.startswith("### CARRIER")] == g["carrier"]
    assert [l for l in elines if l.startswith("### NOCARRIER")] == g["nocarrier"]


@pytest.mark.parametrize("variant", [("wave", False), ("workgroup", False)], ids=["wave-flat", "workgroup-flat"])
@pytest.mark.parametrize("name", G.names())
def test_frames_equal_oracle_when_no_episode_records_are_asked_for(gpu, name, variant):
    """Without episode records (and without saved state) the lattice replay leaves the episodes'
    running totals out (replay_scan_asm / _soft, totals == false): the frames -- bits, starts,
    flags, confidence and amplitude bit patterns -- and the bytes must not notice."""
    M, torch, ctx = gpu
    engine, ring = variant
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    if engine == "workgroup" and cfg.auto_carrier_threshold > 0:
        pytest.skip("--auto-carrier runs on the wavefront engine only")
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    x = g["samples"]
    streams = [x, x[:int(len(x) * 0.7)]] if len(x) < 2_000_000 else [x]
    res = run_gpu_streams(M, torch, ctx, cfg, streams, want=("bytes", "frames", "bits"), engine=engine, ring=ring)
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s)
        nf = int(res["nframes"][i])
        assert nf == len(ref["frames"]), (name, i)
        got = res["frames"][i, :nf]
        for field in ("bits", "start", "flags"):
            assert np.array_equal(got[field], ref["frames"][field]), (name, i, field)
        for field in ("confidence", "amplitude"):
            assert _bits_equal_f32(got[field], ref["frames"][field]), (name, i, field)
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"]
        assert np.array_equal(res["bits"][i, :nf], ref["frames"]["bits"])
        assert int(res["status"][i]) == 0


@pytest.mark.parametrize("name", G.names())
def test_find_frame_batch_matches_reference_trace(gpu, name):
    M, torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    x = g["samples"]
    tr = g["trace"]
    prob = np.zeros(len(tr), M.SEARCH_DTYPE)
    prob["sample_offset"] = tr["offset"]
    prob["navail"] = len(x) - tr["offset"].astype(np.int64)
    prob["try_first"] = tr["first"]
    prob["try_max"] = tr["max"]
    prob["try_step"] = tr["step"]
    prob["search_limit"] = tr["limit"]
    prob["use_sync_string"] = tr["use_sync"]
    r = M.find_frame_batch(ctx, cfg, torch.from_numpy(x).cuda(), prob)
    # reference (fsk.c + FFT shim): integers exact, magnitudes to tolerance
    assert np.array_equal(r["bits"], tr["bits"])
    assert np.array_equal(r["frame_start"], tr["start"])
    fin = np.isfinite(tr["confidence"])
    assert np.array_equal(np.isinf(r["confidence"]), ~fin)
    np.testing.assert_allclose(r["confidence"][fin], tr["confidence"][fin], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["amplitude"], tr["amplitude"], rtol=1e-5, atol=1e-6)
    # oracle: bit for bit
    lib = O.oracle_lib()
    plan = lib.ofsk_plan_new(float(ocfg.sample_rate), ocfg.mark_f, ocfg.space_f, ocfg.band_width)
    pad = np.zeros(int(ocfg.expect_nsamples) + 2 * int(ocfg.try_max[0]) + 64, np.float32)
    xp = np.concatenate([x, pad])
    for row, got in zip(tr, r):
        expect = ocfg.expect_sync if row["use_sync"] else ocfg.expect_data
        conf, bits, ampl, start = O.oracle_find_frame(
            plan, xp[int(row["offset"]):], int(ocfg.expect_nsamples), int(row["first"]),
            int(row["max"]), int(row["step"]), float(row["limit"]), expect)
        assert (bits, start) == (int(got["bits"]), int(got["frame_start"]))
        assert _bits_equal_f32([conf, ampl], [got["confidence"], got["amplitude"]])
        assert int(got["n_positions"]) == lib.ofsk_last_n_positions()
    lib.ofsk_plan_destroy(plan)


def test_legacy_fsk_api_through_c_abi(gpu):
    """fsk_plan_new / fsk_find_frame / fsk_detect_carrier / fsk_set_tones_by_bandshift
    with host buffers, exactly as src/minimodem.c calls them."""
    M, torch, ctx = gpu
    g = G.load("t01_1200")
    ocfg = O.oracle_config("1200")
    plan = M.LegacyPlan(48000.0, ocfg.mark_f, ocfg.space_f, ocfg.band_width)
    assert (plan.fftsize, plan.nbands, plan.b_mark, plan.b_space) == (240, 121, 6, 11)
    lib = O.oracle_lib()
    op = lib.ofsk_plan_new(48000.0, ocfg.mark_f, ocfg.space_f, ocfg.band_width)
    x = np.concatenate([g["samples"], np.zeros(2000, np.float32)])
    rng = np.random.default_rng(5)
    for off in rng.integers(0, len(g["samples"]) - 1, size=40):
        for args in ((0, 60, 20, 2.3, "10dddddddd1"), (20, 50, 6, float("inf"), "10dddddddd1"),
                     (20, 50, 16, 2.3, "10dddddddd1"), (3, 17, 1, 1e9, "1dddddddddd")):
            got = plan.find_frame(x[off:], 440, *args)
            exp = O.oracle_find_frame(op, x[off:], 440, *args)
            assert got[1] == exp[1] and got[3] == exp[3]
            assert _bits_equal_f32([got[0], got[2]], [exp[0], exp[2]])
    # carrier autodetect
    for off in (0, 1000, 5000, 20000):
        win = x[off:off + 40]
        assert plan.detect_carrier(win, 0.001) == lib.ofsk_detect_carrier(
            op, np.ascontiguousarray(win).ctypes.data, 40, 0.001)
    assert plan.detect_carrier(np.zeros(40, np.float32), 0.001) == -1
    plan.set_tones_by_bandshift(8, -3)
    assert (plan.b_mark, plan.b_space) == (8, 5)
    assert plan.f_mark == pytest.approx(1600.0) and plan.f_space == pytest.approx(1000.0)
    # invalid plan: same error behaviour as the reference
    with pytest.raises(OSError):
        M.LegacyPlan(48000.0, 30000.0, 2200.0, 200.0)
    lib.ofsk_plan_destroy(op)
    plan.close()


MODES = ["1200", "300", "12000", "same", "rtty"]


@pytest.mark.parametrize("variant", VARIANTS, ids=VARIANT_IDS)
@pytest.mark.parametrize("mode", MODES)
def test_random_batch_with_noise_matches_oracle(gpu, mode, variant):
    """Seeded synthetic batch: ragged lengths, leading silence, additive noise at
    several SNRs, an all-noise stream, an empty and a too-short stream."""
    M, torch, ctx = gpu
    engine, ring = variant
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    rng = np.random.default_rng(1234)
    nwords = {"1200": 60, "300": 24, "12000": 200, "same": 40, "rtty": 6}[mode]
    hi = 32 if mode == "rtty" else 127
    streams = []
    for i in range(12):
        words = rng.integers(0 if mode == "rtty" else 32, hi, size=nwords + i, dtype=np.uint8)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 60)),
                         amplitude=float(rng.uniform(0.2, 1.0)))
        sigma = [0.0, 0.02, 0.1, 0.25][i % 4]
        if sigma:
            x = (x + rng.normal(0, sigma, x.shape)).astype(np.float32)
        if i == 5:
            x = x[: len(x) // 2 + 7]            # truncated mid-frame
        streams.append(x)
    streams.append(rng.normal(0, 0.3, 20000).astype(np.float32))     # noise only
    streams.append(np.zeros(0, np.float32))                          # empty
    streams.append(np.ones(int(ocfg.expect_nsamples) - 1, np.float32))   # too short
    streams.append(np.zeros(5000, np.float32))                       # silence
    res = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine, ring=ring)
    total = 0
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        assert_stream_equal(res, i, ref, mode)
        total += len(ref["frames"])
    assert total > 10 * nwords


@pytest.mark.parametrize("mode,kw", [("rtty", {}), ("rtty", dict(sample_rate=44100)), ("50", {}), ("30", {}),
                                     ("20", {}), ("rtty", dict(sample_rate=96000)),
                                     ("75", dict(sample_rate=22050)), ("110", dict(sample_rate=96000)),
                                     ("60", dict(sample_rate=44100)), ("100", dict(sample_rate=44100)), ("25", {})],
                         ids=["B1056", "B970", "B960", "B1600", "B2400", "B2112", "B294", "B873", "B735", "B441", "B1920"])
def test_long_windows_through_the_tile(gpu, mode, kw):
    """Bit windows too long for a search's span to sit in LDS are read from global memory
    through the 64-sample tile (wave engine, demod_wave_kernel<10, -1>): bit lengths that are
    a whole number of steps (960, 1600), that end in a short step (1056, 2112, 2400) and in a
    short group (970), with noise, ragged ends and a stream that ends inside a window."""
    M, torch, ctx = gpu
    cfg = M.rx_config(mode, **kw)
    ocfg = O.oracle_config(mode, **kw)
    kernel = M.demod_plan(ctx, cfg, 8, engine="wave")["kernel"]
    # (the shorter of these fit a search slab in LDS at this batch size: whichever instantiation
    # the planner picks, the frames are the oracle's)
    assert "demod_wave_kernel<10, -1>" in kernel or int(cfg.bit_nsamples) < 900, kernel
    rng = np.random.default_rng(77)
    five = cfg.n_data_bits == 5
    streams = []
    for i in range(6):
        words = rng.integers(0 if five else 32, 32 if five else 127, size=5 + i, dtype=np.uint8)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 200)),
                         amplitude=float(rng.uniform(0.3, 1.0)))
        sigma = [0.0, 0.05, 0.2][i % 3]
        if sigma:
            x = (x + rng.normal(0, sigma, x.shape)).astype(np.float32)
        if i == 4:
            x = x[: len(x) - int(cfg.bit_nsamples) // 3]      # ends inside a bit window
        streams.append(x)
    streams.append(rng.normal(0, 0.3, 6 * int(ocfg.expect_nsamples)).astype(np.float32))
    res = run_gpu_streams(M, torch, ctx, cfg, streams, engine="wave", ring=False)
    total = 0
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s)
        assert_stream_equal(res, i, ref, mode)
        total += len(ref["frames"])
    assert total > 20


@pytest.mark.parametrize("mode,tones", [
    ("300", [(1270, 1070), (1570, 1370), (2025, 1825), (980, 780), (3000, 2800)]),
    ("1200", [(1200, 2200), (1500, 2300), (2000, 2800), (1000, 1800), (2600, 3400)]),
])
@pytest.mark.parametrize("ring", [False, True], ids=["flat", "ring"])
def test_auto_carrier_batch_with_different_tones_per_stream(gpu, mode, tones, ring):
    """--auto-carrier over a batch whose streams use different tone pairs, behind
    different amounts of leading silence and noise; plus streams in which no
    carrier is ever found.  Band, start cursor and everything decoded after
    them must equal the oracle's, stream by stream."""
    M, torch, ctx = gpu
    cfg = M.rx_config(mode, auto_carrier_threshold=0.001)
    ocfg = O.oracle_config(mode, auto_carrier_threshold=0.001)
    rng = np.random.default_rng(99)
    streams = []
    for i, (mark, space) in enumerate(tones):
        txcfg = M.rx_config(mode, mark_f=float(mark), space_f=float(space))
        words = rng.integers(32, 127, size=30 + i, dtype=np.uint8)
        x = M.synthesize(txcfg, words, leading_silence=int(rng.integers(0, 30000)),
                         amplitude=float(rng.uniform(0.3, 1.0)))
        if i % 2:
            x = (x + rng.normal(0, 0.0002, x.shape)).astype(np.float32)   # below the threshold
        streams.append(x)
    # the tone is looked for again after 21 searches without confidence (minimodem.c:1297):
    # noise ABOVE the threshold before the signal (a noise band is taken first), and two
    # bursts on different tone pairs in one stream
    for i, (mark, space) in enumerate(tones[:3]):
        txcfg = M.rx_config(mode, mark_f=float(mark), space_f=float(space))
        words = rng.integers(32, 127, size=25 + i, dtype=np.uint8)
        x = M.synthesize(txcfg, words, amplitude=float(rng.uniform(0.4, 1.0)))
        lead = rng.normal(0, 0.01, int(rng.integers(3000, 20000))).astype(np.float32)
        m2, s2 = tones[(i + 2) % len(tones)]
        tx2 = M.rx_config(mode, mark_f=float(m2), space_f=float(s2))
        y = M.synthesize(tx2, rng.integers(32, 127, size=12, dtype=np.uint8), amplitude=0.8)
        gap = np.zeros(int(rng.integers(15000, 30000)), np.float32)
        streams.append(np.concatenate([lead, x, gap, y]).astype(np.float32))
    streams.append(rng.normal(0, 0.02, 60000).astype(np.float32))       # noise above it, nothing else
    streams.append(np.zeros(40000, np.float32))                          # silence: never found
    streams.append(rng.normal(0, 0.0001, 30000).astype(np.float32))      # noise below the threshold
    streams.append(np.zeros(0, np.float32))
    streams.append(np.zeros(17, np.float32))                             # shorter than a scan window
    res = run_gpu_streams(M, torch, ctx, cfg, streams, ring=ring)
    found = 0
    bands_seen = set()
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        assert int(res["carrier_band"][i]) == ref["carrier_band"], (mode, i)
        assert_stream_equal(res, i, ref, mode)
        found += ref["carrier_band"] >= 0
        bands_seen.add(tuple(int(b) for b in ref["episodes"]["b_mark"]))
    assert found >= len(tones) + 3
    assert any(len(set(b)) > 1 for b in bands_seen)      # some stream changed tones between episodes


@pytest.mark.parametrize("mode,kw", [("1200", {}), ("same", {}),
                                     ("300", dict(auto_carrier_threshold=0.001))])
@pytest.mark.parametrize("ring", [False, True], ids=["flat", "ring"])
def test_demod_batch_host_entry_point(gpu, mode, kw, ring):
    """mifsk_demod_batch_host: host pointers in, host results out (ragged rows whose
    stride is not a multiple of 4, --auto-carrier bands included)."""
    M, torch, ctx = gpu
    cfg = M.rx_config(mode, **kw)
    ocfg = O.oracle_config(mode, **kw)
    rng = np.random.default_rng(21)
    streams = []
    for i in range(7):
        words = rng.integers(32, 127, size=20 + 3 * i, dtype=np.uint8)
        streams.append(M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 500)),
                                    amplitude=float(rng.uniform(0.3, 1.0))))
    streams.append(np.zeros(0, np.float32))
    stride = max(len(s) for s in streams) + 1          # odd on purpose
    host = np.zeros((len(streams), stride), np.float32)
    lens = np.zeros(len(streams), np.uint32)
    for i, s in enumerate(streams):
        host[i, :len(s)] = s
        lens[i] = len(s)
    res = M.demod_batch_host(ctx, cfg, host, lens, episodes_cap=16, ring_exact=ring)
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        assert_stream_equal(res, i, ref, mode)
        if kw:
            assert int(res["carrier_band"][i]) == ref["carrier_band"]


@pytest.mark.parametrize("engine", ["wave", "workgroup"])
def test_output_capacity_overflow_is_flagged(gpu, engine):
    M, torch, ctx = gpu
    cfg = M.rx_config("1200")
    x = M.synthesize(cfg, b"hello world, this is more than eight frames")
    d = torch.from_numpy(np.pad(x, (0, (-len(x)) % 4))[None, :]).cuda()
    dl = torch.tensor([len(x)], dtype=torch.int32).cuda()
    out = M.demod_batch(ctx, cfg, d, nsamples=dl, want=("bytes", "episodes"), frames_cap=8,
                        episodes_cap=1, engine=engine)
    torch.cuda.synchronize()
    r = M.results_to_host(out)
    assert int(r["nframes"][0]) == 43 and int(r["nbytes"][0]) == 43
    assert int(r["status"][0]) & 1
    assert r["bytes"][0, :8].tobytes() == b"hello wo"


def test_multi_device_entry_point_shards_and_gathers(gpu):
    """mifsk_demod_batch_host_multi: several device contexts in ONE process, streams sharded
    by mifsk_shard_range, every output row written back by the device that owns the stream.
    (One GPU is visible here: the contexts share it, the code path is the multi-GPU one.)"""
    import ctypes as C
    M, torch, ctx = gpu
    lib = M._lib.load()
    lo, hi = C.c_int(), C.c_int()
    covered = []
    for r in range(3):
        lib.mifsk_shard_range(11, r, 3, C.byref(lo), C.byref(hi))
        assert (lo.value, hi.value) == M.shard_range(11, r, 3)
        covered += list(range(lo.value, hi.value))
    assert covered == list(range(11))
    cfg = M.rx_config("1200")
    ocfg = O.oracle_config("1200")
    rng = np.random.default_rng(77)
    streams = [M.synthesize(cfg, rng.integers(32, 127, size=15 + 2 * i, dtype=np.uint8),
                            leading_silence=int(rng.integers(0, 300))) for i in range(11)]
    stride = (max(len(s) for s in streams) + 3) & ~3
    host = np.zeros((11, stride), np.float32)
    lens = np.zeros(11, np.uint32)
    for i, s in enumerate(streams):
        host[i, :len(s)] = s
        lens[i] = len(s)
    ctxs = [M.Context(0) for _ in range(3)]
    try:
        res = M.demod_batch_host(ctxs, cfg, host, lens, episodes_cap=8)
        one = M.demod_batch_host(ctx, cfg, host, lens, episodes_cap=8)
    finally:
        for c in ctxs:
            c.close()
    for i, s in enumerate(streams):
        assert_stream_equal(res, i, O.oracle_rx_stream(ocfg, s), "multi")
    for k in ("nbytes", "nframes", "nepisodes"):
        assert np.array_equal(res[k], one[k])
    for i in range(11):
        assert np.array_equal(res["bytes"][i, :res["nbytes"][i]], one["bytes"][i, :one["nbytes"][i]])


def test_demod_plan_reports_engine_and_lds(gpu):
    """mifsk_demod_plan: the kernel instantiation, engine and dynamic LDS mifsk_demod_batch uses
    (rocprofv3 does not report dynamic LDS; this is the occupancy evidence)."""
    M, torch, ctx = gpu
    p = M.demod_plan(ctx, M.rx_config("1200"), 1024)
    assert p["engine"] == "workgroup" and p["workgroup_size"] == 192 and "demod_kernel<true, 10, 2>" in p["kernel"]
    assert p["lds_bytes_per_workgroup"] <= 40960 and p["workgroups_per_cu"] == 4
    p = M.demod_plan(ctx, M.rx_config("1200"), 1024, engine="wave")
    assert p["engine"] == "wave" and "demod_wave_kernel<10, 10>" in p["kernel"] and p["lattice_mode"] == 1
    p = M.demod_plan(ctx, M.rx_config("12000"), 8192)
    assert p["engine"] == "wave" and "demod_wave_kernel<4, 1>" in p["kernel"]
    assert p["workgroups_per_cu"] >= 16 and p["frames_per_block"] >= 32
    p = M.demod_plan(ctx, M.rx_config("rtty"), 4096)
    assert p["engine"] == "wave" and p["lattice_mode"] == 0          # half the buffer < one search: no LATTICE
    p = M.demod_plan(ctx, M.rx_config("1200"), 1024, ring_exact=True)
    assert p["engine"] == "wave" and p["lattice_mode"] == 0
    p = M.demod_plan(ctx, M.rx_config("1200", auto_carrier_threshold=0.001), 64)
    assert p["engine"] == "wave"


def _uic_stream(M, cfg, rng, nframes, sigma=0.0, amplitude=1.0):
    """UIC-751-3 telegrams (src/minimodem.c:859-868: 47-bit frames, the 8-bit sync pattern
    11110010 in the expect string and 39 data bits, no start/stop bits).  The reference cannot
    transmit them (its TX frames start bits as zeros), so the bit stream is laid out here --
    mark idle, frames separated by 0..8 idle bits -- and keyed by the host transmitter as raw
    8-bit words without framing on the mode's tones."""
    tx = M.rx_config("600", mark_f=cfg.mark_f, space_f=cfg.space_f, n_data_bits=8, nstartbits=0,
                     nstopbits=0.0)
    bits = [1] * int(rng.integers(24, 64))
    words = []
    for _ in range(nframes):
        d = rng.integers(0, 2, size=39).tolist()
        words.append(sum(b << i for i, b in enumerate(d)))
        bits += [1, 1, 1, 1, 0, 0, 1, 0] + d + [1] * int(rng.integers(0, 9))
    bits += [1] * 16
    bits += [1] * (-len(bits) % 8)
    x = M.synthesize(tx, np.packbits(np.array(bits, np.uint8), bitorder="little"),
                     amplitude=amplitude, leading_silence=int(rng.integers(0, 300)))
    if sigma:
        x = (x + rng.normal(0, sigma, x.shape)).astype(np.float32)
    return x, words


@pytest.mark.parametrize("variant", VARIANTS, ids=VARIANT_IDS)
@pytest.mark.parametrize("mode", ["uic-ground", "uic-train"])
def test_uic_47_bit_frames_match_oracle(gpu, mode, variant):
    """Frames longer than 32 bits on the device: the generic frame_confidence (64-bit frame
    word, 47 magnitudes per candidate) and SCAN batches of 47 windows per candidate, tone +
    noise input, both engines and RING addressing."""
    M, torch, ctx = gpu
    engine, ring = variant
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    assert cfg.expect_n_bits == 47 and cfg.n_data_bits == 39
    rng = np.random.default_rng(751)
    streams, sent = [], []
    for i in range(6):
        x, words = _uic_stream(M, cfg, rng, 10 + i, sigma=[0.0, 0.03, 0.12][i % 3],
                               amplitude=float(rng.uniform(0.3, 1.0)))
        if i == 4:
            x = x[: len(x) - 1500]                    # ends inside a frame
        streams.append(x)
        sent.append(words)
    streams.append(rng.normal(0, 0.3, 30000).astype(np.float32))
    res = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine, ring=ring)
    total = 0
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        assert_stream_equal(res, i, ref, mode)
        total += len(ref["frames"])
    # the clean streams decode (nearly all of) what was sent, all 39 data bits of a telegram
    # (nearly: with no idle bit between two telegrams the reference's own search sometimes
    # settles a sample early and reads a neighbouring bit -- the oracle does the same)
    for i in (0, 3):
        nf = int(res["nframes"][i])
        got = set(int(b) for b in res["bits"][i, :nf])
        assert sum(1 for w in sent[i] if w in got) >= len(sent[i]) - 2, (i, len(sent[i]))
    assert any(w >> 32 for w in sent[0])              # (bits above 32 really were exercised)
    assert total >= 50
    # ... and through the host post-pass (the UIC decoders print one line per telegram)
    out, _err = M.stream_text(cfg, res["bits"][0, :int(res["nframes"][0])],
                              res["episodes"][0, :int(res["nepisodes"][0])], quiet=True)
    assert out.count(b"\n") >= len(sent[0])


@pytest.mark.parametrize("mode", ["uic-ground"])
def test_find_frame_batch_with_47_bit_expect_string(gpu, mode):
    """mifsk_find_frame_batch (the legacy fsk_find_frame, N problems) with the 47-bit expect
    string: every candidate is 47 windows, a chunk of candidates fills W_CAP."""
    M, torch, ctx = gpu
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    rng = np.random.default_rng(7513)
    x, _ = _uic_stream(M, cfg, rng, 8, sigma=0.05)
    offs = rng.integers(0, len(x) - int(cfg.expect_nsamples) - 200, size=48)
    geos = [(0, 120, 40, 2.3), (40, 100, 33, 2.3), (40, 100, 12, float("inf")), (0, 120, 15, float("inf")),
            (0, 200, 3, float("inf"))]               # the last one: 67 candidates = two chunks
    prob = np.zeros(len(offs) * len(geos), M.SEARCH_DTYPE)
    k = 0
    for off in offs:
        for first, tmax, step, limit in geos:
            prob[k] = (int(off), len(x) - int(off), first, tmax, step, limit, 0)
            k += 1
    r = M.find_frame_batch(ctx, cfg, torch.from_numpy(x).cuda(), prob)
    lib = O.oracle_lib()
    plan = lib.ofsk_plan_new(float(ocfg.sample_rate), ocfg.mark_f, ocfg.space_f, ocfg.band_width)
    xp = np.concatenate([x, np.zeros(int(ocfg.expect_nsamples) + 400, np.float32)])
    hits = 0
    for row, got in zip(prob, r):
        conf, bits, ampl, start = O.oracle_find_frame(
            plan, xp[int(row["sample_offset"]):], int(ocfg.expect_nsamples), int(row["try_first"]),
            int(row["try_max"]), int(row["try_step"]), float(row["search_limit"]), ocfg.expect_data)
        assert (bits, start) == (int(got["bits"]), int(got["frame_start"]))
        assert _bits_equal_f32([conf, ampl], [got["confidence"], got["amplitude"]])
        assert int(got["n_positions"]) == lib.ofsk_last_n_positions()
        hits += conf > 0
    lib.ofsk_plan_destroy(plan)
    assert hits >= 5


def _same_or_both_nan(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return np.array_equal(np.isnan(a), np.isnan(b)) and \
        np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)])


@pytest.mark.parametrize("variant", VARIANTS, ids=VARIANT_IDS)
@pytest.mark.parametrize("mode", ["1200", "2400", "12000", "same", "rtty", "300"])
def test_non_finite_samples_give_the_oracles_frames(gpu, mode, variant):
    """NaN / +-Inf samples are DEFINED behaviour in the reference: a bit window holding one
    has a NaN magnitude, a candidate with a NaN confidence never wins a scan (`best < c` is
    false, fsk.c:492) and every compare in the loop sees IEEE semantics (fsk.c:292,336;
    minimodem.c:1278-1292).  The device must produce the oracle's frames around the bad
    samples -- the same frame records and episode totals, NaN for NaN (payload bits of a NaN
    are the one thing not compared: x86 and gfx950 generate different default NaNs)."""
    M, torch, ctx = gpu
    engine, ring = variant
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    rng = np.random.default_rng(292)
    five = cfg.n_data_bits == 5
    streams = []
    for i in range(8):
        nw = {"rtty": 8, "300": 24, "12000": 300}.get(mode, 80)
        words = rng.integers(0 if five else 32, 32 if five else 127, size=nw + i, dtype=np.uint8)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 50)),
                         amplitude=float(rng.uniform(0.4, 1.0)))
        if i % 2:
            x = (x + rng.normal(0, 0.05, x.shape)).astype(np.float32)
        x = x.copy()
        nbad = [1, 3, 12, 40, 2, 5, 1, 9][i]
        pos = rng.integers(len(x) // 8, len(x) - 4, size=nbad)
        kinds = [np.nan, np.inf, -np.inf]
        for j, p in enumerate(pos):
            if i == 3:
                x[p:p + int(cfg.bit_nsamples) * 3] = kinds[j % 3]      # whole windows of them
            else:
                x[p] = kinds[(i + j) % 3]
        if i == 6:
            x[0] = np.nan                                              # ... in the leading silence
        streams.append(x)
    res = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine, ring=ring)
    total = 0
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        nf = int(res["nframes"][i])
        assert nf == len(ref["frames"]), (mode, i, nf, len(ref["frames"]))
        got, exp = res["frames"][i, :nf], ref["frames"]
        for field in ("bits", "start", "flags"):
            assert np.array_equal(got[field], exp[field]), (mode, i, field)
        for field in ("confidence", "amplitude"):
            assert _same_or_both_nan(got[field], exp[field]), (mode, i, field)
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"]
        ne = int(res["nepisodes"][i])
        assert ne == len(ref["episodes"])
        ge, ee = res["episodes"][i, :ne], ref["episodes"]
        for field in ("carrier_nsamples", "first_frame", "nframes", "end_reason", "b_mark"):
            assert np.array_equal(ge[field], ee[field]), (mode, i, field)
        for field in ("confidence_total", "amplitude_total"):
            assert _same_or_both_nan(ge[field], ee[field]), (mode, i, field)
        assert int(res["status"][i]) == 0
        total += nf
    assert total > (40 if mode == "rtty" else 100)


@pytest.mark.parametrize("variant", VARIANTS, ids=VARIANT_IDS)
@pytest.mark.parametrize("mode", ["1200", "12000", "same", "rtty"])
def test_subnormal_scale_audio_takes_the_division_fallback_and_matches_the_oracle(gpu, mode, variant):
    """Audio at amplitudes of 1e-38 ... 1e-44 (float subnormals): magnitudes, class means and
    `avg_sig` are subnormal, the noise bin is below FLT_EPSILON (so confidences are infinite or
    NaN, fsk.c:279,292) -- and the fast quotients of the confidence pass (div_by_rcp: exact
    only for normal quotients) must hand over to the divisions proper.  The frames must be
    the oracle's bit for bit, and the work counter MIFSK_CNT_CONF_FALLBACKS must show that
    the fallback arm of frame_confidence_fixed is what ran."""
    M, torch, ctx = gpu
    engine, ring = variant
    cfg = M.rx_config(mode)
    ocfg = O.oracle_config(mode)
    rng = np.random.default_rng(1938)
    five = cfg.n_data_bits == 5
    amps = [1e-38, 3e-39, 1e-40, 1e-42, 1e-44, 1.0, 9e-39]
    streams = []
    for i, a in enumerate(amps):
        nw = {"rtty": 8, "12000": 300}.get(mode, 60)
        words = rng.integers(0 if five else 32, 32 if five else 127, size=nw + i, dtype=np.uint8)
        # (generated at amplitude 1 and scaled in double: what a recorder with that gain would store)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 50)), amplitude=1.0)
        x = (x.astype(np.float64) * a).astype(np.float32)
        if i == 6:
            # a normal-scale stretch in the middle: the loop crosses between the two arms
            x[len(x) // 3: 2 * len(x) // 3] = (x[len(x) // 3: 2 * len(x) // 3].astype(np.float64) * 1e38).astype(np.float32)
        streams.append(x)
    assert any(np.any((np.abs(s) > 0) & (np.abs(s) < np.finfo(np.float32).tiny)) for s in streams)
    res = run_gpu_streams(M, torch, ctx, cfg, streams, want=("bytes", "frames", "episodes", "bits", "counters"),
                          engine=engine, ring=ring)
    total = 0
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s, ring_mode=ring)
        nf = int(res["nframes"][i])
        assert nf == len(ref["frames"]), (mode, i, nf, len(ref["frames"]))
        got, exp = res["frames"][i, :nf], ref["frames"]
        for field in ("bits", "start", "flags"):
            assert np.array_equal(got[field], exp[field]), (mode, i, field)
        for field in ("confidence", "amplitude"):
            assert _same_or_both_nan(got[field], exp[field]), (mode, i, field)
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"]
        ne = int(res["nepisodes"][i])
        assert ne == len(ref["episodes"])
        for field in ("carrier_nsamples", "first_frame", "nframes", "end_reason", "b_mark"):
            assert np.array_equal(res["episodes"][i, :ne][field], ref["episodes"][field]), (mode, i, field)
        for field in ("confidence_total", "amplitude_total"):
            assert _same_or_both_nan(res["episodes"][i, :ne][field], ref["episodes"][field]), (mode, i, field)
        total += nf
    fb = res["counters"][:, 24]
    # every stream whose samples are subnormal and that produced a frame at all went through the
    # fallback (the amplitude-1 stream may have, too: a candidate lying in digital silence has
    # a class mean of exactly 0)
    took = [i for i in range(len(amps)) if i != 5 and int(res["nframes"][i]) > 0]
    assert took and all(int(fb[i]) > 0 for i in took), (fb, res["nframes"])
    assert total > 0
