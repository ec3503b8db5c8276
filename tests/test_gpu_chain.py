"""Chained launches (mifsk_launch_info.chain_groups / chain_chunks): a batch cut
into groups of streams x time chunks, every (group, chunk) one launch of the resumable kernel,
must give the frames, bytes and episodes of the single launch -- i.e. the oracle's -- whatever the
cut.  The cut is forced onto small batches here (MIFSK_EXPERIMENT + MIFSK_CHAIN = "G,K": the
library itself only cuts batches of more streams than the chip holds)."""
import numpy as np
import pytest

import _golden as G
import _oracle as O
from test_gpu_parity import assert_stream_equal, run_gpu_streams

pytestmark = pytest.mark.gpu

CUTS = ["2,3", "1,5", "3,7"]


@pytest.fixture(scope="module")
def gpu():
    import torch
    import minimodem_amd as M
    assert torch.cuda.is_available(), "these tests need a real MI355X"
    ctx = M.Context()
    yield M, torch, ctx
    ctx.close()


def _chained(M, ctx, cfg, n, nsamples, engine="wave"):
    p = M.demod_plan(ctx, cfg, n, engine=engine, nsamples=nsamples)
    return p["chain_groups"], p["chain_chunks"]


@pytest.mark.parametrize("engine", ["wave", "workgroup"])
@pytest.mark.parametrize("cut", CUTS)
@pytest.mark.parametrize("name", G.names())
def test_chained_launches_give_the_oracles_streams_on_goldens(gpu, monkeypatch, name, cut, engine):
    M, torch, ctx = gpu
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    x = g["samples"]
    if len(x) > 2_000_000:
        pytest.skip("one long stream: covered by the slab tests")
    if engine == "workgroup" and cfg.auto_carrier_threshold > 0:
        pytest.skip("--auto-carrier runs on the wavefront engine")
    streams = [x, x[:int(len(x) * 0.61)], x[int(len(x) * 0.13):], x]
    monkeypatch.setenv("MIFSK_EXPERIMENT", "1")
    monkeypatch.setenv("MIFSK_CHAIN", cut)
    groups, chunks = _chained(M, ctx, cfg, len(streams), (len(x) + 3) & ~3, engine)
    if not groups:
        # only where the mode's kernel instantiation has no resumable twin (its batches are never
        # cut): the wavefront engine's fixed-length ones; every workgroup instantiation has one
        assert engine == "wave"
        monkeypatch.setenv("MIFSK_CHAIN", "0,0")
        plain = M.demod_plan(ctx, cfg, len(streams), engine="wave", nsamples=(len(x) + 3) & ~3)["kernel"]
        assert any(t in plain for t in ("<10, 10>", "<10, 5>", "<4, 1>")), plain
        pytest.skip("no resumable twin of " + plain)
    assert (groups, chunks) == tuple(int(v) for v in cut.split(","))
    assert ", true>" in M.demod_plan(ctx, cfg, len(streams), engine=engine, nsamples=(len(x) + 3) & ~3)["kernel"]
    res = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine)
    ocfg = O.oracle_config(**g["cfg_kwargs"])
    for i, s in enumerate(streams):
        assert_stream_equal(res, i, O.oracle_rx_stream(ocfg, s), "%s cut %s" % (name, cut))


@pytest.mark.parametrize("engine", ["wave", "workgroup"])
@pytest.mark.parametrize("mode,opts", [("rtty", {}), ("300", {}), ("same", {}), ("110", {}), ("1200", {}), ("2400", {})])
def test_chained_equals_single_launch_on_noisy_ragged_batch(gpu, monkeypatch, mode, opts, engine):
    M, torch, ctx = gpu
    if engine == "workgroup" and mode in ("rtty", "same"):
        pytest.skip("covered on the wavefront engine (the workgroup engine's direct lattice is slow here)")
    cfg = M.rx_config(mode, **opts)
    ocfg = O.oracle_config(mode, **opts)
    rng = np.random.default_rng(77)
    streams = []
    for i in range(10):
        nwords = int(rng.integers(20, 60))
        five = cfg.n_data_bits == 5
        words = rng.integers(0 if five else 32, 32 if five else 127, size=nwords).astype(np.uint8)
        x = M.synthesize(cfg, words, amplitude=0.7, leading_silence=int(rng.integers(0, 3000)))
        x = np.concatenate([x, np.zeros(int(rng.integers(0, 5000)), np.float32)])
        x = x + rng.normal(0, [0.0, 0.05, 0.2, 0.5][i % 4], len(x)).astype(np.float32)
        streams.append(x.astype(np.float32))
    monkeypatch.setenv("MIFSK_EXPERIMENT", "1")
    monkeypatch.setenv("MIFSK_CHAIN", "0,0")
    single = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine)
    stride = (max(len(s) for s in streams) + 3) & ~3
    for cut in ("2,4", "3,9"):
        monkeypatch.setenv("MIFSK_CHAIN", cut)
        if not _chained(M, ctx, cfg, len(streams), stride, engine)[0]:
            pytest.skip("no resumable twin for this mode's instantiation")
        res = run_gpu_streams(M, torch, ctx, cfg, streams, engine=engine)
        for key in ("nframes", "nbytes", "nepisodes", "status"):
            assert np.array_equal(res[key], single[key]), (mode, cut, key)
        for i in range(len(streams)):
            nf, nb, ne = int(single["nframes"][i]), int(single["nbytes"][i]), int(single["nepisodes"][i])
            assert res["frames"][i, :nf].tobytes() == single["frames"][i, :nf].tobytes(), (mode, cut, i)
            assert res["bytes"][i, :nb].tobytes() == single["bytes"][i, :nb].tobytes()
            assert res["episodes"][i, :ne].tobytes() == single["episodes"][i, :ne].tobytes()
    for i in (0, 3, 7):
        assert_stream_equal(res, i, O.oracle_rx_stream(ocfg, streams[i]), mode)


def test_chained_without_episode_records(gpu, monkeypatch):
    """... and with no episode records asked for (the replay then leaves the running totals out,
    in every chunk): frames and bytes are the single launch's."""
    M, torch, ctx = gpu
    cfg = M.rx_config("same")
    ocfg = O.oracle_config("same")
    rng = np.random.default_rng(5)
    streams = []
    for i in range(8):
        words = rng.integers(32, 127, size=int(rng.integers(30, 80))).astype(np.uint8)
        x = M.synthesize(cfg, np.concatenate([np.full(16, 0xAB, np.uint8), words]), amplitude=0.6)
        x = x + rng.normal(0, [0.0, 0.03, 0.1, 0.2][i % 4], len(x)).astype(np.float32)
        streams.append(x.astype(np.float32))
    monkeypatch.setenv("MIFSK_EXPERIMENT", "1")
    monkeypatch.setenv("MIFSK_CHAIN", "2,5")
    stride = (max(len(s) for s in streams) + 3) & ~3
    assert _chained(M, ctx, cfg, len(streams), stride) == (2, 5)
    res = run_gpu_streams(M, torch, ctx, cfg, streams, want=("bytes", "frames"), engine="wave")
    for i, s in enumerate(streams):
        ref = O.oracle_rx_stream(ocfg, s)
        nf = int(res["nframes"][i])
        assert nf == len(ref["frames"])
        assert res["frames"][i, :nf].tobytes() == ref["frames"].tobytes(), i
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"]


def test_chained_edge_cases_equal_single_launch(gpu, monkeypatch):
    """Empty rows, rows shorter than a chunk, fewer streams than groups, and output arrays too
    small for the frames (the truncation flag and the counts must be the single launch's)."""
    M, torch, ctx = gpu
    cfg = M.rx_config("300")
    rng = np.random.default_rng(11)
    full = M.synthesize(cfg, rng.integers(32, 127, size=60).astype(np.uint8), amplitude=0.8)
    streams = [full, np.zeros(0, np.float32), full[:900], full[:len(full) // 3], np.zeros(5000, np.float32), full]
    monkeypatch.setenv("MIFSK_EXPERIMENT", "1")

    def run(cut, which, frames_cap=None):
        monkeypatch.setenv("MIFSK_CHAIN", cut)
        sel = [streams[i] for i in which]
        n = len(sel)
        stride = (max([len(s) for s in sel] + [4]) + 3) & ~3
        host = np.zeros((n, stride), np.float32)
        lens = np.zeros(n, np.int32)
        for i, x in enumerate(sel):
            host[i, :len(x)] = x
            lens[i] = len(x)
        out = M.demod_batch(ctx, cfg, torch.from_numpy(host).cuda(), nsamples=torch.from_numpy(lens).cuda(),
                            want=("bytes", "frames", "episodes", "bits"), episodes_cap=16, frames_cap=frames_cap,
                            force_engine=True)
        torch.cuda.synchronize()
        return M.results_to_host(out)

    for which, cap in ((range(6), None), (range(6), 7), ([0], None), ([0, 2], 3)):
        which = list(which)
        single = run("0,0", which, cap)
        for cut in ("2,4", "3,9", "1,2"):
            res = run(cut, which, cap)
            for key in ("nframes", "nbytes", "nepisodes", "status"):
                assert np.array_equal(res[key], single[key]), (which, cap, cut, key)
            for i in range(len(which)):
                nf = min(int(single["nframes"][i]), single["frames"].shape[1])
                nb = min(int(single["nbytes"][i]), single["bytes"].shape[1])
                ne = min(int(single["nepisodes"][i]), single["episodes"].shape[1])
                assert res["frames"][i, :nf].tobytes() == single["frames"][i, :nf].tobytes(), (which, cap, cut, i)
                assert res["bytes"][i, :nb].tobytes() == single["bytes"][i, :nb].tobytes()
                assert res["episodes"][i, :ne].tobytes() == single["episodes"][i, :ne].tobytes()
    assert int(run("2,4", list(range(6)), 7)["status"][0]) != 0		# (the full stream does not fit 7 frames)


def test_which_batches_are_cut(gpu):
    """The library's own rule: flat wavefront-engine batches of more streams than the chip holds
    at once, streams long enough to cut, an instantiation with a resumable twin."""
    M, torch, ctx = gpu
    rtty = M.rx_config("rtty")
    n = int(30 * rtty.sample_rate)
    p = M.demod_plan(ctx, rtty, 4096, nsamples=n)
    assert (p["chain_groups"], p["chain_chunks"]) == (2, 8) and p["kernel"].endswith("<10, -1, true>")
    assert M.demod_plan(ctx, rtty, 1024, nsamples=n)["chain_groups"] == 0		# one round: nothing to fill
    assert M.demod_plan(ctx, rtty, 4096, nsamples=8 * rtty.samplebuf_size)["chain_groups"] == 0	# too short
    assert M.demod_plan(ctx, rtty, 4096, nsamples=n, ring_exact=True)["chain_groups"] == 0
    # (12000 baud runs demod_wave_kernel<4, 1>: the resumable instantiations are the generic ones)
    assert M.demod_plan(ctx, M.rx_config("12000"), 8192, nsamples=96000)["chain_groups"] == 0
