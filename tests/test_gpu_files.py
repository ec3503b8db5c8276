"""The step before the path, end to end (SURVEY 8 f3, 8 d "H2D-inclusive"): a LIST of WAV
files -> one batch (mifsk_demod_files: headers parsed, raw samples pread() into pinned memory,
PCM16 converted and --Xrxnoise added on the device, the receive loop over chunks while the next
chunk is read and copied), and the pipelined host entry mifsk_demod_batch_host[_ex] for
streams that start in host memory.  Checked per file against the reference program itself
(oracle/_ref/minimodem_ref --rx --file: stdout and the CARRIER / NOCARRIER lines) and against
the oracle restatement frame for frame.  Reference: src/simpleaudio-sndfile.c:42-74,
src/minimodem.c:1014-1032."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

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


def _make_files(M, tmp, n, rng, mode="1200", rates=(48000,), kw=None):
    """n WAV files made by the host transmitter (bit-identical to `minimodem --tx`,
    tests/test_cli.py): mixed PCM16 / float32, mixed lengths, leading silence, some with
    noise.  Returns [(path, float32 samples as the reference reads them, rate, is_s16)]."""
    kw = kw or {}
    out = []
    for i in range(n):
        rate = rates[i % len(rates)]
        cfg = M.rx_config(mode, sample_rate=rate, **kw)
        five = cfg.n_data_bits == 5
        words = rng.integers(0 if five else 32, 32 if five else 127, size=int(rng.integers(3, 90)),
                             dtype=np.uint8)
        s16 = bool(i % 3 != 1)
        x = M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 3000)),
                         amplitude=float(rng.uniform(0.3, 1.0)), s16=s16)
        if i % 4 == 3:
            x = (x + rng.normal(0, 0.05, x.shape)).astype(np.float32)
        path = os.path.join(tmp, "f%04d.wav" % i)
        if s16:
            pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16)
            O.write_wav(path, pcm, rate, True)
            x = pcm.astype(np.float32) / np.float32(32768.0)
        else:
            O.write_wav(path, x, rate, False)
        out.append((path, x, rate, s16))
    return out


def _ref_rx(path, rx_args):
    r = subprocess.run([O.MINIMODEM_REF, "--rx", "--file", path] + rx_args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return r.returncode, r.stdout, r.stderr.decode("latin-1")


def test_256_files_mixed_formats_equal_the_reference_program(gpu, tmp_path):
    """320 reference-format WAVs -- PCM16 and float32, 0.05 s .. 0.8 s, two sample rates,
    clean and noisy -- plus a file that is not a WAV, a stereo file and a missing one, as ONE
    call.  Every file's text and CARRIER / NOCARRIER lines must equal what the reference
    program prints for it (RING addressing = the reference's buffer semantics)."""
    M, torch, ctx = gpu
    if not O.have_ref():
        pytest.skip("oracle/_ref/minimodem_ref not built")
    rng = np.random.default_rng(2024)
    files = _make_files(M, str(tmp_path), 320, rng, rates=(48000, 22050))
    paths = [f[0] for f in files]
    bad = str(tmp_path / "notwav.wav")
    open(bad, "wb").write(b"this is not a RIFF file at all" * 10)
    stereo = str(tmp_path / "stereo.wav")
    import struct
    with open(stereo, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + 400) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000, 192000, 4, 16))
        f.write(b"data" + struct.pack("<I", 400) + bytes(400))
    missing = str(tmp_path / "nope.wav")
    empty = str(tmp_path / "empty.wav")
    O.write_wav(empty, np.zeros(0, np.int16), 48000, True)
    allp = paths[:100] + [bad] + paths[100:200] + [stereo, missing] + paths[200:] + [empty]
    res, stats = M.demod_files(ctx, allp, "1200", ring_exact=True)
    assert len(res) == len(allp)
    by_path = {r["path"]: r for r in res}
    assert by_path[bad]["error"] == -22 and by_path[stereo]["error"] == -95        # EINVAL, ENOTSUP
    assert by_path[missing]["error"] == -2                                          # ENOENT
    assert by_path[empty]["error"] == 0 and len(by_path[empty]["bits"]) == 0
    assert stats["streams"] == 321 and stats["bytes_h2d"] > 0
    with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 4)) as ex:
        refs = list(ex.map(lambda p: _ref_rx(p, ["1200"]), paths))
    nframes = 0
    for (path, x, rate, s16), (rc, rout, rerr) in zip(files, refs):
        r = by_path[path]
        assert r["error"] == 0 and rc == 0, path
        assert int(r["cfg"].sample_rate) == rate
        out, err = M.stream_text(r["cfg"], r["bits"], r["episodes"])
        assert out == rout, path
        assert err == rerr, path
        nframes += len(r["bits"])
    assert nframes > 8000


@pytest.mark.parametrize("mode,kw,rxnoise", [("1200", {}, 0.0), ("rtty", {}, 0.0), ("300", {}, 0.25),
                                             ("same", {}, 0.0), ("1200", dict(auto_carrier_threshold=0.001), 0.0)],
                         ids=["1200", "rtty", "300-rxnoise", "same", "1200-auto"])
@pytest.mark.parametrize("ring", [False, True], ids=["flat", "ring"])
def test_files_equal_oracle_frame_for_frame(gpu, tmp_path, mode, kw, rxnoise, ring):
    """Per-frame records and episodes of every file == the oracle on the samples the reference
    would read from it (PCM16 / 32768, --Xrxnoise as the constant it really is), flat and RING."""
    M, torch, ctx = gpu
    rng = np.random.default_rng(7)
    files = _make_files(M, str(tmp_path), 40, rng, mode=mode, rates=(48000, 44100), kw=kw)
    res, _ = M.demod_files(ctx, [f[0] for f in files], mode, rxnoise=rxnoise, ring_exact=ring,
                           want_frames=True, **kw)
    total = 0
    for (path, x, rate, s16), r in zip(files, res):
        assert r["error"] == 0 and r["status"] == 0
        ocfg = O.oracle_config(mode, sample_rate=rate, **kw)
        if rxnoise:
            x = (x + (np.float32(0) - np.float32(0.5)) * (np.float32(rxnoise) * np.float32(2))).astype(np.float32)
        ref = O.oracle_rx_stream(ocfg, x, ring_mode=ring)
        assert r["frames"].tobytes() == ref["frames"].tobytes(), path
        assert r["episodes"].tobytes() == ref["episodes"].tobytes(), path
        assert r["bytes"] == ref["bytes"]
        if kw:
            assert r["carrier_band"] == ref["carrier_band"]
        total += len(ref["frames"])
    assert total > 300


@pytest.mark.parametrize("fmt", ["f32", "s16"])
@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_pipelined_host_entry_over_several_chunks(gpu, fmt, pinned):
    """mifsk_demod_batch_host_ex on a batch that spans several 64 MB chunks: ragged lengths,
    float32 and PCM16 rows, page-locked (DMA from the caller's memory) and ordinary (staged by
    threads) sources -- results equal the oracle's, and equal the device-resident call's."""
    M, torch, ctx = gpu
    cfg = M.rx_config("1200")
    ocfg = O.oracle_config("1200")
    rng = np.random.default_rng(99)
    nstreams, width = 150, 400000                       # 240 MB of float32: four chunks
    alloc = M.host_alloc if pinned else (lambda shape, dt: np.zeros(shape, dt))
    host = alloc((nstreams, width), np.int16 if fmt == "s16" else np.float32)
    host[:] = 0
    lens = np.zeros(nstreams, np.uint32)
    ref_x = []
    for i in range(nstreams):
        nw = int(rng.integers(40, 990))
        x = M.synthesize(cfg, rng.integers(32, 127, size=nw, dtype=np.uint8),
                         leading_silence=int(rng.integers(0, 2000)), amplitude=float(rng.uniform(0.3, 0.99)),
                         s16=fmt == "s16")
        if i % 5 == 0:
            x = np.clip(x + rng.normal(0, 0.04, x.shape), -1, 0.9999).astype(np.float32)
        if fmt == "s16":
            pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16)
            host[i, :len(pcm)] = pcm
            x = pcm.astype(np.float32) / np.float32(32768.0)
        else:
            host[i, :len(x)] = x
        lens[i] = len(x)
        ref_x.append(x)
    res = M.demod_batch_host(ctx, cfg, host, lens, episodes_cap=8, stats=True)
    st = res["stats"]
    assert st["chunks"] >= (2 if fmt == "s16" else 4) and st["streams"] == nstreams
    assert st["source_pinned"] == (1 if pinned else 0)
    for i in range(0, nstreams, 3):
        ref = O.oracle_rx_stream(ocfg, ref_x[i])
        nf = int(res["nframes"][i])
        assert res["frames"][i, :nf].tobytes() == ref["frames"].tobytes(), i
        assert res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"], i
        ne = int(res["nepisodes"][i])
        assert res["episodes"][i, :ne].tobytes() == ref["episodes"].tobytes()
    # the same batch resident in HBM through mifsk_demod_batch
    d = torch.from_numpy(np.stack([np.pad(x, (0, width - len(x))) for x in ref_x])).cuda()
    out = M.demod_batch(ctx, cfg, d, nsamples=torch.from_numpy(lens.astype(np.int32)).cuda(),
                        want=("bytes", "frames"), frames_cap=res["frames"].shape[1])
    torch.cuda.synchronize()
    dev = M.results_to_host(out)
    assert np.array_equal(dev["nframes"], res["nframes"].astype(dev["nframes"].dtype))
    for i in range(nstreams):
        nf = int(res["nframes"][i])
        assert dev["frames"][i, :nf].tobytes() == res["frames"][i, :nf].tobytes()
    if pinned:
        M.host_free(host)


def test_unreadable_file_is_its_own_error_and_lengths_are_classed(gpu, tmp_path, monkeypatch):
    """(1) a file whose samples cannot be read after its header was parsed (it
    shrank, pread failed) fails alone -- mifsk_demod_files returns 0 and the other files are
    decoded; (2) one long recording among short ones does not size everybody's output arrays:
    the batch is cut into length classes, every file still equals the oracle."""
    M, torch, ctx = gpu
    rng = np.random.default_rng(77)
    files = _make_files(M, str(tmp_path), 24, rng)
    # one recording 40 x longer than the rest
    cfg = M.rx_config("1200")
    words = rng.integers(32, 127, size=3000, dtype=np.uint8)
    x = M.synthesize(cfg, words, leading_silence=100)
    long_path = os.path.join(str(tmp_path), "long.wav")
    O.write_wav(long_path, x, 48000, False)
    files.append((long_path, x, 48000, False))
    bad = files[5][0]
    os.rename(bad, bad.replace("f0005", "shrunk0005"))
    files[5] = (bad.replace("f0005", "shrunk0005"),) + files[5][1:]
    monkeypatch.setenv("MIFSK_EXPERIMENT", "1")
    monkeypatch.setenv("MIFSK_TEST_FAULT_READ", "shrunk")
    res, stats = M.demod_files(ctx, [f[0] for f in files], "1200")
    assert len(res) == len(files)
    ocfg = O.oracle_config("1200")
    for i, (path, x, rate, s16) in enumerate(files):
        if "shrunk" in path:
            assert res[i]["error"] == -5, res[i]["error"]        # -EIO, this file only
            continue
        assert res[i]["error"] == 0, (path, res[i]["error"])
        ref = O.oracle_rx_stream(ocfg, x, ring_mode=False)
        assert bytes(res[i]["bytes"]) == bytes(ref["bytes"]), path
    assert len(res[-1]["bytes"]) == 3000


def test_cli_decodes_a_list_of_files_as_one_batch(gpu, tmp_path):
    """`minimodem_mifsk_batch --rx --file a --file b ... 1200`: the batch binding as a program
    (csrc/mifsk_cli.c over mifsk_demod_files), against the reference run once per file."""
    M, torch, ctx = gpu
    if not O.have_ref():
        pytest.skip("oracle/_ref/minimodem_ref not built")
    cli = os.path.join(O.REF_DIR, "minimodem_mifsk_batch")
    rng = np.random.default_rng(5)
    files = _make_files(M, str(tmp_path), 24, rng)
    args = [cli, "--rx"]
    for f in files:
        args += ["--file", f[0]]
    r = subprocess.run(args + ["--batch-stats", "1200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode("latin-1")
    want_out = b""
    want_err = ""
    for f in files:
        rc, o, e = _ref_rx(f[0], ["1200"])
        want_out += o
        want_err += "### FILE %s\n" % f[0] + e
    assert r.stdout == want_out
    got_err = r.stderr.decode("latin-1")
    assert got_err.startswith(want_err) and "### BATCH files=24" in got_err[len(want_err):]
    # a file that cannot be read makes the exit status non-zero, the others are still decoded
    r = subprocess.run([cli, "--rx", "--file", files[0][0], "--file", str(tmp_path / "missing.wav"), "1200"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and r.stdout == _ref_rx(files[0][0], ["1200"])[1]
