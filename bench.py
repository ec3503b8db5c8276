#!/usr/bin/env python3
"""bench.py -- audio samples/sec demodulated on MI355X (whole job), 48 kHz f32.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1200|rtty|12000|same]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1] (the one the metric is quoted on):
per GPU a batch of 1024 independent synthetic streams x 10 s (480000 samples) of
48 kHz mono f32 Bell-202 audio, generated the way `minimodem --tx` generates it,
resident in HBM before the timed region.  `--config` selects the other BASELINE
entries (same JSON shape, `config.workload` names the entry):
    rtty    configs[2]  RTTY 45.45 baud, 4096 streams x 30 s
    12000   configs[3]  12000 baud, 8192 streams x 2 s per GPU (65536 over 8 GPUs)
    same    configs[4]  NOAA SAME 520.83 baud, 8192 streams x 10 s per GPU, amplitude 0.5,
                        AWGN SNR sweep inf/20/12/9/6/3 dB + the reference's DC-offset sweep
    1200noise           configs[1]'s batch under the impairments the reference's own noise tests
                        apply at 1200 baud (tests/40-noise.test:14-20): clean, AWGN 20/12/9/6 dB,
                        DC offset 0.05/0.10/0.50, interleaved over the 1024 streams

One "step" = one pass of the receive path over the whole resident batch
(mifsk_demod_batch: frame search + bit correlation + receive loop on the device)
plus, for N > 1, the gather of decoded bytes to rank 0 over RCCL (overlapped with
the next step).  Streams shard across ranks with no data-path collective.
`--scaling weak` (default) keeps the per-GPU batch fixed; `--scaling strong` keeps
the job's total fixed (8 x the per-GPU batch: 65536 streams for --config 12000,
the size BASELINE.json states) and divides it over the ranks.

Timing: first of all K launches are timed COLD (`cold_ms_per_step`: what a
25-launch job sees right after host-side set-up); then the kernel is launched untimed for 0.3 s
(`--preheat-ms`: the first launches after set-up run at idle clocks, and the driver's K = 20 is
7-11 ms of GPU work); then K launches SERIALLY on one stream, one event pair around them
(`roofline.*`, `value_serial`); then the W warm-up and the K timed passes go through the library's
pipeline (mifsk_pipeline_*, `--pipeline` P: pass i on lane i mod P -- a context, a HIP stream and
an output set of the library's own -- so that it fills the CUs the late streams of pass i - 1
leave idle), bracketed by barrier + synchronize as the contract says (`value`, `ms_per_step`).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : HBM-read roofline of the demod kernel, measured live with
                 events on the launch stream (serial launches, see above)
  value_serial : samples / roofline.kernel_ms_avg -- one batch at a time, no overlap
  cold_ms_per_step : K serial launches before any preheat
  pipeline     : passes in flight (asked for / in effect / hardware queues), ms per pass and
                 fraction of 8 TB/s of the timed passes; every output set verified
  small_batch  : configs[1] at 64 / 256 / 512 streams, launched serially (the chain-bound regime)
  preheat      : the untimed launches before the warm-up
  cpu_baseline : the reference's own CPU path (oracle/_ref: unmodified
                 src/*.c + FFT shim), one process per host core, on a bounded
                 sample of the same batch
  cpu_port     : the oracle restatement (direct 2-bin DFT) on one core, on the
                 full batch (1200) or a bounded sample, output compared with the GPU's
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# Passes in flight need hardware queues of their own: HIP multiplexes a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (4 by default -- the null stream and three lanes), and lanes
# that share one run one after the other.  Read when the HIP runtime starts, i.e. before torch is
# imported (in main()); a value the caller has set is kept.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSAMPLES = int(os.environ.get("MIFSK_BENCH_NSAMPLES", "480000"))	# 10 s at 48 kHz (override: experiments only)
HBM_PEAK = 8.0e12		# B/s, MI355X spec (MI355X_MICROARCH.md)
PCIE_PEAK = 63.0e9		# B/s, PCIe gen5 x16 per direction
WORLDLOADS_DEFAULT_STREAMS = 1024

# name -> (BASELINE.json entry, rx mode, streams per GPU, seconds, word range, amplitude)
WORKLOADS = {
    "1200": ("configs[1]: Bell202 1200-baud, 48 kHz f32, batch of 1024 synthetic streams",
             "1200", 1024, NSAMPLES / 48000.0, (0x20, 0x7F), 1.0),
    "rtty": ("configs[2]: RTTY 45.45-baud (long correlation windows), 48 kHz, 4096 streams",
             "rtty", 4096, 30.0, (0, 32), 1.0),
    "12000": ("configs[3]: 12000-baud, 48 kHz f32, 65536 streams over 8 GPUs = 8192 per GPU",
              "12000", 8192, 2.0, (0x20, 0x7F), 1.0),
    "same": ("configs[4]: NOAA SAME 520.83-baud with additive-noise SNR sweep (tests/40-style)",
             "same", 8192, 10.0, (0x20, 0x7F), 0.5),
    # not a BASELINE entry of its own: configs[1]'s workload, same size, under impairments --
    # what the headline kernel (which speculates on the lock holding) does when it breaks
    "1200noise": ("configs[1] under impairments: Bell202 1200-baud, 48 kHz f32, 1024 streams, clean / AWGN "
                  "20, 12, 9, 6 dB / DC offset 0.05, 0.10, 0.50 interleaved (reference tests/40-noise.test:14-20)",
                  "1200", 1024, NSAMPLES / 48000.0, (0x20, 0x7F), 1.0),
}
# conditions interleaved over a batch by GLOBAL stream id (stream g gets condition g % len):
# the SNR sweep of SURVEY 8(d) plus the reference's whole DC-offset list (tests/40-noise.test:20:
# 0.0 0.05 0.10 0.50)
SAME_CONDITIONS = [("snr_db", None), ("snr_db", 20), ("snr_db", 12), ("snr_db", 9), ("snr_db", 6),
                   ("snr_db", 3), ("dc", 0.05), ("dc", 0.10), ("dc", 0.50)]
NOISE1200_CONDITIONS = [("snr_db", None), ("snr_db", 20), ("snr_db", 12), ("snr_db", 9), ("snr_db", 6),
                        ("dc", 0.05), ("dc", 0.10), ("dc", 0.50)]
CONDITIONS = {"same": SAME_CONDITIONS, "1200noise": NOISE1200_CONDITIONS}
# conditions under which the payload is not expected to survive (not a pass criterion)
LOSSY = (("snr_db", 12), ("snr_db", 9), ("snr_db", 6), ("snr_db", 3))


def condition_label(kind, v):
    return "clean" if v is None else ("%g dB SNR" % v if kind == "snr_db" else "DC offset %g" % v)


def apply_conditions(name, torch, samples, lens, lo, rank, amplitude):
    """AWGN at the condition's SNR (signal power amplitude^2 / 2, seeded per rank) or the
    reference's --Xrxnoise DC term, condition (lo + i) % len(conditions) on row i, over each row's
    own length (`lens`; None: the whole row) -- the zero padding beyond it stays zero."""
    conds = CONDITIONS[name]
    nc = len(conds)
    p_sig = amplitude ** 2 / 2
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + rank)
    cols = torch.arange(samples.shape[1], device="cuda")[None, :]
    for k, (kind, v) in enumerate(conds):
        first = (k - lo) % nc
        rows = samples[first::nc]
        inside = 1.0 if lens is None else (cols < lens[first::nc][:, None]).to(torch.float32)
        if kind == "snr_db" and v is not None:
            sigma = float(np.sqrt(p_sig / 10 ** (v / 10)))
            rows += torch.randn(rows.shape, generator=g, device="cuda", dtype=torch.float32) * sigma * inside
        elif kind == "dc":
            rows -= np.float32(v) * inside


def make_stream(M, cfg, gid):
    """Stream `gid` of the Bell-202 batch: seeded printable payload, 0..40
    samples of leading silence, zero tail up to NSAMPLES."""
    rng = np.random.default_rng(1234 + gid)
    lead = int(rng.integers(0, 41))
    nbytes = (NSAMPLES - lead - 4 * 40) // 400
    payload = rng.integers(0x20, 0x7F, size=nbytes, dtype=np.uint8)
    x = M.synthesize(cfg, payload, leading_silence=lead)
    assert len(x) <= NSAMPLES
    return x, payload


def stream_words(name, cfg, gid, nsamp):
    """(words, leading silence) of stream `gid` of a device-generated batch; any rank can
    regenerate any stream's payload from its global id."""
    lo, hi = WORKLOADS[name][4]
    rng = np.random.default_rng(4321 + gid)
    frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
    nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 41 - (16 * frame if cfg.do_rx_sync else 0)) / frame) - 2
    # SAME frames have no start/stop bits: the reference's byte alignment on the preamble holds
    # only for streams that start on a bit boundary (tests/test_gpu_fullsize.py)
    lead = 0 if name == "same" else int(rng.integers(0, 41))
    return rng.integers(lo, hi, size=nwords, dtype=np.uint8), lead


def write_wav_f32(path, x, sr):
    import struct
    data = np.asarray(x, "<f4").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt ")
        f.write(struct.pack("<IHHIIHH", 16, 3, 1, sr, sr * 4, 4, 32))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


def _ref_decode(args):
    """one reference process (runs in a worker thread; the work is in the child)"""
    exe, path, mode = args
    r = subprocess.run([exe, "--rx", "--quiet", "--file", path, mode],
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    return r.stdout


def cpu_baselines(name, mode, sample_rate, host, lens, gpu_bytes, gpu_nbytes, gpu_text, budget_s=18.0, port=True):
    """Rank 0, N=1 only.  Times the CPU checkers on this box's host cores on `host` (float32
    [k, n]: the first k streams of the batch) and cross-checks their output against the GPU's
    (gpu_text[i] = what the GPU path prints for stream i: device frame bits through the host
    post-pass mifsk_stream_text)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    ocfg = O.oracle_config(mode)
    out = {}
    ncores = os.cpu_count() or 1

    # (1) the oracle restatement, one core
    if port or not O.have_ref():
        t0 = time.perf_counter()
        mismatches, nsamp = 0, 0
        for i in range(host.shape[0]):
            r = O.oracle_rx_stream(ocfg, host[i, :lens[i]], ring_mode=False)
            nsamp += int(lens[i])
            if r["bytes"] != gpu_bytes[i, :gpu_nbytes[i]].tobytes():
                mismatches += 1
        dt = time.perf_counter() - t0
        out["cpu_port"] = {
            "value": nsamp / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": "first %d streams (%d samples) through oracle/fsk_oracle.c (direct 2-bin DFT, f64 fma), "
                      "1 thread; decoded bytes compared with the GPU's: %d mismatching streams"
                      % (host.shape[0], nsamp, mismatches),
            "seconds": dt, "mismatching_streams": mismatches,
        }

    # (2) the reference program itself (unmodified src/*.c + shims), one process per core,
    # on as many streams as fit the time budget (calibrated on the first few)
    if O.have_ref():
        tmp = tempfile.mkdtemp(prefix="mifsk-bench-")
        paths = []

        def wav(i):
            p = os.path.join(tmp, "s%05d.wav" % i)
            write_wav_f32(p, host[i, :lens[i]], sample_rate)
            return p
        probe = min(host.shape[0], max(2, ncores))
        paths = [wav(i) for i in range(probe)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncores) as ex:
            outs = list(ex.map(_ref_decode, [(O.MINIMODEM_REF, p, mode) for p in paths]))
        t_probe = time.perf_counter() - t0
        per_round = max(t_probe, 1e-3)			# one stream per core takes this long
        rounds = max(1, int(budget_s / per_round))
        nref = min(host.shape[0], rounds * ncores)
        paths += [wav(i) for i in range(probe, nref)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=ncores) as ex:
            outs = list(ex.map(_ref_decode, [(O.MINIMODEM_REF, p, mode) for p in paths[:nref]]))
        dt = time.perf_counter() - t0
        bad = sum(1 for i in range(nref) if outs[i] != gpu_text[i])
        nsamp = int(sum(int(lens[i]) for i in range(nref)))
        for p in paths:
            os.unlink(p)
        os.rmdir(tmp)
        out["cpu_baseline"] = {
            "value": nsamp / dt, "unit": "samples/s", "cores": ncores, "kind": "reference",
            "sample": "first %d streams (%d samples) through oracle/_ref/minimodem_ref --rx --file "
                      "(reference src/*.c unmodified; FFTW3f absent in this image, FFT = oracle "
                      "double-precision shim), one process per host core, %d at a time; stdout "
                      "compared with the GPU's bytes: %d mismatching streams" % (nref, nsamp, ncores, bad),
            "seconds": dt, "mismatching_streams": bad,
        }
    else:
        out["cpu_baseline"] = dict(out["cpu_port"])
    return out


def kernel_source_id():
    """identity of the kernel sources a profile was taken with"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "minimodem_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")) or fn == "Makefile":
            with open(os.path.join(d, fn), "rb") as f:
                h.update(fn.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def hbm_traffic(name):
    """HBM bytes per kernel launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note), as recorded
    under profiles/ for this workload by tools/profile_round.sh -- reported only while the record
    was taken with the kernel sources that are running now; None otherwise (a stale ratio would
    silently survive a kernel change)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_hbm_traffic.json" % name)), reverse=True):
        try:
            with open(path) as f:
                rec = json.load(f)
            if rec.get("kernel_source_id") != kernel_source_id():
                continue
            return {"bytes_per_launch": rec["hbm_bytes_per_launch"],
                    "source": "profiles/" + os.path.basename(path),
                    "fetch_size_kb_raw": rec["fetch_size_kb_raw"], "write_size_kb_raw": rec["write_size_kb_raw"]}
        except Exception:					# noqa: BLE001
            continue
    return None


class RankFailed(RuntimeError):
    """some rank (maybe not this one) failed in a phase every rank has now left"""


def agree(torch, dist, err, what):
    """Every rank calls this at the end of a phase with its own exception (or None): an
    all-reduce of the failure flag, so that either all ranks go on or all raise RankFailed --
    a rank that threw never leaves the others waiting inside a collective it skipped."""
    flag = 1.0 if err is not None else 0.0
    if dist is not None:
        t = torch.tensor([flag], dtype=torch.float64, device="cuda" if torch.cuda.is_available() else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flag = float(t.item())
    if flag:
        raise RankFailed("%s: %s" % (what, repr(err) if err is not None else "another rank failed"))


_PIPES = {}


def get_pipeline(M, torch, depth):
    """one pipeline per depth and process (lanes made anew for every workload end up sharing
    hardware queues)"""
    if depth not in _PIPES:
        _PIPES[depth] = M.Pipeline(torch.cuda.current_device(), depth=depth)
    return _PIPES[depth]


def run_workload(name, args, M, torch, dist, ctx, rank, world, steps, warmup, cpu_leg, oracle_leg=True):
    """One BASELINE entry on this rank's shard: synthesize the batch on the device (or host for
    configs[1]), W untimed + K timed passes bracketed by barrier + synchronize, kernel time by
    events on the launch stream.  Returns the JSON line as a dict on rank 0, None elsewhere.
    Collective-safe: the phases that can fail on one rank alone (allocation, synthesis, a
    launch) end in agree(); inside the timed loop a rank whose launch failed still takes part
    in every gather and reports afterwards."""
    entry, mode, per_gpu, seconds, _, amplitude = WORKLOADS[name]
    cfg = M.rx_config(mode)
    per_gpu = args.streams or per_gpu
    if args.scaling == "strong":
        total_streams = per_gpu * 8			# the whole-job size BASELINE.json states for 8 GPUs
    else:
        total_streams = per_gpu * world
    lo, hi = M.shard_range(total_streams, rank, world)
    nstreams = hi - lo
    nsamp = NSAMPLES if name in ("1200", "1200noise") else int(seconds * cfg.sample_rate)
    stride = (nsamp + 3) & ~3

    # ---- synthetic batch, resident in HBM before anything is timed ----------
    payloads = [None] * nstreams
    setup_err = None
    samples = lens = bufs = None
    try:
        samples, lens = make_batch(name, M, torch, ctx, cfg, rank, lo, nstreams, nsamp, stride, amplitude, payloads)
        torch.cuda.synchronize()
    except Exception as e:				# noqa: BLE001 -- reported through agree()
        setup_err = e
    agree(torch, dist, setup_err, "%s: batch synthesis" % name)
    total_samples_local = float(nstreams * nsamp if lens is None else int(lens.sum()))

    frames_cap = M.max_frames(cfg, stride)
    want = ("bytes",)
    kw = dict(want=want, frames_cap=frames_cap, nsamples=lens, engine=args.engine, episodes_cap=8)
    # Passes in flight (--pipeline P): pass i runs on stream i mod P with its own context (each
    # context owns its launch scratch) and its own output set, so that pass i + 1 fills the CUs
    # that pass i's late streams leave idle (a launch ends ~20 % after its mean stream on
    # configs[1], and long after it under impairments: DESIGN.md section 6).
    # (0 = this workload's own depth: three where the launch is one kernel whose tail is short
    # against its body -- K = 20, each lane on its own copy of the batch: 2, 3, 4, 5 in flight
    # give 0.351, 0.351, 0.355, 0.357 ms per pass on configs[1] -- four where the launches are
    # long serial chains or chained dispatches: 1200noise 0.90 -> 0.77, SAME 7.7 -> 7.3 ms)
    asked = int(args.pipeline) if args.pipeline > 0 else {"1200": 3, "12000": 3}.get(name, 4)
    pipe = None
    try:
        # The library's pipeline (mifsk_pipeline_*): lanes, streams and one output set per lane are
        # its own; every pass reads the ONE resident batch (a copy per lane measures the same:
        # profiles/r05_history.md section 6).  The first launch (module load, tables) happens here.
        pipe = get_pipeline(M, torch, asked)
        pipe.outputs(nstreams, frames_cap, episodes_cap=8, want=want)
        bufs = [M.demod_batch(ctx, cfg, samples, **kw)]		# the serial launches' output set
        for _ in range(pipe.depth):
            pipe.submit(cfg, samples, nsamples=lens, engine=args.engine)
        pipe.drain()
        torch.cuda.synchronize()
    except Exception as e:				# noqa: BLE001
        setup_err = e
    agree(torch, dist, setup_err, "%s: first launch" % name)
    return timed_workload(name, args, M, torch, dist, ctx, pipe, rank, world, steps, warmup, cpu_leg, oracle_leg,
                          cfg, mode, entry, samples, lens, bufs, kw, payloads, lo, nstreams, nsamp, stride,
                          frames_cap, total_streams, total_samples_local)


def make_batch(name, M, torch, ctx, cfg, rank, lo, nstreams, nsamp, stride, amplitude, payloads):
    """the rank's shard of the synthetic batch -> (samples on the device, lengths or None)"""
    if name in ("1200", "1200noise"):
        # host generator (threaded; it releases the GIL)
        host = np.zeros((nstreams, stride), np.float32)

        def gen(i):
            x, p = make_stream(M, cfg, lo + i)
            host[i, :len(x)] = x
            payloads[i] = p

        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            list(ex.map(gen, range(nstreams)))
        samples = torch.from_numpy(host).cuda()
        lens = None
        del host
        if name == "1200noise":
            apply_conditions(name, torch, samples, lens, lo, rank, amplitude)
    else:
        # device generator (mifsk_tx_synthesize_batch; bit-identical to the host one)
        wl = [stream_words(name, cfg, lo + i, nsamp) for i in range(nstreams)]
        nw = len(wl[0][0])
        words = np.stack([w for w, _ in wl])
        for i in range(nstreams):
            payloads[i] = wl[i][0]
        lead = torch.tensor([l for _, l in wl], dtype=torch.int32).cuda()
        samples, lens = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda(), stride=stride,
                                           leading_silence=lead, amplitude=amplitude)
        assert int(lens.max()) <= stride and nw > 0
        if name == "same":
            apply_conditions(name, torch, samples, lens, lo, rank, amplitude)
    return samples, lens


def oracle_verdict(name, mode, M, torch, ctx, cfg, samples, lens, kw, frames_cap, lo, world):
    """One more (untimed) pass that also writes frame records and episodes, then the oracle
    over every stream of the shard (tests/_oracle.py oracle_batch_mismatches).  Never raises:
    a checker that could not run says so in the line."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as O
        k2 = dict(kw)
        k2["want"] = ("bytes", "frames", "episodes")
        k2["episodes_cap"] = 64
        out = M.demod_batch(ctx, cfg, samples, **k2)
        torch.cuda.synchronize()
        res = M.results_to_host(out)
        del out
        threads = max(1, (os.cpu_count() or 1) // max(1, world))
        nc = len(CONDITIONS[name]) if name in CONDITIONS else 0
        groups = (lambda i: (lo + i) % nc) if nc else None
        t0 = time.perf_counter()
        bad, by_group, secs = O.oracle_batch_mismatches(O.oracle_config(mode), samples, lens, res,
                                                        threads=threads, groups=groups)
        nsamp_total = float(samples.shape[0] * samples.shape[1] if lens is None else int(lens.sum()))
        v = {"streams": int(samples.shape[0]), "mismatching_streams": len(bad),
             # (the restated loop on every host core at once: the strongest CPU figure of this
             # line -- samples of the shard / wall time of the oracle pass, D2H not included)
             "samples_per_s_all_cores": nsamp_total / max(secs, 1e-9),
             "compared": "every stream: nframes, frame records (bits, start, flags, confidence and "
                         "amplitude bit patterns), episodes, bytes",
             "threads": threads, "oracle_seconds": secs, "seconds": time.perf_counter() - t0}
        if bad:
            v["first_mismatching"] = [int(lo + i) for i in bad[:8]]
        if by_group:
            v["frames_equal_oracle_by_condition"] = {int(k): "%d/%d" % tuple(c) for k, c in sorted(by_group.items())}
        return v
    except Exception as e:					# noqa: BLE001
        return {"streams": int(samples.shape[0]), "mismatching_streams": None, "error": repr(e)}


def work_counters(name, M, torch, ctx, cfg, samples, lens, kw):
    """One more untimed pass with the kernels' per-stream work counters: general-path iterations
    (every pass of the reference's loop the lattice speculation did not cover), refinements
    (minimodem.c:1357-1389), frames accepted from the lattice.  None if the pass fails."""
    try:
        k2 = dict(kw)
        k2["want"] = ("bytes", "counters")
        out = M.demod_batch(ctx, cfg, samples, **k2)
        torch.cuda.synchronize()
        c = out["counters"].cpu().numpy().view(np.uint64).astype(np.float64)
        idx = {v: k for k, v in M.COUNTER_NAMES.items()}
        cols = {"general_path_iterations": c[:, idx["iterations"]], "refinements": c[:, idx["refines"]],
                "lattice_frames": c[:, idx["bulk_frames"]]}
        cols["frames"] = out["nframes"].cpu().numpy().astype(np.float64)
        return cols
    except Exception:					# noqa: BLE001 -- diagnostic only
        return None


def timed_workload(name, args, M, torch, dist, ctx, pipeline, rank, world, steps, warmup, cpu_leg, oracle_leg,
                   cfg, mode, entry, samples, lens, bufs, kw, payloads, lo, nstreams, nsamp, stride,
                   frames_cap, total_streams, total_samples_local):
    pipe = pipeline.depth			# lanes in effect (asked for: pipeline.depth_requested)
    nbuf = pipe					# pass t writes the library's set t % depth
    pending = [None] * nbuf
    # what a stream can decode at most is known on the host (its length): the gather ships
    # that many columns, not the whole frames_cap-wide buffer
    cols = int(M.max_frames(cfg, nsamp if lens is None else int(lens.max())))
    rows = [M.shard_range(total_streams, r, world)[1] - M.shard_range(total_streams, r, world)[0] for r in range(world)]
    # (--native-gather: the same exchange through the library's own entry, mifsk_gather_* -- RCCL
    # opened by libmifsk, a communicator of its own, the gather enqueued on the lane's stream)
    Gatherer = M.NativeGatherer if args.native_gather else M.ByteGatherer
    gatherer = Gatherer(dist, rank, world, cols=min(cols, frames_cap), rows=rows, slots=max(2, nbuf),
                        loopback=args.gather_self and world == 1)
    failure = [None]
    wait_s = [0.0]

    def wait_all(ws):
        t = time.perf_counter()
        for w in ws:
            w.wait()
        wait_s[0] += time.perf_counter() - t


    gather_on = world > 1 or args.gather_self

    def step(i):
        tk = pipeline.next_ticket()
        b = tk % nbuf
        # (everything of this pass is ordered on its lane's stream: the gather that last read
        # this output set, the launch, the next gather's sends)
        if pending[b] is not None:		# its buffers are about to be overwritten
            wait_all(pending[b])
            pending[b] = None
        try:
            pipeline.submit(cfg, samples, nsamples=lens, after=None, engine=args.engine)
        except Exception as e:			# noqa: BLE001 -- this rank still joins every gather
            failure[0] = failure[0] or e
            return
        if gather_on:
            # decoded bytes -> rank 0, grouped send/recv (each peer uses its own xGMI link),
            # queued on the lane's stream behind the pass
            out = pipeline.result(tk)
            with torch.cuda.stream(pipeline.stream(tk)):
                pending[b] = gatherer.start(out["bytes"], out["nbytes"])

    def drain():
        for b in range(nbuf):
            if pending[b] is not None:
                wait_all(pending[b])
                pending[b] = None
        pipeline.drain()

    # COLD: K serial launches as the first thing after set-up (the state a short job finds the
    # device in: the first launches after idling run at idle clocks), by the host's clock
    cold_ms = None
    if failure[0] is None:
        try:
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(steps):
                M.demod_batch(ctx, cfg, samples, out=bufs[0], **kw)
            torch.cuda.synchronize()
            cold_ms = (time.perf_counter() - tc) * 1e3 / max(1, steps)
        except Exception as e:			# noqa: BLE001
            failure[0] = e

    # Before the contract's W warm-up passes: bring the GPU out of its idle state.  A pass of
    # configs[1] is 0.45 ms and the driver's W = 5, K = 20 is 11 ms after seconds of host-side
    # set-up: the first two dozen launches after idling run 6-8 % slower than every later one
    # (same box, same process: K = 20 -> 0.450-0.457 ms per launch, K = 200 -> 0.4239,
    # profiles/r05_history.md section 5).  Untimed, kernel only (no gather), disclosed in the line.
    preheat = {"ms": 0.0, "launches": 0}
    if args.preheat_ms > 0 and failure[0] is None:
        tp = time.perf_counter()
        try:
            while (time.perf_counter() - tp) * 1e3 < args.preheat_ms and preheat["launches"] < 4096:
                for _ in range(8):
                    M.demod_batch(ctx, cfg, samples, out=bufs[0], **kw)
                    preheat["launches"] += 1
                torch.cuda.synchronize()
        except Exception as e:			# noqa: BLE001
            failure[0] = e
        preheat["ms"] = (time.perf_counter() - tp) * 1e3

    # With passes in flight on several streams no event pair brackets "a launch": the kernel's own
    # average duration is taken here, from K launches on ONE stream between the preheat and the
    # warm-up passes (untimed by the contract) -- the figure the rocprofv3 summary under profiles/
    # is compared with -- and the K timed passes below are timed by the contract's clock alone.
    serial_evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    try:
        serial_evs[0].record()			# (torch's current stream, where ctx launches by default)
        for i in range(steps):
            M.demod_batch(ctx, cfg, samples, out=bufs[0], **kw)
        serial_evs[1].record()
        torch.cuda.synchronize()
    except Exception as e:			# noqa: BLE001
        failure[0] = failure[0] or e

    for i in range(warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wait_s[0] = 0.0

    # (the kernel's own duration: the serial launches above, one event pair around the K of them)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # (this rank's own clock stopped after the closing barrier: the same for all; what differs
    # per rank is how long its launches and its waits for the gather took)
    agree(torch, dist, failure[0], "%s: timed loop" % name)

    kernel_ms = [serial_evs[0].elapsed_time(serial_evs[1]) / max(1, steps)] * steps
    total_samples = total_samples_local
    per_rank = None
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([total_samples_local], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_samples = float(t.item())
        # root against peers: if the efficiency at N > 1 is short, this says whether the root
        # (which also receives N - 1 shards per step) is the slow rank, and by how much
        mine = torch.tensor([float(np.mean(kernel_ms)), float(np.max(kernel_ms)), wait_s[0] / max(1, steps) * 1e3],
                            dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"kernel_ms_avg": [float(a[0]) for a in allr], "kernel_ms_max": [float(a[1]) for a in allr],
                    "gather_wait_ms_per_step": [float(a[2]) for a in allr],
                    "gather_bytes_per_peer_per_step": gatherer.bytes_per_peer(nstreams)}

    # every output set the timed passes wrote against the serial launches' set (bufs[0])
    res = M.results_to_host(bufs[0])
    gpu_bytes, gpu_nbytes = res["bytes"], res["nbytes"]
    sets_equal = 0
    for b in range(nbuf):
        r = M.results_to_host(pipeline.result(b))
        same = np.array_equal(r["nbytes"], gpu_nbytes) and np.array_equal(r["nframes"], res["nframes"])
        if same:
            mask = np.arange(gpu_bytes.shape[1])[None, :] < gpu_nbytes[:, None].astype(np.int64)
            same = np.array_equal(r["bytes"][mask], gpu_bytes[mask])
        sets_equal += bool(same)

    # ---- the checker leg (outside every timed region): the WHOLE shard against the oracle,
    # frame for frame -- bits, starts, flags, confidence and amplitude bit patterns, episodes,
    # bytes -- on this box's host cores (every rank checks its own shard on its share of them)
    oracle = None
    if oracle_leg:
        oracle = oracle_verdict(name, mode, M, torch, ctx, cfg, samples, lens, kw, frames_cap, lo, world)
        if dist is not None:
            t = torch.tensor([oracle["mismatching_streams"] if oracle.get("error") is None else 0.0,
                              oracle["streams"], 1.0 if oracle.get("error") else 0.0],
                             dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            oracle["mismatching_streams_all_ranks"] = int(t[0].item())
            oracle["streams_all_ranks"] = int(t[1].item())
            oracle["ranks_failed"] = int(t[2].item())

    def stream_ok(b, nb, payload, gid):
        got = b[:nb].tobytes()
        if name == "1200":
            return got == payload.tobytes()
        if name in CONDITIONS and CONDITIONS[name][gid % len(CONDITIONS[name])] in LOSSY:
            return None				# whether it survives the noise is not a criterion
        return payload.tobytes() in got		# (rtty: 5-bit words; the leader may add a frame in front)

    verdicts = [stream_ok(gpu_bytes[i], int(gpu_nbytes[i]), payloads[i], lo + i) for i in range(nstreams)]
    ok_streams = sum(1 for v in verdicts if v)
    judged = sum(1 for v in verdicts if v is not None)
    # SURVEY 8(d), configs[4]: streams whose whole payload survives, per condition of the sweep
    # (what the reference's algorithm loses to the noise -- the GPU's frames equal the oracle's
    # either way, tests/test_gpu_fullsize.py -- and what byte error rate is left)
    by_condition = None
    if name in CONDITIONS:
        by_condition = {}
        work = work_counters(name, M, torch, ctx, cfg, samples, lens, kw) if rank == 0 else None
        for k, (kind, v) in enumerate(CONDITIONS[name]):
            label = condition_label(kind, v)
            mine = [i for i in range(nstreams) if (lo + i) % len(CONDITIONS[name]) == k]
            whole = sent = out = 0
            for i in mine:
                got = gpu_bytes[i][:int(gpu_nbytes[i])].tobytes()
                pay = payloads[i].tobytes()
                whole += pay in got
                sent += len(pay)
                out += len(got)
            by_condition[label] = {"streams": len(mine), "whole_payload": whole,
                                   "bytes_decoded_over_sent": out / max(1, sent)}
            if oracle is not None and "frames_equal_oracle_by_condition" in oracle:
                # the decode error rate vs the ORACLE per condition (streams whose
                # every frame record equals the oracle's on the identical buffer)
                by_condition[label]["frames_equal_oracle"] = oracle["frames_equal_oracle_by_condition"].get(k)
            if work is not None:
                # how often the speculation on the lock breaks under this condition (per stream)
                by_condition[label]["per_stream"] = {key: float(np.mean(col[mine])) for key, col in work.items()}
    # the bytes gathered from the peers are checked too (rank 0): each peer's streams are
    # regenerated from their global ids
    peers_ok = peers_judged = 0
    if rank == 0 and world > 1:
        for r in range(1, world):
            plo, phi = M.shard_range(total_streams, r, world)
            rb, rn = gatherer.received(r)
            pb = rb.cpu().numpy()
            pn = rn.cpu().numpy()
            for j in range(phi - plo):
                gid = plo + j
                if name in ("1200", "1200noise"):
                    rng = np.random.default_rng(1234 + gid)
                    lead_ = int(rng.integers(0, 41))
                    pay = rng.integers(0x20, 0x7F, size=(NSAMPLES - lead_ - 4 * 40) // 400, dtype=np.uint8)
                else:
                    pay = stream_words(name, cfg, gid, nsamp)[0]
                v = stream_ok(pb[j], int(pn[j]), pay, gid)
                if v is not None:
                    peers_judged += 1
                    peers_ok += bool(v)

    gather_self_ok = None
    if args.gather_self and world == 1:
        # what the last gather delivered (to this rank, from itself) against the set it was sent from
        rb, rn = gatherer.received(0)
        last = M.results_to_host(pipeline.result((pipeline.next_ticket() - 1) % nbuf))
        gather_self_ok = bool(np.array_equal(rn.cpu().numpy(), last["nbytes"])
                              and np.array_equal(rb.cpu().numpy(), last["bytes"][:, :rb.shape[1]]))

    line = None
    if rank == 0:
        value = total_samples * steps / dt
        kavg = float(np.mean(kernel_ms)) * 1e-3
        achieved = total_samples_local * 4.0 / kavg
        launch = M.demod_plan(ctx, cfg, nstreams, engine=args.engine, nsamples=stride)	# what the library launches
        line = {
            "metric": "audio samples/sec demodulated (whole node), %s-baud 48 kHz f32"
                      % {"1200": "1200", "1200noise": "1200", "rtty": "45.45", "12000": "12000", "same": "520.83"}[name],
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s; %d streams x %d samples per GPU (BASELINE.json)"
                                   % (entry, nstreams, nsamp),
                       "streams_per_gpu": nstreams, "samples_per_stream": nsamp,
                       "total_streams": total_streams,
                       "input_dtype": "f32", "accumulate_dtype": "f64",
                       "sharding": ("independent streams per rank, decoded bytes gathered to "
                                    "rank 0 over RCCL") if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": hbm_traffic(name),
                         "kernel": launch["kernel"], "launch": launch,
                         "kernel_ms_avg": kavg * 1e3, "kernel_ms_min": float(np.min(kernel_ms)),
                         "events": "one pair around K launches on one stream, between the preheat and the warm-up passes",
                         "algorithmic_bytes_per_launch": total_samples_local * 4.0},
            "payload_roundtrip_ok_streams": "%d/%d" % (ok_streams, judged),
            "device": ctx.device_name,
            # (one batch at a time per GPU: the job's samples over the slowest rank's kernel time)
            "value_serial": total_samples / (max(per_rank["kernel_ms_avg"]) * 1e-3 if per_rank else kavg),
            "cold_ms_per_step": cold_ms,
            "pipeline": {"passes_in_flight": pipe, "asked_for": pipeline.depth_requested,
                         "hw_queues": pipeline.hw_queues, "through": "mifsk_pipeline_* (C ABI)",
                         "ms_per_pass": dt / steps * 1e3,
                         "hbm_frac_of_the_timed_passes": total_samples_local * 4.0 * steps / dt / HBM_PEAK,
                         "output_sets_equal_to_serial_launch": "%d/%d" % (sets_equal, nbuf),
                         "note": "pass i runs on lane i mod P of the library's pipeline (a context, a HIP stream "
                                 "and an output set each; one resident batch): it fills the CUs the late streams "
                                 "of pass i - 1 leave idle; roofline.* and value_serial are the kernel launched "
                                 "serially on one stream"},
            "preheat": {"untimed_launches_before_warmup": preheat["launches"], "ms": preheat["ms"],
                        "why": "the first launches after host-side set-up run at idle clocks"},
        }
        if by_condition is not None:
            line["payload_by_condition"] = by_condition
        if oracle is not None:
            line["oracle_mismatching_streams"] = (oracle.get("mismatching_streams_all_ranks")
                                                  if world > 1 else oracle["mismatching_streams"])
            line["oracle"] = oracle
        if world > 1:
            line["payload_roundtrip_ok_streams_gathered_from_peers"] = "%d/%d" % (peers_ok, peers_judged)
        if gather_self_ok is not None:
            line["gather_self_ok"] = gather_self_ok
        if world > 1 or args.gather_self:
            line["gather_through"] = "mifsk_gather_* (C ABI)" if args.native_gather else "torch.distributed (ByteGatherer)"
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["ranks"] = {"world_size": int(dist.get_world_size()), "backend": str(dist.get_backend())}
        if world == 1 and cpu_leg:
            # a bounded sample of the same batch on the host cores ("ref": the reference program
            # alone on 64 streams -- what the default run does for the workloads after the first)
            k = nstreams if name == "1200" else {"rtty": 256, "12000": 1024, "same": 512, "1200noise": 256}[name]
            if cpu_leg == "ref":
                k = 64
            k = min(k, nstreams)
            hs = samples[:k].cpu().numpy()
            hl = np.full(k, nsamp, np.int64) if lens is None else lens[:k].cpu().numpy().astype(np.int64)
            # what the GPU path prints for those streams (frame bits -> databits post-pass)
            o2 = M.demod_batch(ctx, cfg, samples[:k], nsamples=None if lens is None else lens[:k],
                               want=("bits", "episodes"), frames_cap=frames_cap, episodes_cap=64,
                               engine=args.engine)
            torch.cuda.synchronize()
            r2 = M.results_to_host(o2)
            gpu_text = [M.stream_text(cfg, r2["bits"][i, :int(r2["nframes"][i])],
                                      r2["episodes"][i, :min(64, int(r2["nepisodes"][i]))], quiet=True)[0]
                        for i in range(k)]
            line.update(cpu_baselines(name, mode, int(cfg.sample_rate), hs, hl, gpu_bytes, gpu_nbytes,
                                      gpu_text, budget_s=6.0 if cpu_leg == "ref" else 18.0,
                                      port=cpu_leg != "ref"))
        if world == 1 and name == "1200" and not args.no_h2d:
            line["h2d_inclusive"] = h2d_inclusive(M, torch, ctx, cfg, samples, nsamp, frames_cap, gpu_bytes,
                                                  gpu_nbytes)
        if world == 1 and name == "1200" and cpu_leg is True:
            line["legacy_dropin"] = legacy_dropin(samples, nsamp, int(cfg.sample_rate))
        if world == 1 and name == "1200" and args.config is None and failure[0] is None:
            line["small_batch"] = small_batch(M, torch, ctx, cfg, samples, nsamp, steps, kw)
    del samples, bufs
    torch.cuda.empty_cache()
    return line


def small_batch(M, torch, ctx, cfg, samples, nsamp, steps, kw):
    """configs[1] at 64 / 256 / 512 streams, K launches serially on one stream after the preheat:
    below one workgroup per CU (256 streams) the launch lasts as long as ONE stream's serial chain
    whatever the batch size -- the regime a small job is in (no time parallelism inside a stream:
    DESIGN.md section 8)."""
    out = {"unit": "ms per launch (serial, events around K launches)"}
    for n in (64, 256, 512):
        if n > samples.shape[0]:
            continue
        try:
            sub = samples[:n]
            buf = M.demod_batch(ctx, cfg, sub, **kw)
            for _ in range(3):
                M.demod_batch(ctx, cfg, sub, out=buf, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                M.demod_batch(ctx, cfg, sub, out=buf, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / max(1, steps)
            out[str(n)] = {"kernel_ms_avg": ms, "samples_per_s": n * nsamp / (ms * 1e-3),
                           "roofline_frac": n * nsamp * 4.0 / (ms * 1e-3) / HBM_PEAK}
        except Exception as e:				# noqa: BLE001 -- a diagnostic leg must not kill the line
            out[str(n)] = {"error": repr(e)}
    return out


def h2d_inclusive(M, torch, ctx, cfg, samples, nsamp, frames_cap, gpu_bytes, gpu_nbytes):
    """SURVEY 8(d)'s third timing: the same batch starting in HOST memory (pinned), through the
    library's pipelined host entry -- chunked H2D on a copy stream overlapped with the demod of
    the chunk before, results copied back -- once as f32 (4 B per sample over PCIe) and once as
    the S16 a WAV file holds (2 B per sample, converted on the device).  Never `value`."""
    out = {"unit": "samples/s", "pcie_peak_GBps": PCIE_PEAK / 1e9}
    nstreams = samples.shape[0]
    for fmt in ("f32", "s16"):
        try:
            if fmt == "f32":
                host = M.host_alloc((nstreams, samples.shape[1]), np.float32)
                host[:] = samples.cpu().numpy()
            else:
                host = M.host_alloc((nstreams, samples.shape[1]), np.int16)
                host[:] = torch.round(samples * 32767.0).to(torch.int16).cpu().numpy()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                r = M.demod_batch_host(ctx, cfg, host, frames_cap=frames_cap, episodes_cap=8,
                                       want=("bytes",))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            ok = sum(1 for i in range(nstreams)
                     if r["bytes"][i, :int(r["nbytes"][i])].tobytes()
                     == gpu_bytes[i, :int(gpu_nbytes[i])].tobytes())
            bps = 4.0 if fmt == "f32" else 2.0
            out[fmt] = {"value": nstreams * nsamp / best, "seconds": best,
                        "pcie_GBps": nstreams * nsamp * bps / best / 1e9,
                        "frac_of_pcie_peak": nstreams * nsamp * bps / best / PCIE_PEAK,
                        "streams_equal_to_resident_run": "%d/%d" % (ok, nstreams)}
            M.host_free(host)
        except Exception as e:					# noqa: BLE001 -- a diagnostic leg must not kill the line
            out[fmt] = {"error": repr(e)}
    return out


def legacy_dropin(samples, nsamp, sample_rate):
    """What INTEGRATION.md section 1 costs: the reference's unmodified main() over the five legacy
    fsk_* symbols (oracle/_ref/minimodem_mifsk: one H2D + launch + D2H per fsk_find_frame call),
    next to the same main() with integration/minimodem-rx-batch.patch (oracle/_ref/
    minimodem_mifsk_rxbatch: the file as a batch of one), on ONE stream of configs[1] read from a
    WAV file -- process start, file read and context creation included in both.  Never `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    out = {"unit": "samples/s", "sample": "stream 0 of the batch (%d samples) as a float32 WAV, one process, wall clock" % nsamp}
    tmp = tempfile.mkdtemp(prefix="mifsk-legacy-")
    path = os.path.join(tmp, "s0.wav")
    try:
        write_wav_f32(path, samples[0, :nsamp].cpu().numpy(), sample_rate)
        texts = {}
        for key, exe in (("five_legacy_symbols", os.path.join(O.REF_DIR, "minimodem_mifsk")),
                         ("rx_batch_patch", os.path.join(O.REF_DIR, "minimodem_mifsk_rxbatch")),
                         ("reference_cpu", O.MINIMODEM_REF)):
            if not os.path.exists(exe):
                out[key] = {"error": "not built"}
                continue
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--rx", "--quiet", "--file", path, "1200"],
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            texts[key] = r.stdout
            out[key] = {"value": nsamp / best, "seconds": best, "returncode": r.returncode}
        if len(texts) > 1:
            ref = texts.get("reference_cpu", next(iter(texts.values())))
            out["outputs_identical"] = all(t == ref for t in texts.values())
    except Exception as e:					# noqa: BLE001 -- a diagnostic leg must not kill the line
        out["error"] = repr(e)
    finally:
        try:
            os.unlink(path)
            os.rmdir(tmp)
        except OSError:
            pass
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks
    ourselves (one process per GPU, rendezvous on 127.0.0.1) and pass their output through."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS),
                    help="one BASELINE entry only (default: configs[1] as the line, the others "
                         "with a few steps each under its \"configs\" key)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the config's)")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip the CPU legs (the timed baselines and the whole-batch oracle verdict)")
    ap.add_argument("--no-h2d", action="store_true", help="skip the H2D-inclusive leg")
    ap.add_argument("--no-extra", action="store_true", help="configs[1] only: skip configs[2..4]")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="passes in flight: pass i is launched on stream i mod P (1 = one stream; "
                         "0 = the workload's own default, 3 or 4)")
    ap.add_argument("--preheat-ms", type=float, default=300.0,
                    help="untimed kernel launches before the W warm-up passes (0 = none)")
    ap.add_argument("--gather-self", action="store_true",
                    help="N = 1 only: run the N > 1 step structure anyway -- process group on the nccl "
                         "backend, the gather of every pass as a send to this rank itself on the lane's stream")
    ap.add_argument("--native-gather", action="store_true",
                    help="N > 1 (or --gather-self): gather through mifsk_gather_* (C ABI, RCCL opened by the "
                         "library) instead of torch.distributed's grouped isend / irecv")
    ap.add_argument("--engine", default=None, choices=["wave", "workgroup"],
                    help="force a receive-loop engine (default: the library chooses)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)

    # ONE JSON line on stdout and nothing else: libraries below print there too (RCCL writes a
    # version banner through C stdio, flushed when the process exits -- after the line).  The line
    # goes to a private copy of stdout; file descriptor 1 itself is pointed at stderr, on every rank.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(line_fd, (json.dumps(obj) + "\n").encode())

    if os.environ.get("MIFSK_BENCH_DRYRUN"):
        # the launch path without a GPU (tests/test_distributed_cpu.py): rendezvous over gloo,
        # shard the batch, reduce, print -- everything bench.py does around the kernels
        import torch
        import torch.distributed as dist
        import minimodem_amd as M
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        # the job's size exactly as run_workload() derives it (--config, --streams, --scaling)
        per_gpu = args.streams or WORKLOADS[args.config or "1200"][2]
        total_streams = per_gpu * 8 if args.scaling == "strong" else per_gpu * world
        lo, hi = M.shard_range(total_streams, rank, world)
        t = torch.tensor([float(hi - lo)], dtype=torch.float64)
        table = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_gather(table, torch.tensor([lo, hi], dtype=torch.int64))
            dist.barrier()
        else:
            table = [torch.tensor([lo, hi], dtype=torch.int64)]
        if rank == 0:
            emit({"dry_run": True, "n_gpus": world, "total_streams": int(t.item()),
                  "scaling": args.scaling, "config": args.config or "1200",
                  "shards": [[int(a[0]), int(a[1])] for a in table],
                  "launcher": os.environ.get("TORCHELASTIC_RUN_ID", "") != "" or world == 1})
        if world > 1:
            dist.destroy_process_group()
        return

    import torch
    import minimodem_amd as M

    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.gather_self:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    ctx = M.Context(local_rank)

    name = args.config or "1200"
    line = run_workload(name, args, M, torch, dist, ctx, rank, world, args.steps, args.warmup,
                        cpu_leg=not args.no_cpu, oracle_leg=not args.no_cpu)
    if args.config is None and not args.no_extra:
        # the other BASELINE entries at their stated per-GPU sizes, the same K and W (with passes in
        # flight a handful of passes would mostly measure the pipeline filling and draining), device
        # generator; CPU leg: the reference program on 64 streams of each (its stdout against the
        # GPU's): driver-visible kernel time, roofline fraction and reference comparison per entry
        extra = {}
        for other in ("1200noise", "12000", "same", "rtty"):	# (shortest kernels first, the 12 ms one last)
            try:
                sub = run_workload(other, args, M, torch, dist, ctx, rank, world,
                                   args.steps, args.warmup, cpu_leg=False if args.no_cpu else "ref",
                                   oracle_leg=not args.no_cpu)
            except RankFailed as e:
                # (raised on EVERY rank by agree(): nobody is left inside a collective)
                sub = {"error": repr(e)} if rank == 0 else None
                torch.cuda.empty_cache()
            if rank == 0 and sub is not None:
                if "error" in sub:
                    extra[other] = sub
                    continue
                rf = sub["roofline"]
                extra[other] = {
                    "workload": sub["config"]["workload"], "value": sub["value"], "unit": sub["unit"],
                    "steps": sub["steps"], "ms_per_step": sub["ms_per_step"],
                    "value_serial": sub["value_serial"], "cold_ms_per_step": sub["cold_ms_per_step"],
                    "kernel": rf["kernel"], "kernel_ms_avg": rf["kernel_ms_avg"],
                    "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                    "roofline": {"bound": "hbm", "achieved": rf["achieved"], "peak": rf["peak"],
                                 "unit": rf["unit"], "frac": rf["frac"], "traffic": rf["traffic"]},
                    "launch": rf["launch"],
                    "pipeline": {k: sub["pipeline"][k] for k in ("passes_in_flight", "asked_for", "hw_queues", "ms_per_pass",
                                                                 "hbm_frac_of_the_timed_passes",
                                                                 "output_sets_equal_to_serial_launch")},
                    "payload_roundtrip_ok_streams": sub["payload_roundtrip_ok_streams"],
                    "oracle_mismatching_streams": sub.get("oracle_mismatching_streams"),
                    "oracle": sub.get("oracle"),
                }
                if "cpu_baseline" in sub:
                    # the reference PROGRAM on a sample of this workload too (64 streams)
                    extra[other]["cpu_baseline"] = sub["cpu_baseline"]
                    extra[other]["reference_mismatching_streams"] = sub["cpu_baseline"].get("mismatching_streams")
                if "per_rank" in sub:
                    extra[other]["per_rank"] = sub["per_rank"]
                if "payload_by_condition" in sub:
                    extra[other]["payload_by_condition"] = sub["payload_by_condition"]
                if "payload_roundtrip_ok_streams_gathered_from_peers" in sub:
                    extra[other]["payload_roundtrip_ok_streams_gathered_from_peers"] = \
                        sub["payload_roundtrip_ok_streams_gathered_from_peers"]
        if rank == 0:
            line["configs"] = extra
    if rank == 0:
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
