#!/usr/bin/env python3
"""bench.py -- audio samples/sec demodulated, Bell-202 1200 baud, 48 kHz f32.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): per GPU, a batch of 1024 independent
synthetic streams x 10 s (480000 samples) of 48 kHz mono f32 Bell-202 audio,
generated the way `minimodem --tx` generates it (csrc/mifsk_tx.cpp), resident
in HBM before the timed region.  One "step" = one pass of the receive path over
the whole resident batch (mifsk_demod_batch: frame search + bit correlation +
receive loop on the device) plus, for N > 1, the gather of decoded bytes to
rank 0 over RCCL (overlapped with the next step).  Streams shard across ranks
with no data-path collective, so scaling is weak (1024 streams per GPU).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     : HBM-read roofline of the demod kernel, measured live with
                 events on the launch stream
  cpu_baseline : the reference's own CPU path (oracle/_ref: unmodified
                 src/*.c + FFT shim) timed on this box on a bounded sample
  cpu_port     : the oracle restatement (direct 2-bin DFT, 1 core) on the FULL
                 batch, whose output is also compared byte-for-byte with the GPU's
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NSTREAMS_PER_GPU = 1024
NSAMPLES = int(os.environ.get("MIFSK_BENCH_NSAMPLES", "480000"))	# 10 s at 48 kHz (override: experiments only)
HBM_PEAK = 8.0e12		# B/s, MI355X spec (MI355X_MICROARCH.md)


def make_stream(M, cfg, gid):
    """Stream `gid` of the synthetic batch: seeded printable payload, 0..40
    samples of leading silence, zero tail up to NSAMPLES."""
    rng = np.random.default_rng(1234 + gid)
    lead = int(rng.integers(0, 41))
    nbytes = (NSAMPLES - lead - 4 * 40) // 400
    payload = rng.integers(0x20, 0x7F, size=nbytes, dtype=np.uint8)
    x = M.synthesize(cfg, payload, leading_silence=lead)
    assert len(x) <= NSAMPLES
    return x, payload


def write_wav_f32(path, x, sr):
    import struct
    data = np.asarray(x, "<f4").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt ")
        f.write(struct.pack("<IHHIIHH", 16, 3, 1, sr, sr * 4, 4, 32))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


def cpu_baselines(host, payloads, gpu_bytes, gpu_nbytes):
    """Rank 0, N=1 only.  Times the CPU checkers on this box's host cores and
    cross-checks their output against the GPU's (parity at full size)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    ocfg = O.oracle_config("1200")
    out = {}

    # (1) the oracle restatement, one core, the FULL batch
    t0 = time.perf_counter()
    mismatches = 0
    for i in range(host.shape[0]):
        r = O.oracle_rx_stream(ocfg, host[i], ring_mode=False)
        if r["bytes"] != gpu_bytes[i, :gpu_nbytes[i]].tobytes() or r["bytes"] != payloads[i].tobytes():
            mismatches += 1
    dt = time.perf_counter() - t0
    out["cpu_port"] = {
        "value": host.size / dt, "unit": "samples/s", "cores": 1, "kind": "port",
        "sample": "all %d streams x %d samples through oracle/fsk_oracle.c (direct 2-bin DFT, "
                  "f64 fma), 1 thread; output compared byte-for-byte with the GPU's and with "
                  "the transmitted payload: %d mismatching streams" % (host.shape[0], host.shape[1],
                                                                      mismatches),
        "seconds": dt, "mismatching_streams": mismatches,
    }

    # (2) the reference program itself (unmodified src/*.c + shims) on a bounded sample
    if O.have_ref():
        nref = min(384, host.shape[0])		# ~13 s of single-core work
        tmp = tempfile.mkdtemp(prefix="mifsk-bench-")
        paths = []
        for i in range(nref):
            p = os.path.join(tmp, "s%03d.wav" % i)
            write_wav_f32(p, host[i], 48000)
            paths.append(p)
        t0 = time.perf_counter()
        bad = 0
        for i, p in enumerate(paths):
            r = subprocess.run([O.MINIMODEM_REF, "--rx", "--quiet", "--file", p, "1200"],
                               stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            if r.stdout != gpu_bytes[i, :gpu_nbytes[i]].tobytes():
                bad += 1
        dt = time.perf_counter() - t0
        for p in paths:
            os.unlink(p)
        os.rmdir(tmp)
        out["cpu_baseline"] = {
            "value": nref * host.shape[1] / dt, "unit": "samples/s", "cores": 1,
            "kind": "reference",
            "sample": "first %d streams x %d samples through oracle/_ref/minimodem_ref --rx --file "
                      "(reference src/*.c unmodified; FFTW3f absent in this image, FFT = oracle "
                      "double-precision shim), 1 process; stdout compared with the GPU's bytes: "
                      "%d mismatching streams" % (nref, host.shape[1], bad),
            "seconds": dt, "mismatching_streams": bad,
        }
    else:
        out["cpu_baseline"] = dict(out["cpu_port"])
    return out


def hbm_traffic():
    """HBM bytes per kernel launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
    note), as recorded in profiles/ by tools/profile_round.sh for this workload; None
    when no such record is committed."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        return {"bytes_per_launch": rec["hbm_bytes_per_launch"], "source": "profiles/r01_hbm_traffic.json",
                "fetch_size_kb_raw": rec["fetch_size_kb_raw"], "write_size_kb_raw": rec["write_size_kb_raw"]}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=NSTREAMS_PER_GPU, help="streams per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import torch
    import minimodem_amd as M

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    ctx = M.Context(local_rank)
    cfg = M.rx_config("1200")
    nstreams = args.streams
    total_streams = nstreams * world
    lo, hi = M.shard_range(total_streams, rank, world)
    assert hi - lo == nstreams

    # ---- synthetic batch (host, threaded; the generator releases the GIL) ----
    host = np.zeros((nstreams, NSAMPLES), np.float32)
    payloads = [None] * nstreams

    def gen(i):
        x, p = make_stream(M, cfg, lo + i)
        host[i, :len(x)] = x
        payloads[i] = p

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        list(ex.map(gen, range(nstreams)))
    samples = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()

    frames_cap = M.max_frames(cfg, NSAMPLES)
    want = ("bytes",)
    bufs = [M.demod_batch(ctx, cfg, samples, want=want, frames_cap=frames_cap) for _ in range(2)]
    torch.cuda.synchronize()

    pending = [None, None]

    def step(i, events=None):
        b = i & 1
        if pending[b] is not None:		# its buffers are about to be overwritten
            for w in pending[b]:
                w.wait()
            pending[b] = None
        if events is not None:
            events[0].record()
        M.demod_batch(ctx, cfg, samples, want=want, frames_cap=frames_cap, out=bufs[b])
        if events is not None:
            events[1].record()
        if world > 1:
            pending[b] = gather_async(bufs[b])

    gatherer = M.ByteGatherer(dist, rank, world)

    def gather_async(buf):
        """decoded bytes -> rank 0, grouped send/recv (each peer uses its own xGMI link)"""
        return gatherer.start(buf["bytes"], buf["nbytes"])

    def drain():
        for b in (0, 1):
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None

    for i in range(args.warmup):
        step(i)
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()

    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, evs[i])
    drain()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    res = M.results_to_host(bufs[(args.steps - 1) & 1]) if args.steps else M.results_to_host(bufs[0])
    gpu_bytes, gpu_nbytes = res["bytes"], res["nbytes"]
    ok_streams = sum(1 for i in range(nstreams)
                     if gpu_bytes[i, :gpu_nbytes[i]].tobytes() == payloads[i].tobytes())

    if rank == 0:
        samples_per_step = float(total_streams) * NSAMPLES
        value = samples_per_step * args.steps / dt
        kavg = float(np.mean(kernel_ms)) * 1e-3
        achieved = nstreams * NSAMPLES * 4.0 / kavg
        line = {
            "metric": "audio samples/sec demodulated (whole node), 1200-baud 48 kHz f32",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Bell202 1200-baud, 48 kHz f32, batch of %d synthetic streams "
                                   "x 480000 samples per GPU (BASELINE.json configs[1])" % nstreams,
                       "streams_per_gpu": nstreams, "samples_per_stream": NSAMPLES,
                       "sharding": "independent streams per rank, decoded bytes gathered to "
                                   "rank 0 over RCCL" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK, "traffic": hbm_traffic(),
                         "kernel": "mifsk::demod_kernel<true>",
                         "kernel_ms_avg": kavg * 1e3, "kernel_ms_min": float(np.min(kernel_ms)),
                         "algorithmic_bytes_per_launch": nstreams * NSAMPLES * 4.0},
            "payload_roundtrip_ok_streams": "%d/%d" % (ok_streams, nstreams),
            "device": ctx.device_name,
        }
        if world == 1 and not args.no_cpu:
            line.update(cpu_baselines(host, payloads, gpu_bytes, gpu_nbytes))
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
