#!/usr/bin/env python3
"""Randomised parity soak (run on the GPU box): many streams per mode with random
payloads, amplitudes, leading silence, additive noise, DC offset, clipping, rate
slop (resampled TX rate) and truncation; every frame record and episode from the
device must equal the oracle's bit for bit.   python tools/soak.py [--seed N] [--streams N]"""
import argparse
from concurrent.futures import ThreadPoolExecutor
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O          # noqa: E402  (the checker)
import minimodem_amd as M    # noqa: E402

MODES = [("1200", {}), ("300", {}), ("12000", {}), ("same", {}), ("rtty", {}), ("tdd", {}),
         ("1200", dict(n_data_bits=7)), ("1200", dict(msb_first=1)), ("2400", {}),
         ("1200", dict(sample_rate=44100)), ("600", dict(nstopbits=2.0)),
         ("1200", dict(auto_carrier_threshold=0.001)), ("300", dict(auto_carrier_threshold=0.001)),
         # long windows: SCAN with shared segments (wave engine), other bit lengths than RTTY's
         ("rtty", dict(sample_rate=44100)), ("50", {}), ("rtty", dict(auto_carrier_threshold=0.001))]


def make_stream(rng, cfg, mode):
    slow = mode in ("rtty", "tdd", "50")
    nwords = int(rng.integers(1, 12 if slow else 120))
    hi = 1 << min(8, int(cfg.n_data_bits))
    words = rng.integers(0, hi, size=nwords, dtype=np.uint8)
    # rate slop: transmit at a slightly different baud rate than the receiver expects
    txcfg = cfg
    if rng.random() < 0.3:
        txcfg = M.rx_config(mode if not mode.replace(".", "").isdigit()
                            else str(float(mode) * float(rng.uniform(0.985, 1.015))))
        if txcfg.n_data_bits != cfg.n_data_bits:
            txcfg = cfg
    x = M.synthesize(txcfg, words, amplitude=float(rng.uniform(0.05, 1.0)),
                     leading_silence=int(rng.integers(0, 3000)), s16=bool(rng.integers(0, 2)))
    r = rng.random()
    if r < 0.5:
        x = x + rng.normal(0, float(rng.choice([0.001, 0.02, 0.1, 0.3])), x.shape)
    if rng.random() < 0.15:
        x = x + float(rng.uniform(-0.3, 0.3))              # DC offset
    if rng.random() < 0.1:
        x = np.clip(x, -0.2, 0.2)                          # clipping
    if rng.random() < 0.15:
        x = x[: int(len(x) * rng.uniform(0.2, 0.95))]      # cut mid-stream
    if rng.random() < 0.05:
        x = np.concatenate([x, np.zeros(int(rng.integers(1, 20000))), x])   # two bursts
    return np.ascontiguousarray(x, dtype=np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--streams", type=int, default=96)
    ap.add_argument("--engine", default="wave", choices=["wave", "workgroup"])
    ap.add_argument("--ring", action="store_true", help="RING addressing (wave engine), oracle ring_mode")
    ap.add_argument("--slabs", type=int, default=0,
                    help="feed every stream in this many pieces, cut at random places, through "
                         "mifsk_demod_slab (SlabSession) instead of one mifsk_demod_batch call")
    ap.add_argument("--chain", action="store_true",
                    help="force chained launches (MIFSK_EXPERIMENT, MIFSK_CHAIN = a random groups x "
                         "chunks cut per configuration) onto these small batches: wave engine, flat")
    ap.add_argument("--no-episodes", action="store_true",
                    help="ask for bytes and frames only: the lattice replay then leaves the episodes' "
                         "running totals out (replay_scan_*, totals == false)")
    args = ap.parse_args()
    import torch
    ctx = M.Context(0)
    rng = np.random.default_rng(args.seed)
    total_frames = bad = chained = 0
    for mode, kw in MODES:
        if args.engine == "workgroup" and kw.get("auto_carrier_threshold"):
            continue                    # --auto-carrier runs on the wavefront engine only
        cfg = M.rx_config(mode, **kw)
        ocfg = O.oracle_config(mode, **kw)
        streams = [make_stream(rng, cfg, mode) for _ in range(args.streams)]
        maxn = max(len(s) for s in streams)
        stride = (maxn + 3) & ~3
        host = np.zeros((len(streams), stride), np.float32)
        lens = np.zeros(len(streams), np.int32)
        for i, s in enumerate(streams):
            host[i, :len(s)] = s
            lens[i] = len(s)
        t = time.time()
        cut = ""
        if args.chain:
            os.environ["MIFSK_EXPERIMENT"] = "1"
            os.environ["MIFSK_CHAIN"] = "%d,%d" % (int(rng.integers(1, 4)), int(rng.integers(2, 12)))
            p = M.demod_plan(ctx, cfg, len(streams), engine=args.engine, ring_exact=args.ring, nsamples=stride)
            cut = " cut %dx%d" % (p["chain_groups"], p["chain_chunks"]) if p["chain_groups"] else " (not cut)"
            chained += 1 if p["chain_groups"] else 0
        if args.slabs:
            # every stream cut at its own random places; the pieces' outputs concatenated
            cuts = [sorted(int(c) for c in rng.integers(0, len(s) + 1, size=args.slabs - 1)) for s in streams]
            sess = M.SlabSession(ctx, cfg, len(streams), episodes_cap=64, ring_exact=args.ring,
                                 engine=None if args.ring or kw.get("auto_carrier_threshold") else args.engine)
            acc = [dict(frames=[], episodes=[], bytes=b"") for _ in streams]
            for k in range(args.slabs):
                new = []
                for i, s in enumerate(streams):
                    e = [0] + cuts[i] + [len(s)]
                    new.append(s[e[k]:e[k + 1]])
                r = sess.feed(new, final=(k == args.slabs - 1))
                for i in range(len(streams)):
                    n, ne = int(r["nframes"][i]), int(r["nepisodes"][i])
                    acc[i]["frames"].append(r["frames"][i, :n].copy())
                    acc[i]["episodes"].append(r["episodes"][i, :ne].copy())
                    acc[i]["bytes"] += r["bytes"][i, :int(r["nbytes"][i])].tobytes()
        else:
            res = M.results_to_host(M.demod_batch(ctx, cfg, torch.from_numpy(host).cuda(),
                                                  nsamples=torch.from_numpy(lens).cuda(),
                                                  want=("bytes", "frames") if args.no_episodes else ("bytes", "frames", "episodes"),
                                                  episodes_cap=64,
                                                  engine=args.engine, ring_exact=args.ring))
        # the oracle over every stream on all host cores (ofsk_rx_stream is re-entrant, ctypes
        # releases the GIL)
        def check(i):
            s = streams[i]
            ref = O.oracle_rx_stream(ocfg, s, ring_mode=args.ring)
            if args.slabs:
                fr = np.concatenate(acc[i]["frames"])
                ep = np.concatenate(acc[i]["episodes"])
                n, ne = len(fr), len(ep)
                ok = (fr.tobytes() == ref["frames"].tobytes() and ep.tobytes() == ref["episodes"].tobytes()
                      and acc[i]["bytes"] == ref["bytes"])
            else:
                n = int(res["nframes"][i])
                ne = 0 if args.no_episodes else int(res["nepisodes"][i])
                ok = (n == len(ref["frames"])
                      and res["frames"][i, :n].tobytes() == ref["frames"].tobytes()
                      and res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"])
                if not args.no_episodes:
                    ok = ok and ne == len(ref["episodes"]) and res["episodes"][i, :ne].tobytes() == ref["episodes"].tobytes()
            return ok, n, len(ref["frames"])

        nf = 0
        with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as ex:
            for i, (ok, n, nref) in enumerate(ex.map(check, range(len(streams)))):
                if not ok:
                    bad += 1
                    print("MISMATCH mode %s %r stream %d (len %d): gpu %d frames, oracle %d"
                          % (mode, kw, i, len(streams[i]), n, nref))
                nf += n
        total_frames += nf
        print("%-6s %-40s %4d streams %7d frames  %.1f s%s" % (mode, kw, len(streams), nf, time.time() - t, cut), flush=True)
    print("seed %d (%s engine, %s addressing%s): %d frames compared, %d mismatching streams"
          % (args.seed, args.engine, "ring" if args.ring else "flat",
             ", %d slabs per stream" % args.slabs if args.slabs else
             ", %d configurations chained" % chained if args.chain else "", total_frames, bad)
          + (" (no episode records)" if args.no_episodes else ""))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
