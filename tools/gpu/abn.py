#!/usr/bin/env python3
"""Same-box, same-process A/B/... of several builds of libmifsk on ONE resident batch (run on the
GPU box):

    python tools/gpu/abn.py --config 1200 --libs base,main,x1 --rounds 5 --steps 10

`--libs` names builds minimodem_amd/libmifsk_<tag>.so (`main` = libmifsk.so).  The batch of
bench.py's workload `--config` is synthesized once; the builds are timed alternately (events
around K launches per round).  Before timing, every build's full output (frame records,
episodes, bytes) is compared with the FIRST build's: a variant whose results differ from the
parity-green base is reported and not timed.  One process, one import of torch, one batch: a
tenth of the GPU-minutes `ab.sh` takes for the same comparison."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="1200")
    ap.add_argument("--libs", default="base,main")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--engine", default=None)
    ap.add_argument("--counters", action="store_true", help="print mean work counters per build")
    args = ap.parse_args()

    import torch
    import bench
    import minimodem_amd as M
    from minimodem_amd import _lib

    tags = args.libs.split(",")
    paths = [os.path.join(ROOT, "minimodem_amd", "libmifsk.so" if t == "main" else "libmifsk_%s.so" % t) for t in tags]
    for p in paths:
        assert os.path.exists(p), p

    libs, ctxs = {}, {}

    def use(tag, path):
        if tag not in libs:
            _lib._lib = None
            _lib.LIB_PATH = path
            libs[tag] = _lib.load()
            ctxs[tag] = M.Context(0)
        _lib._lib = libs[tag]
        return ctxs[tag]

    name = args.config
    entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
    ctx0 = use(tags[0], paths[0])
    cfg = M.rx_config(mode)
    n = args.streams or per_gpu
    nsamp = bench.NSAMPLES if name in ("1200", "1200noise") else int(seconds * cfg.sample_rate)
    stride = (nsamp + 3) & ~3
    payloads = [None] * n
    samples, lens = bench.make_batch(name, M, torch, ctx0, cfg, 0, 0, n, nsamp, stride, amplitude, payloads)
    torch.cuda.synchronize()
    frames_cap = M.max_frames(cfg, stride)
    total_bytes = 4.0 * float(n * nsamp if lens is None else int(lens.sum()))

    ref = None
    good = []
    for t, p in zip(tags, paths):
        ctx = use(t, p)
        out = M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes", "frames", "episodes"),
                            frames_cap=frames_cap, episodes_cap=64, engine=args.engine)
        torch.cuda.synchronize()
        res = M.results_to_host(out)
        del out
        sig = {}
        nf = res["nframes"].astype(np.int64)
        ne = np.minimum(res["nepisodes"].astype(np.int64), 64)
        nb = res["nbytes"].astype(np.int64)
        import hashlib
        h = hashlib.sha256()
        for i in range(n):
            h.update(res["frames"][i, :nf[i]].tobytes())
            h.update(res["episodes"][i, :ne[i]].tobytes())
            h.update(res["bytes"][i, :nb[i]].tobytes())
        sig = (h.hexdigest(), int(nf.sum()), int(ne.sum()), int(nb.sum()))
        if ref is None:
            ref = sig
        same = sig == ref
        print("%-10s results %s  (%d frames, %d episodes, %d bytes)%s"
              % (t, "== first build" if same else "DIFFER from first build", sig[1], sig[2], sig[3],
                 "" if same else "   <-- NOT TIMED"), flush=True)
        if same:
            good.append((t, p))
        if args.counters:
            oc = M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes", "counters"),
                               frames_cap=frames_cap, engine=args.engine)
            torch.cuda.synchronize()
            c = oc["counters"].cpu().numpy().view(np.uint64).astype(np.float64)
            print("           " + "  ".join("%s %.1f" % (M.COUNTER_NAMES[k], c[:, k].mean())
                                            for k in range(8) if c[:, k].max() > 0), flush=True)
            del oc

    kw = dict(want=("bytes",), frames_cap=frames_cap, nsamples=lens, engine=args.engine, episodes_cap=8)
    bufs = {}
    for t, p in good:
        ctx = use(t, p)
        bufs[t] = M.demod_batch(ctx, cfg, samples, **kw)
        for _ in range(2):
            M.demod_batch(ctx, cfg, samples, out=bufs[t], **kw)
    torch.cuda.synchronize()
    times = {t: [] for t, _ in good}
    for r in range(args.rounds):
        order = good if r % 2 == 0 else good[::-1]
        for t, p in order:
            ctx = use(t, p)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                M.demod_batch(ctx, cfg, samples, out=bufs[t], **kw)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / args.steps)
    base = None
    summary = {}
    for t, _ in good:
        a = np.array(times[t])
        med = float(np.median(a))
        base = base or med
        summary[t] = med
        print("%-10s %s  ms/launch: median %.4f  min %.4f  max %.4f   frac %.3f   vs first %+.1f %%"
              % (t, name, med, a.min(), a.max(), total_bytes / (med * 1e-3) / bench.HBM_PEAK, 100.0 * (med / base - 1.0)),
              flush=True)
    print(json.dumps({"config": name, "streams": n, "ms_median": summary}))


if __name__ == "__main__":
    main()
