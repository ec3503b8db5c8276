#!/usr/bin/env python3
"""Diagnostic (round 4): is configs[1]'s kernel waiting for HBM, or is the per-stream chain the
bound?  The same 1024 Bell-202 streams cut to 30 000 ... 480 000 samples: the short batches
(123 / 246 MB) stay resident in the 256 MB Infinity Cache between launches, the long ones stream
from HBM.  If the time per sample is the same in both regimes, HBM latency is not what the
workers wait for.  Prints ms, ns per stream-sample and the slope between consecutive sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
import minimodem_amd as M


def timed(ctx, cfg, d, reps=12):
    out = M.demod_batch(ctx, cfg, d, want=("bytes",))
    for _ in range(3):
        M.demod_batch(ctx, cfg, d, want=("bytes",), out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); M.demod_batch(ctx, cfg, d, want=("bytes",), out=out); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.min(ts)), float(np.median(ts))


def main():
    n = int(os.environ.get("STREAMS", "1024"))
    ctx = M.Context(0); cfg = M.rx_config("1200")
    host = np.zeros((n, bench.NSAMPLES), np.float32)
    for i in range(n):
        x, _ = bench.make_stream(M, cfg, i); host[i, :len(x)] = x
    full = torch.from_numpy(host).cuda()
    prev = None
    for ns in (30000, 60000, 120000, 240000, 480000):
        d = full[:, :ns].contiguous()
        lo, med = timed(ctx, cfg, d)
        line = "%7d samples x %d streams (%6.1f MB): min %.4f median %.4f ms  %.3f ns/sample/stream-slot" % (
            ns, n, n * ns * 4 / 1e6, lo, med, med * 1e6 / ns)
        if prev:
            line += "   slope %.3f ns per extra sample (%.2f TB/s marginal)" % (
                (med - prev[1]) * 1e6 / (ns - prev[0]), n * (ns - prev[0]) * 4 / ((med - prev[1]) * 1e-3) / 1e12)
        print(line, flush=True)
        prev = (ns, med)
        del d
    # the engines side by side on the full batch
    for eng in ("workgroup", "wave"):
        out = M.demod_batch(ctx, cfg, full, want=("bytes",), engine=eng)
        torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); M.demod_batch(ctx, cfg, full, want=("bytes",), out=out, engine=eng); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        print("engine %-9s full batch: median %.4f ms" % (eng, float(np.median(ts))), flush=True)


if __name__ == "__main__":
    main()
