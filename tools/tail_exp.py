#!/usr/bin/env python3
"""Diagnostic: how much do refine events cost?  Times batches made of copies of
one stream (fewest / median / most refines) against the mixed benchmark batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import torch
import minimodem_amd as M

def timed(ctx, cfg, d, reps=6):
    M.demod_batch(ctx, cfg, d, want=("bytes",))
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); M.demod_batch(ctx, cfg, d, want=("bytes",)); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts), sum(ts) / len(ts)

def main():
    n = 1024
    ctx = M.Context(0); cfg = M.rx_config("1200")
    host = np.zeros((n, bench.NSAMPLES), np.float32)
    for i in range(n):
        x, _ = bench.make_stream(M, cfg, i); host[i, :len(x)] = x
    d = torch.from_numpy(host).cuda()
    out = M.demod_batch(ctx, cfg, d, want=("bytes", "counters"))
    ref = out["counters"][:, 4].cpu().numpy()
    print("refines: min %d median %d max %d mean %.2f" % (ref.min(), np.median(ref), ref.max(), ref.mean()))
    print("mixed batch            min %.4f avg %.4f ms" % timed(ctx, cfg, d))
    for name, idx in (("fewest", int(ref.argmin())), ("median", int(np.argsort(ref)[n // 2])), ("most", int(ref.argmax()))):
        dd = torch.from_numpy(np.repeat(host[idx:idx + 1], n, axis=0)).cuda()
        print("1024 x stream %4d (%2d refines, %s)  min %.4f avg %.4f ms" % ((idx, ref[idx], name) + timed(ctx, cfg, dd)))
        if os.environ.get("MIFSK_LIBRARY"):
            c = M.demod_batch(ctx, cfg, dd, want=("bytes", "counters"))["counters"].cpu().numpy().astype(float).mean(axis=0)
            print("    " + "  ".join("%s %.0f" % (M.COUNTER_NAMES[k], c[k]) for k in sorted(M.COUNTER_NAMES)))

if __name__ == "__main__":
    main()
