#!/usr/bin/env python3
"""Print the kernel's per-stream work counters for a bench.py workload (diagnostic; run on the
GPU box with the profile build):
    MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so python tools/counters.py --config same"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="1200", choices=sorted(bench.WORKLOADS))
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--engine", default=None)
    args = ap.parse_args()
    import torch
    import minimodem_amd as M
    ctx = M.Context(0)
    entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[args.config]
    n = args.streams or per_gpu
    cfg = M.rx_config(mode)
    nsamp = bench.NSAMPLES if args.config in ("1200", "1200noise") else int(seconds * cfg.sample_rate)
    d, lens = bench.make_batch(args.config, M, torch, ctx, cfg, 0, 0, n, nsamp, (nsamp + 3) & ~3, amplitude, [None] * n)
    for _ in range(2):
        out = M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes", "counters"), engine=args.engine)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes", "counters"), out=out, engine=args.engine); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    raw0 = out["counters"].cpu().numpy().view(np.uint64)
    # the worker's three counters carry the staging phase's parts in their high words
    # (profile build, linear rounds): wait for the round's samples / registers -> LDS /
    # issue of the next round's loads
    hi = {13: "w_stage_wait", 14: "w_stage_write", 15: "w_stage_issue",
          17: "solo_fine_load", 18: "solo_fine_correlate", 19: "solo_fine_score", 21: "solo_fine_select"}
    hiv = {k: (raw0[:, k] >> np.uint64(32)).astype(np.float64) for k in hi}
    c = (raw0 & np.uint64(0xFFFFFFFF)).astype(np.float64)
    for k in range(c.shape[1]):
        if k not in hi:
            c[:, k] = raw0[:, k].astype(np.float64)
    nf = out["nframes"].cpu().numpy()
    print("%s: streams %d  frames/stream mean %.1f  kernel %.3f ms" % (args.config, n, nf.mean(), float(np.median(ts))))
    for idx, name in sorted(M.COUNTER_NAMES.items()):
        col = c[:, idx]
        if col.max() > 0:
            print("%-16s mean %12.1f  min %12.0f  max %12.0f" % (name, col.mean(), col.min(), col.max()))
    for k, name in hi.items():
        if hiv[k].max() > 0:
            print("%-16s mean %12.1f  min %12.0f  max %12.0f" % (name, hiv[k].mean(), hiv[k].min(), hiv[k].max()))
    raw = out["counters"].cpu().numpy().view(np.uint64)
    if (raw[:, 23] != 0).any():
        # where the streams sit inside the launch: start and end of each on the chip-wide
        # 100 MHz clock (s_memrealtime), its XCD from HW_REG_XCC_ID
        end = (raw[:, 23] & np.uint64(0xFFFFFFFF)).astype(np.float64) * 1e-2      # microseconds
        start = (raw[:, 23] >> np.uint64(32)).astype(np.float64) * 1e-2
        end = np.where(end < start, end + 2.0 ** 32 * 1e-2, end)                    # (32-bit wrap)
        xcc = (raw[:, 22] >> np.uint64(32)).astype(np.int64)
        t0 = start.min()
        print("streams per XCD:", np.bincount(xcc, minlength=8).tolist())
        print("stream starts %.1f ... %.1f us after the first, ends %.1f ... %.1f us (mean %.1f); events: %.1f us"
              % ((start - t0).min(), (start - t0).max(), (end - t0).min(), (end - t0).max(), (end - t0).mean(),
                 float(np.median(ts)) * 1e3))
        dur = end - start
        print("stream duration min %.1f mean %.1f max %.1f us; resident streams on average %.0f"
              % (dur.min(), dur.mean(), dur.max(), dur.sum() / (end.max() - t0)))
        refs = c[:, 4]
        for lo_, hi_ in ((0, 1), (2, 3), (4, 6), (7, 9), (10, 1 << 30)):
            m = (refs >= lo_) & (refs <= hi_)
            if m.any():
                print("  %4d streams with %s refinements: duration mean %.1f max %.1f us"
                      % (m.sum(), ("%d-%d" % (lo_, hi_)) if hi_ < 1 << 20 else ">= %d" % lo_, dur[m].mean(), dur[m].max()))
        for xcd in range(8):
            m = xcc == xcd
            if m.any():
                print("  XCD %d: last end %.1f us, mean duration %.1f us" % (xcd, (end[m] - t0).max(), dur[m].mean()))
    tot = c[:, 8].mean()
    if tot > 0:
        for idx in (9, 10, 11, 12, 13, 14, 16):
            print("  %-14s %5.1f%% of the mean stream's cycles" % (M.COUNTER_NAMES[idx], 100 * c[:, idx].mean() / tot))


if __name__ == "__main__":
    main()
