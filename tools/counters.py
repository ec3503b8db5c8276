#!/usr/bin/env python3
"""Print the kernel's per-stream work counters for the benchmark workload
(diagnostic; run on the GPU box):  python tools/counters.py [--streams N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--mode", default="1200")
    args = ap.parse_args()
    import torch
    import minimodem_amd as M
    ctx = M.Context(0)
    cfg = M.rx_config(args.mode)
    host = np.zeros((args.streams, bench.NSAMPLES), np.float32)
    for i in range(args.streams):
        x, _ = bench.make_stream(M, cfg, i)
        host[i, :len(x)] = x
    d = torch.from_numpy(host).cuda()
    for _ in range(2):
        out = M.demod_batch(ctx, cfg, d, want=("bytes", "counters"))
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy().astype(np.float64)
    nf = out["nframes"].cpu().numpy()
    print("streams %d  frames/stream mean %.1f" % (args.streams, nf.mean()))
    for idx, name in sorted(M.COUNTER_NAMES.items()):
        col = c[:, idx]
        print("%-16s mean %12.1f  min %12.0f  max %12.0f" % (name, col.mean(), col.min(), col.max()))
    tot = c[:, 8].mean()
    for idx in (9, 10, 11, 12, 13, 14, 15):
        print("  %-14s %5.1f%% of kernel cycles" % (M.COUNTER_NAMES[idx], 100 * c[:, idx].mean() / tot))


if __name__ == "__main__":
    main()
