#!/usr/bin/env python3
"""Per-condition cycle breakdown of the workgroup engine on bench.py's `1200noise` batch (profile
build; run on the GPU box):
    MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so python tools/gpu/noise_ctr.py [--streams N]
Prints, per impairment, the mean per stream of the event counts and of the master's cycle totals,
and the cycles per refinement left after subtracting the clean streams' chain."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=0)
    ap.add_argument("--config", default="1200noise")
    args = ap.parse_args()
    import torch
    import minimodem_amd as M
    ctx = M.Context(0)
    name = args.config
    entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
    n = args.streams or per_gpu
    cfg = M.rx_config(mode)
    nsamp = bench.NSAMPLES
    d, lens = bench.make_batch(name, M, torch, ctx, cfg, 0, 0, n, nsamp, (nsamp + 3) & ~3, amplitude, [None] * n)
    for _ in range(3):
        out = M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes", "counters"))
    torch.cuda.synchronize()
    raw = out["counters"].cpu().numpy().view(np.uint64)
    lo = (raw & np.uint64(0xFFFFFFFF)).astype(np.float64)
    hi = (raw >> np.uint64(32)).astype(np.float64)
    full = raw.astype(np.float64)
    conds = bench.CONDITIONS.get(name, [("snr_db", None)])
    cols = [("iter", full[:, 0]), ("refines", full[:, 4]), ("hits", full[:, 5]), ("lat_batches", full[:, 7]),
            ("stages", full[:, 2]), ("total", full[:, 8]), ("scan", full[:, 9]), ("wait", full[:, 10]),
            ("conf", full[:, 11]), ("bulk", full[:, 12]), ("general", full[:, 16]), ("restart", lo[:, 17]),
            ("scan1", lo[:, 18]), ("scan2", lo[:, 19]), ("replay", full[:, 20]), ("scanwait", lo[:, 21]),
            ("sf_load", hi[:, 17]), ("sf_corr", hi[:, 18]), ("sf_score", hi[:, 19]), ("sf_sel", hi[:, 21]),
            ("b_pre", full[:, 25]), ("b_mid", full[:, 26]), ("b_out", lo[:, 27]), ("b_pass", hi[:, 27]),
            ("w_stage", lo[:, 13]), ("w_corr", lo[:, 14]), ("w_barrier", lo[:, 15])]
    print("%-14s" % "condition" + "".join("%10s" % c for c, _ in cols))
    rows = {}
    for k, (kind, v) in enumerate(conds):
        m = np.arange(n) % len(conds) == k
        rows[k] = [float(col[m].mean()) for _, col in cols]
        print("%-14s" % bench.condition_label(kind, v) + "".join("%10.0f" % x for x in rows[k]))
    base = rows[0]
    ti = [c for c, _ in cols].index("total")
    ri = [c for c, _ in cols].index("refines")
    for k, (kind, v) in enumerate(conds):
        if k and rows[k][ri] > base[ri] + 1:
            print("%-14s cycles per extra refinement: %.0f" % (bench.condition_label(kind, v),
                  (rows[k][ti] - base[ti]) / (rows[k][ri] - base[ri])))


if __name__ == "__main__":
    main()
