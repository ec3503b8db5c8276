#!/usr/bin/env python3
"""BASELINE configs[4]: NOAA SAME with an additive-noise SNR sweep and the reference's
DC-offset sweep (tests/40-noise.test), on one GPU.  Reports, per condition, the share
of streams whose whole payload is decoded without a single byte error, and checks
a sample of streams against the oracle (identical buffers -> identical bytes).
    python tools/same_snr_sweep.py [--streams N]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import _oracle as O          # the checker
import minimodem_amd as M


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=512)
    args = ap.parse_args()
    ctx = M.Context(0)
    cfg = M.rx_config("same")
    ocfg = O.oracle_config("same")
    rng = np.random.default_rng(40)
    n, nwords = args.streams, 120
    words = rng.integers(32, 127, size=(n, nwords), dtype=np.uint8)
    clean, lens = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda(), amplitude=0.5)
    nsamp = int(lens[0])
    p_sig = 0.5 ** 2 / 2
    conds = [("clean", None, 0.0)] + [("SNR %2d dB" % s, s, 0.0) for s in (20, 12, 9, 6, 3)] \
        + [("DC offset %.2f" % d, None, d) for d in (0.05, 0.10, 0.50)]
    print("%-16s %12s %14s" % ("condition", "error-free", "== oracle"))
    for name, snr, dc in conds:
        x = clean.clone()
        if snr is not None:
            sigma = float(np.sqrt(p_sig / 10 ** (snr / 10)))
            g = torch.Generator(device="cuda"); g.manual_seed(1000 + snr)
            x[:, :nsamp] += sigma * torch.randn((n, nsamp), generator=g, device="cuda")
        if dc:
            M.ingest_rxnoise(ctx, x, dc, nsamples=lens)     # the reference's --Xrxnoise term
        res = M.results_to_host(M.demod_batch(ctx, cfg, x, nsamples=lens, want=("bytes",)))
        ok = 0
        for i in range(n):
            got = res["bytes"][i, :int(res["nbytes"][i])].tobytes()
            ok += words[i].tobytes() in got                 # the whole 120-byte payload, no error
        same = 0
        host = x.cpu().numpy()
        sample = range(0, n, max(1, n // 16))
        for i in sample:
            ref = O.oracle_rx_stream(ocfg, host[i, :nsamp])
            same += ref["bytes"] == res["bytes"][i, :int(res["nbytes"][i])].tobytes()
        print("%-16s %7d/%-4d %9d/%d" % (name, ok, n, same, len(sample)))


if __name__ == "__main__":
    main()
