#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one GPU (diagnostic; the
driver's metric is bench.py = configs[1]).  Streams are generated on the device
(mifsk_tx_synthesize_batch), timed with events, and decoded words checked against
what was transmitted.   python tools/bench_configs.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minimodem_amd as M

HBM_PEAK = 8.0e12
# name, mode, streams on one GPU, seconds of audio, word range
CONFIGS = [
    ("configs[1] Bell202 1200 baud", "1200", 1024, 10.0, (32, 127)),
    ("configs[2] RTTY 45.45 baud", "rtty", 4096, 30.0, (0, 32)),
    ("configs[3] 12000 baud (one GPU's 8192 of 65536)", "12000", 8192, 2.0, (32, 127)),
    ("configs[4] NOAA SAME 520.83 baud", "same", 8192, 10.0, (32, 127)),
    ("300 baud (Bell 103)", "300", 1024, 10.0, (32, 127)),
]


def main():
    ctx = M.Context(0)
    rng = np.random.default_rng(1)
    for name, mode, nstreams, seconds, (lo, hi) in CONFIGS:
        cfg = M.rx_config(mode)
        nsamp = int(seconds * cfg.sample_rate)
        frame = float(cfg.frame_n_bits) * cfg.nsamples_per_bit
        frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
        nwords = int((nsamp - 6 * cfg.nsamples_per_bit - (16 * frame if cfg.do_rx_sync else 0)) / frame) - 2
        words = torch.from_numpy(rng.integers(lo, hi, size=(nstreams, nwords), dtype=np.uint8)).cuda()
        stride = (nsamp + 3) & ~3
        x, n = M.synthesize_batch(ctx, cfg, words, stride=stride)
        assert int(n.max()) <= stride, (int(n.max()), stride)
        out = M.demod_batch(ctx, cfg, x, nsamples=n, want=("bits",))
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); M.demod_batch(ctx, cfg, x, nsamples=n, want=("bits",), out=out); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        res = M.results_to_host(out)
        w = words.cpu().numpy()
        ok = 0
        for i in range(0, nstreams, max(1, nstreams // 64)):
            got = res["bits"][i, :int(res["nframes"][i])].astype(np.uint8).tobytes()
            ok += w[i].tobytes() in got
        total = float(n.sum())
        ms = float(np.median(ts))
        print("%-48s %5d streams x %8d samples: %8.3f ms  %.3e samples/s  %.1f %% of HBM roofline  "
              "(payload found in %d/%d sampled streams)"
              % (name, nstreams, nsamp, ms, total / ms * 1e3, 100 * total * 4 / (ms * 1e-3) / HBM_PEAK,
                 ok, len(range(0, nstreams, max(1, nstreams // 64)))))
        del x, out, words
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
