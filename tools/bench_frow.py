#!/usr/bin/env python3
"""Timing of the kernels either side of the hot path on one GPU, with events:
S16 ingest (2 B read + 4 B written per sample), the device transmitter (4 B written per
sample), and the receive loop with --auto-carrier (fsk_detect_carrier inside the loop) on the
bench.py batch.   python tools/bench_frow.py      (rocprofv3 --kernel-trace --stats -- python ... )"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import minimodem_amd as M

HBM_PEAK = 8.0e12


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)) * 1e-3


def main():
    ctx = M.Context(0)
    cfg = M.rx_config("1200")
    ns, n = 1024, 480000
    rng = np.random.default_rng(0)
    words = torch.from_numpy(rng.integers(32, 127, size=(ns, 1195), dtype=np.uint8)).cuda()
    lead = torch.from_numpy(rng.integers(0, 41, size=ns).astype(np.int32)).cuda()
    # transmitter: float and S16-valued output
    t = timed(lambda: M.synthesize_batch(ctx, cfg, words, stride=n, leading_silence=lead))
    print("tx_synth_kernel     %d x %d samples: %.3f ms  %.2f TB/s written (%.1f %% of the HBM roofline)"
          % (ns, n, t * 1e3, ns * n * 4 / t / 1e12, 100 * ns * n * 4 / t / HBM_PEAK))
    x, lens = M.synthesize_batch(ctx, cfg, words, stride=n, leading_silence=lead, s16=True)
    pcm = torch.round(x * 32768.0).to(torch.int16)
    t = timed(lambda: M.ingest_s16(ctx, pcm, nsamples=lens, stride=n))
    print("ingest_s16_kernel   %d x %d samples: %.3f ms  %.2f TB/s moved (2 B in + 4 B out per sample; %.1f %% of the HBM roofline)"
          % (ns, n, t * 1e3, ns * n * 6 / t / 1e12, 100 * ns * n * 6 / t / HBM_PEAK))
    y = M.ingest_s16(ctx, pcm, nsamples=lens, stride=n)
    assert torch.equal(y, x)
    # the receive loop with the tone looked for inside it
    for label, kw in (("fixed tones", {}), ("--auto-carrier", dict(auto_carrier_threshold=0.001))):
        c = M.rx_config("1200", **kw)
        out = M.demod_batch(ctx, c, x, nsamples=lens, want=("bytes",), engine="wave")
        t = timed(lambda: M.demod_batch(ctx, c, x, nsamples=lens, want=("bytes",), out=out, engine="wave"))
        nb = out["nbytes"].cpu().numpy()
        print("demod_wave_kernel   %-15s %.3f ms  %.3e samples/s  (%d bytes decoded)"
              % (label, t * 1e3, float(lens.sum()) / t, int(nb.sum())))


if __name__ == "__main__":
    main()
