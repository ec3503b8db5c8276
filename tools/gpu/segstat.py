"""Shared-segment SCAN statistics on an RTTY-shaped batch: scans through the segment path,
index-order fallback passes, per stream (event counters 20 / 21)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import minimodem_amd as M

mode = sys.argv[1] if len(sys.argv) > 1 else "rtty"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ctx = M.Context(0)
cfg = M.rx_config(mode)
n, secs = 1024, 30.0
nsamp = int(secs * cfg.sample_rate)
frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 41) / frame) - 2
rng = np.random.default_rng(5)
five = cfg.n_data_bits == 5
words = torch.from_numpy(rng.integers(0 if five else 32, 32 if five else 127, size=(n, nwords), dtype=np.uint8)).cuda()
lead = torch.from_numpy(rng.integers(0, 41, size=n).astype(np.int32)).cuda()
d, lens = M.synthesize_batch(ctx, cfg, words, stride=(nsamp + 3) & ~3, leading_silence=lead, amplitude=0.8)
if sigma:
    d += torch.randn(d.shape, device="cuda") * sigma
out = M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes", "counters"))
torch.cuda.synchronize()
c = out["counters"].cpu().numpy().astype(np.float64)
print("%s sigma %.2f: per stream: iterations %.1f scans %.1f positions %.1f refines %.1f | shared-segment scans %.1f, index-order fallback passes %.2f (%.1f%% of the scans)"
      % (mode, sigma, c[:, 0].mean(), c[:, 1].mean(), c[:, 6].mean(), c[:, 4].mean(), c[:, 20].mean(), c[:, 21].mean(),
         100 * c[:, 21].sum() / max(1.0, c[:, 20].sum())))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
o2 = M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes",))
torch.cuda.synchronize()
e0.record()
for _ in range(3):
    M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes",), out=o2)
e1.record(); torch.cuda.synchronize()
print("  %d streams x %.0f s: %.2f ms  [%s]" % (n, secs, e0.elapsed_time(e1) / 3, M.demod_plan(ctx, cfg, n)["kernel"]))
