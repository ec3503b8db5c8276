"""Experiment: what a plain launch looks like from inside before / after a chained one (profile build)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MIFSK_EXPERIMENT"] = "1"
os.environ["MIFSK_CHAIN"] = "0,0"
import torch
import minimodem_amd as M
import bench
name = "rtty"
entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
ctx = M.Context(0)
cfg = M.rx_config(mode)
n = per_gpu
nsamp = int(seconds * cfg.sample_rate)
stride = (nsamp + 3) & ~3
wl = [bench.stream_words(name, cfg, i, nsamp) for i in range(n)]
words = np.stack([w for w, _ in wl])
lead = torch.tensor([l for _, l in wl], dtype=torch.int32).cuda()
samples, lens = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda(), stride=stride, leading_silence=lead, amplitude=amplitude)
torch.cuda.synchronize()
fc = M.max_frames(cfg, stride)
outc = M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes", "counters"), frames_cap=fc)
outp = M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes",), frames_cap=fc)
torch.cuda.synchronize()

def look(label):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes", "counters"), frames_cap=fc, out=outc); e1.record()
    torch.cuda.synchronize()
    raw = outc["counters"].cpu().numpy().view(np.uint64)
    end = (raw[:, 23] & np.uint64(0xFFFFFFFF)).astype(np.float64) * 1e-2
    start = (raw[:, 23] >> np.uint64(32)).astype(np.float64) * 1e-2
    dur = end - start
    cyc = raw[:, 8].astype(np.float64)
    span = end.max() - start.min()
    c = outc["counters"].cpu().numpy().astype(np.float64)
    print("   counters:", {M.COUNTER_NAMES.get(i, i): round(float(c[:, i].mean()), 1) for i in range(24) if i not in (22, 23)})
    print("%-28s kernel %.2f ms; stream duration mean %.0f max %.0f us; resident on average %.0f; cycles per stream %.3g = %.2f GHz; XCDs %s"
          % (label, e0.elapsed_time(e1), dur.mean(), dur.max(), dur.sum() / span, cyc.mean(), (cyc / dur).mean() * 1e-3,
             np.bincount(((raw[:, 22] >> np.uint64(32)) & np.uint64(15)).astype(int), minlength=8).tolist()), flush=True)

def plain_time(label, ctx=ctx):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); M.demod_batch(ctx, cfg, samples, nsamples=lens, want=("bytes",), frames_cap=fc, out=outp); e1.record()
    torch.cuda.synchronize()
    print("%-28s %.2f ms" % (label, e0.elapsed_time(e1)), flush=True)

import ctypes as C
from minimodem_amd import _lib
def tabs(label, c=ctx):
    o = (C.c_ulonglong * 16)()
    _lib.load().mifsk_debug_tables(c.handle, C.byref(cfg), o)
    print(label, [hex(v) for v in o], flush=True)
tabs("tables before")
look("before")
plain_time("plain")
os.environ["MIFSK_CHAIN"] = sys.argv[1] if len(sys.argv) > 1 else "2,8"
plain_time("chained")
plain_time("chained")
os.environ["MIFSK_CHAIN"] = "0,0"
plain_time("plain")
os.environ["MIFSK_VERIFY_TABLES"] = "1"
plain_time("plain")
del os.environ["MIFSK_VERIFY_TABLES"]
look("after")

tabs("tables after ")
print(M.demod_plan(ctx, cfg, n, nsamples=stride))
ctx2 = M.Context(0)
plain_time("plain, new context", ctx2)
plain_time("plain, new context", ctx2)
plain_time("plain, old context")
tabs("tables ctx2  ", ctx2)
print(M.demod_plan(ctx2, cfg, n, nsamples=stride))
