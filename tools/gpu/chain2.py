"""Experiment: the library's own chained launches, toggled inside one process."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MIFSK_EXPERIMENT"] = "1"
os.environ["MIFSK_CHAIN"] = "0,0"
import torch
import minimodem_amd as M
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "rtty"
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
kw = dict(want=("bytes",), frames_cap=fc, nsamples=lens, episodes_cap=8)
ref = M.demod_batch(ctx, cfg, samples, **kw)
torch.cuda.synchronize()
ref_b = ref["bytes"].clone(); ref_n = ref["nbytes"].clone()

def timed(label, stream=None, reps=4):
    def fn():
        M.demod_batch(ctx, cfg, samples, out=ref, **kw)
    st = stream or torch.cuda.current_stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
    ok = bool((ref["nbytes"] == ref_n).all()) and bool((ref["bytes"] == ref_b).all())
    print("%-50s %.3f ms  same=%s" % (label, e0.elapsed_time(e1) / reps, ok), flush=True)

timed("plain, before any chain")
side = torch.cuda.Stream()
timed("plain on a side stream", side)
os.environ["MIFSK_CHAIN"] = sys.argv[2] if len(sys.argv) > 2 else "2,8"
timed("chained from a side stream", side)
timed("chained from the null stream")
os.environ["MIFSK_CHAIN"] = "0,0"
timed("plain again (null stream)")
timed("plain again (side stream)", side)
