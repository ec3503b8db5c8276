"""Experiment: which of the chained launch's ingredients slows every later kernel down."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MIFSK_EXPERIMENT"] = "1"
os.environ["MIFSK_CHAIN"] = "0,0"
import torch
import minimodem_amd as M
from minimodem_amd import _lib
import bench

hip = C.CDLL("libamdhip64.so")
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
kw = dict(want=("bytes",), frames_cap=fc, nsamples=lens, episodes_cap=8)
ref = M.demod_batch(ctx, cfg, samples, **kw)
torch.cuda.synchronize()

def timed(label, reps=3):
    def fn():
        M.demod_batch(ctx, cfg, samples, out=ref, **kw)
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-60s %.3f ms" % (label, e0.elapsed_time(e1) / reps), flush=True)

timed("plain, at the start")
streams = []
for i in range(3):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    streams.append(s)
timed("after creating three non-blocking streams")
evs = []
for i in range(4):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0
    evs.append(e)
for s in streams:
    for e in evs:
        hip.hipStreamWaitEvent(s, e, 0)
timed("after waiting on unrecorded events")
p = C.c_void_p()
assert hip.hipMalloc(C.byref(p), 4096 * 96) == 0
for s in streams:
    hip.hipMemsetAsync(p, 0, 4096 * 96, s)
torch.cuda.synchronize()
timed("after hipMalloc + memsets on those streams")
# the resumable kernel, whole streams, on one of those streams through the slab API
lib = _lib.load()
state = torch.zeros((n, M.STATE_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
out = {"nframes": torch.zeros(n, dtype=torch.int32, device="cuda"), "status": torch.zeros(n, dtype=torch.int32, device="cuda"),
       "bytes": torch.zeros((n, fc), dtype=torch.uint8, device="cuda"), "nbytes": torch.zeros(n, dtype=torch.int32, device="cuda")}
io = _lib.DemodIO()
io.d_samples = samples.data_ptr(); io.stream_stride = samples.stride(0); io.d_nsamples = lens.data_ptr()
io.nsamples = stride; io.nstreams = n
io.d_bytes = out["bytes"].data_ptr(); io.d_nbytes = out["nbytes"].data_ptr(); io.d_nframes = out["nframes"].data_ptr()
io.frames_cap = fc; io.d_status = out["status"].data_ptr(); io.flags = 0
def slab(stream_ptr):
    state.zero_(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    import time
    t = time.time()
    rc = lib.mifsk_demod_slab(ctx.handle, C.byref(cfg), C.byref(io), C.c_void_p(state.data_ptr()), None, 1, stream_ptr)
    assert rc == 0
    torch.cuda.synchronize()
    return (time.time() - t) * 1e3

def chain(streams, K=8, G=2):
    bounds = [n * g // G for g in range(G + 1)]
    limits = [(torch.minimum(lens, torch.full_like(lens, (k + 1) * nsamp // K)) if k < K - 1 else lens).contiguous() for k in range(K)]
    ios = {}
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        for k in range(K):
            o = _lib.DemodIO()
            o.d_samples = samples[lo].data_ptr(); o.stream_stride = samples.stride(0); o.d_nsamples = limits[k][lo:].data_ptr()
            o.nsamples = stride; o.nstreams = hi - lo
            o.d_bytes = out["bytes"][lo].data_ptr(); o.d_nbytes = out["nbytes"][lo:].data_ptr(); o.d_nframes = out["nframes"][lo:].data_ptr()
            o.frames_cap = fc; o.d_status = out["status"][lo:].data_ptr(); o.flags = 0
            ios[g, k] = (lo, o)
    fork, done = evs[0], evs[1:]
    def run():
        state.zero_()
        hip.hipEventRecord(fork, C.c_void_p(0))
        for g in range(G):
            hip.hipStreamWaitEvent(streams[g], fork, 0)
        for k in range(K):
            for g in range(G):
                lo, o = ios[g, k]
                rc = lib.mifsk_demod_slab(ctx.handle, C.byref(cfg), C.byref(o), C.c_void_p(state[lo].data_ptr()), None, 1 if k == K - 1 else 0, streams[g])
                assert rc == 0
        for g in range(G):
            hip.hipEventRecord(done[g], streams[g])
            hip.hipStreamWaitEvent(C.c_void_p(0), done[g], 0)
    run(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3

print("chain through the slab API on my own streams: %.3f ms" % chain(streams))
timed("plain after that")
ts = [torch.cuda.Stream() for _ in range(2)]
print("chain through the slab API on torch streams: %.3f ms" % chain([C.c_void_p(t.cuda_stream) for t in ts]))
timed("plain after that")
os.environ["MIFSK_CHAIN"] = "2,8"
timed("the library's chain")
os.environ["MIFSK_CHAIN"] = "0,0"
timed("plain after that")
