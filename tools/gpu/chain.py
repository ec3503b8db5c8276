"""Experiment: a batch cut into G groups of streams x K time chunks, each (group, chunk) one launch
of the resumable kernel (mifsk_demod_slab), the groups on separate HIP streams -- the hardware
dispatcher fills the slots one group's stragglers leave with the other group's next chunk.
Timing only (outputs restart at index 0 per launch).  chain.py <config> [G K] ..."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import minimodem_amd as M
from minimodem_amd import _lib
import bench

name = sys.argv[1]
combos = [(int(a), int(b)) for a, b in zip(sys.argv[2::2], sys.argv[3::2])] or [(1, 1), (2, 4)]
entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
ctx = M.Context(0)
cfg = M.rx_config(mode)
n = per_gpu
nsamp = bench.NSAMPLES if name == "1200" else int(seconds * cfg.sample_rate)
stride = (nsamp + 3) & ~3
wl = [bench.stream_words(name, cfg, i, nsamp) for i in range(n)]
words = np.stack([w for w, _ in wl])
lead = torch.tensor([l for _, l in wl], dtype=torch.int32).cuda()
samples, lens = M.synthesize_batch(ctx, cfg, torch.from_numpy(words).cuda(), stride=stride,
                                   leading_silence=lead, amplitude=amplitude)
if name == "same":
    p_sig = amplitude ** 2 / 2
    g = torch.Generator(device="cuda"); g.manual_seed(1000)
    for k, (kind, v) in enumerate(bench.SAME_CONDITIONS):
        rows = samples[k::8]
        if kind == "snr_db" and v is not None:
            rows += torch.randn(rows.shape, generator=g, device="cuda", dtype=torch.float32) * float(np.sqrt(p_sig / 10 ** (v / 10)))
        elif kind == "dc":
            rows -= np.float32(v)
torch.cuda.synchronize()
fc = M.max_frames(cfg, stride)
kw = dict(want=("bytes",), frames_cap=fc, nsamples=lens, episodes_cap=8)
ref = M.demod_batch(ctx, cfg, samples, **kw)
torch.cuda.synchronize()

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

print("%s: %d streams x %d; plain demod_batch %.3f ms" % (name, n, nsamp, timed(lambda: M.demod_batch(ctx, cfg, samples, out=ref, **kw))))
lib = _lib.load()
dev = samples.device
state = torch.zeros((n, M.STATE_DTYPE.itemsize), dtype=torch.uint8, device=dev)
out = {"nframes": torch.zeros(n, dtype=torch.int32, device=dev), "status": torch.zeros(n, dtype=torch.int32, device=dev),
       "bytes": torch.zeros((n, fc), dtype=torch.uint8, device=dev), "nbytes": torch.zeros(n, dtype=torch.int32, device=dev)}

def chain(G, K):
    streams = [torch.cuda.Stream() for _ in range(G)]
    bounds = [n * g // G for g in range(G + 1)]
    limits = []
    for k in range(K):
        lim = torch.minimum(lens, torch.full_like(lens, (k + 1) * nsamp // K)) if k < K - 1 else lens
        limits.append(lim.contiguous())
    ios = []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        for k in range(K):
            io = _lib.DemodIO()
            io.d_samples = samples[lo].data_ptr()
            io.stream_stride = samples.stride(0)
            io.d_nsamples = limits[k][lo:].data_ptr()
            io.nsamples = stride
            io.nstreams = hi - lo
            io.d_bytes = out["bytes"][lo].data_ptr(); io.d_nbytes = out["nbytes"][lo:].data_ptr()
            io.d_nframes = out["nframes"][lo:].data_ptr(); io.frames_cap = fc
            io.d_status = out["status"][lo:].data_ptr()
            io.flags = 0
            ios.append((g, k, lo, io))
    def run():
        state.zero_()
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for k in range(K):
            for g in range(G):
                _, _, lo, io = ios[g * K + k]
                rc = lib.mifsk_demod_slab(ctx.handle, C.byref(cfg), C.byref(io), C.c_void_p(state[lo].data_ptr()), None,
                                          1 if k == K - 1 else 0, C.c_void_p(streams[g].cuda_stream))
                assert rc == 0, rc
        for s in streams:
            cur.wait_stream(s)
    return run

for G, K in combos:
    t = timed(chain(G, K))
    fin = int((state.cpu().numpy().view(M.STATE_DTYPE).reshape(n)["flags"] & 4 != 0).sum())
    print("  G=%d groups x K=%d chunks: %.3f ms   (finished streams %d/%d)" % (G, K, t, fin, n))
