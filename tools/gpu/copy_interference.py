#!/usr/bin/env python3
"""8-GPU readiness without the node: what does a CU-resident copy cost the
demod launch it shares the chip with?  At N = 8 the root receives 7 peers' decoded bytes per step
through RCCL copy kernels while its own demod kernel runs (configs[3]: 7 x 19.7 MB per 1.3 ms
step; configs[1]: 7 x 1.2 MB per 0.45 ms).  Here: the config's kernel alone, then with a
device-to-device copy of that many bytes per step running concurrently on a second stream -- a
plain torch copy kernel, which takes as many CUs as it likes: an upper bound on what RCCL's few
channel workgroups do.  Round 5: also as SEVEN copies on seven streams (one per peer, each a
peer's share), which is how the grouped irecv's arrive: seven concurrent kernels instead of one."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import torch
import minimodem_amd as M

ctx = M.Context(0)
out = {}
for name in ("1200", "12000"):
    entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
    cfg = M.rx_config(mode)
    nsamp = 480000 if name == "1200" else int(seconds * cfg.sample_rate)
    rng = np.random.default_rng(9)
    frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
    nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 41) / frame) - 2
    words = torch.from_numpy(rng.integers(0x20, 0x7F, size=(per_gpu, nwords), dtype=np.uint8)).cuda()
    lead = torch.from_numpy(rng.integers(0, 41, size=per_gpu).astype(np.int32)).cuda()
    x, lens = M.synthesize_batch(ctx, cfg, words, stride=(nsamp + 3) & ~3, leading_silence=lead)
    cols = int(M.max_frames(cfg, nsamp))
    gather_bytes = 7 * (per_gpu * cols + 4 * per_gpu)
    src = torch.empty(gather_bytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    peers = [torch.cuda.Stream() for _ in range(7)]
    per_peer = gather_bytes // 7
    res = M.demod_batch(ctx, cfg, x, nsamples=lens, want=("bytes",))
    torch.cuda.synchronize()

    def once(with_copy):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        if with_copy == 1:
            with torch.cuda.stream(side):
                for _ in range(2):          # (keeps copying while the kernel runs)
                    dst.copy_(src)
        elif with_copy == 7:
            for k, st in enumerate(peers):
                with torch.cuda.stream(st):
                    for _ in range(2):
                        dst[k * per_peer:(k + 1) * per_peer].copy_(src[k * per_peer:(k + 1) * per_peer])
        e0.record()
        M.demod_batch(ctx, cfg, x, nsamples=lens, want=("bytes",), out=res)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    # (out of the idle state first -- the launches after a pause run 6-8 % slower, bench.py's
    # preheat -- then the three variants in turn, so that what drift is left is shared)
    for _ in range(400 if name == "1200" else 150):
        M.demod_batch(ctx, cfg, x, nsamples=lens, want=("bytes",), out=res)
    torch.cuda.synchronize()
    ts = {0: [], 1: [], 7: []}
    for _ in range(16):
        for mode_ in (0, 1, 7):
            ts[mode_].append(once(mode_))
    alone, shared, seven = (float(np.median(ts[k])) for k in (0, 1, 7))
    out[name] = {"kernel_ms_alone": alone, "kernel_ms_beside_copy": shared, "slowdown": shared / alone,
                 "kernel_ms_beside_7_copies": seven, "slowdown_7_streams": seven / alone,
                 "copy_bytes_per_step": gather_bytes, "bytes_per_peer": per_peer}
    print(name, json.dumps(out[name]), flush=True)
    del x, src, dst
    torch.cuda.empty_cache()
