#!/usr/bin/env python3
"""Bell-202 batches that are not one round of workgroups (1024 on an MI355X): one launch of
demod_kernel<true,10,2> against chained launches of its resumable twin (G groups x K chunks),
the library's own choice last.  Streams made on the device; bytes compared with the plain run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MIFSK_EXPERIMENT"] = "1"
import torch
import minimodem_amd as M

ctx = M.Context(0); cfg = M.rx_config("1200")
nmax = int(os.environ.get("NMAX", "5120")); nsamp = 480000
rng = np.random.default_rng(3)
nwords = (nsamp - 41 - 6 * 40) // 400 - 2
words = torch.from_numpy(rng.integers(0x20, 0x7F, size=(nmax, nwords), dtype=np.uint8)).cuda()
lead = torch.from_numpy(rng.integers(0, 41, size=nmax).astype(np.int32)).cuda()
x, lens = M.synthesize_batch(ctx, cfg, words, stride=nsamp, leading_silence=lead)
torch.cuda.synchronize()

def timed(n, chain):
    if chain is None:
        os.environ.pop("MIFSK_CHAIN", None)
    else:
        os.environ["MIFSK_CHAIN"] = chain
    d = x[:n]
    plan = M.demod_plan(ctx, cfg, n, nsamples=nsamp)
    out = M.demod_batch(ctx, cfg, d, want=("bytes",))
    for _ in range(2):
        M.demod_batch(ctx, cfg, d, want=("bytes",), out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); M.demod_batch(ctx, cfg, d, want=("bytes",), out=out); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), out, plan

for n in [int(v) for v in os.environ.get("SIZES", "1024 1536 2048 3000 4096 5000").split()]:
    base_ms, base_out, _ = timed(n, "0,0")
    ref_b = base_out["bytes"].clone(); ref_n = base_out["nbytes"].clone()
    line = "%5d streams: one launch %.3f ms (%.1f %% of 8 TB/s)" % (n, base_ms, n * nsamp * 4 / (base_ms * 1e-3) / 8e12 * 100)
    for chain in ("2,2", "2,3", "2,4", "3,3", None):
        ms, out, plan = timed(n, chain)
        same = bool(torch.equal(out["nbytes"], ref_n) and torch.equal(out["bytes"], ref_b))
        tag = chain if chain is not None else "library (%d,%d)" % (plan["chain_groups"], plan["chain_chunks"])
        line += " | %s: %.3f%s" % (tag, ms, "" if same else " MISMATCH")
    print(line, flush=True)
