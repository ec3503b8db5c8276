#!/usr/bin/env python3
"""What pipelining consecutive passes buys: the same batch launched K times on ONE stream, and
alternately on P streams (one context each, one output set each), so that pass i + 1 fills the
CUs that pass i's late streams leave idle.  Outputs of every set compared with the serial run's."""
import argparse, os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import minimodem_amd as M

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="1200")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--pipes", default="1,2,3,4")
ap.add_argument("--copies", default="shared", choices=["shared", "separate", "contiguous"],
                help="what the lanes read: one batch, a clone each, or slices of one big allocation")
args = ap.parse_args()
ctx0 = M.Context(0)
P = max(int(p) for p in args.pipes.split(","))
ctxs = [ctx0] + [M.Context(0) for _ in range(P - 1)]
streams = [torch.cuda.Stream() for _ in range(P)]
for name in args.config.split(","):
  entry, mode, per_gpu, seconds, _, amplitude = bench.WORKLOADS[name]
  cfg = M.rx_config(mode)
  n = per_gpu
  nsamp = bench.NSAMPLES if name in ("1200", "1200noise") else int(seconds * cfg.sample_rate)
  stride = (nsamp + 3) & ~3
  payloads = [None] * n
  samples, lens = bench.make_batch(name, M, torch, ctx0, cfg, 0, 0, n, nsamp, stride, amplitude, payloads)
  torch.cuda.synchronize()		# (the batch is made on torch's current stream)
  frames_cap = M.max_frames(cfg, stride)
  kw = dict(want=("bytes",), frames_cap=frames_cap, nsamples=lens, episodes_cap=8)
  if args.copies == "separate":
      src = [samples] + [samples.clone() for _ in range(P - 1)]
  elif args.copies == "contiguous":
      big = torch.empty((P,) + tuple(samples.shape), dtype=samples.dtype, device=samples.device)
      big[:] = samples
      src = [big[k] for k in range(P)]
  else:
      src = [samples] * P
  torch.cuda.synchronize()
  bufs = []
  for c, st, x in zip(ctxs, streams, src):
      bufs.append(M.demod_batch(c, cfg, x, stream=st, **kw))
  torch.cuda.synchronize()
  ref = M.results_to_host(bufs[0])
  for b in bufs[1:]:
      r = M.results_to_host(b)
      assert np.array_equal(r["nbytes"], ref["nbytes"]) and np.array_equal(r["bytes"], ref["bytes"])
  total_bytes = 4.0 * float(n * nsamp if lens is None else int(lens.sum()))
  # out of the idle state
  for i in range(300 if name.startswith("1200") else 40):
      M.demod_batch(ctxs[0], cfg, samples, stream=streams[0], out=bufs[0], **kw)
  torch.cuda.synchronize()
  for p in [int(x) for x in args.pipes.split(",")]:
      ts = []
      for rep in range(5):
          torch.cuda.synchronize()
          t0 = time.perf_counter()
          for i in range(args.steps):
              k = i % p
              M.demod_batch(ctxs[k], cfg, src[k], stream=streams[k], out=bufs[k], **kw)
          torch.cuda.synchronize()
          ts.append((time.perf_counter() - t0) / args.steps * 1e3)
      med = float(np.median(ts))
      for k in range(p):
          r = M.results_to_host(bufs[k])
          assert np.array_equal(r["nbytes"], ref["nbytes"]) and np.array_equal(r["bytes"], ref["bytes"]), "pipelined output differs"
      print("%s  %d stream(s): %.4f ms per pass (min %.4f)  %.1f %% of 8 TB/s  x%.2f" % (name, p, med, min(ts), 100 * total_bytes / (med * 1e-3) / bench.HBM_PEAK, 0 if p == 1 else base / med) if p > 1 else
            "%s  1 stream: %.4f ms per pass (min %.4f)  %.1f %% of 8 TB/s" % (name, med, min(ts), 100 * total_bytes / (med * 1e-3) / bench.HBM_PEAK), flush=True)
      if p == 1:
          base = med
