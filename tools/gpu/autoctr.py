"""Work counters of the receive loop with --auto-carrier on the bench.py batch (profile build):
MIFSK_LIBRARY=$PWD/minimodem_amd/libmifsk_prof.so python tools/gpu/autoctr.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import minimodem_amd as M

ctx = M.Context(0)
n = 1024
for auto in (0.0, 0.001):
    cfg = M.rx_config("1200", auto_carrier_threshold=auto) if auto else M.rx_config("1200")
    host = np.zeros((n, bench.NSAMPLES), np.float32)
    for i in range(n):
        x, _ = bench.make_stream(M, M.rx_config("1200"), i)
        host[i, :len(x)] = x
    d = torch.from_numpy(host).cuda()
    for _ in range(2):
        out = M.demod_batch(ctx, cfg, d, want=("bytes", "counters"), engine="wave")
    torch.cuda.synchronize()
    c = out["counters"].cpu().numpy().astype(np.float64)
    print("auto_carrier =", auto, " nbytes mean", out["nbytes"].float().mean().item())
    for idx, name in sorted(M.COUNTER_NAMES.items()):
        col = c[:, idx]
        if col.max() > 0:
            print("  %-16s mean %12.1f  min %12.0f  max %12.0f" % (name, col.mean(), col.min(), col.max()))
    nd = out["counters"].cpu().numpy().view(np.uint64)[:, 22] & np.uint64(0xFFFFFFFF)
    print("  n_detect mean %.1f max %.0f" % (nd.mean(), nd.max()))
