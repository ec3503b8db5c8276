import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import minimodem_amd as M
ctx = M.Context(0)
cfg = M.rx_config("1200")
x = M.synthesize(cfg, b"hello world, this is more than eight frames")
d = torch.from_numpy(np.pad(x, (0, (-len(x)) % 4))[None, :]).cuda()
dl = torch.tensor([len(x)], dtype=torch.int32).cuda()
for fc, ec, want in ((8, 1, ("bytes", "episodes")), (64, 1, ("bytes", "episodes")), (8, 4, ("bytes", "episodes")),
                     (8, 1, ("bytes", "episodes", "counters")), (None, 8, ("bytes", "episodes", "frames", "counters"))):
    out = M.demod_batch(ctx, cfg, d, nsamples=dl, want=want, frames_cap=fc, episodes_cap=ec)
    torch.cuda.synchronize()
    r = M.results_to_host(out)
    print(fc, ec, "nframes", r["nframes"], "nbytes", r["nbytes"], "neps", r["nepisodes"], "status", r["status"],
          r.get("counters", [None])[0] if "counters" in r else "")
