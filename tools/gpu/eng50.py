"""Engine comparison on a long-window, linear-lattice mode (50 baud at 48 kHz: 960 samples per
bit): python tools/gpu/eng50.py   (on the GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import minimodem_amd as M

ctx = M.Context(0)
for mode, n, secs in (("50", 2048, 20.0), ("150", 2048, 10.0)):
    cfg = M.rx_config(mode)
    nsamp = int(secs * cfg.sample_rate)
    frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
    nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 41) / frame) - 2
    rng = np.random.default_rng(5)
    words = torch.from_numpy(rng.integers(32, 127, size=(n, nwords), dtype=np.uint8)).cuda()
    lead = torch.from_numpy(rng.integers(0, 41, size=n).astype(np.int32)).cuda()
    d, lens = M.synthesize_batch(ctx, cfg, words, stride=(nsamp + 3) & ~3, leading_silence=lead, amplitude=0.8)
    outs = {}
    for eng in ("wave", "workgroup"):
        for _ in range(2):
            out = M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes",), engine=eng)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            M.demod_batch(ctx, cfg, d, nsamples=lens, want=("bytes",), out=out, engine=eng)
        e1.record(); torch.cuda.synchronize()
        outs[eng] = (out["bytes"].cpu().numpy().copy(), out["nbytes"].cpu().numpy().copy())
        print("%s baud (%d samples per bit), %d streams x %.0f s, %s engine: %.2f ms  [%s]"
              % (mode, cfg.bit_nsamples, n, secs, eng, e0.elapsed_time(e1) / 3,
                 M.demod_plan(ctx, cfg, n, engine=eng)["kernel"]))
    assert (outs["wave"][1] == outs["workgroup"][1]).all()
    for i in range(n):
        k = outs["wave"][1][i]
        assert (outs["wave"][0][i, :k] == outs["workgroup"][0][i, :k]).all()
