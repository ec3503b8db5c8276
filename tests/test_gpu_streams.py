"""Launches in flight on several HIP streams (bench.py --pipeline, INTEGRATION.md 2a'): six
contexts, six streams, the same batch launched on all of them back to back -- with the output
tensors allocated by the call itself -- must each give what a lone launch gives.  (Round 5: the
wrapper zero-filled its outputs on torch's CURRENT stream while the kernel ran on the stream it was
given; with the launches truly concurrent -- eight hardware queues -- a fill landed on a kernel's
results.)  With the fills put back on the current stream the script below fails under both
queue settings (tools/gpu/racecheck.py: ('rtty', 1, 1, 'nframes'), ('1200', 2, 5, 'nbytes')).
In a subprocess: GPU_MAX_HW_QUEUES is read when HIP starts."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.environ["MIFSK_ROOT"])
import minimodem_amd as M

torch.manual_seed(5)
rng = np.random.default_rng(5)
for mode, nstreams, seconds in (("1200", 384, 2.0), ("rtty", 96, 6.0)):
    cfg = M.rx_config(mode)
    ctxs = [M.Context(0) for _ in range(6)]
    streams = [torch.cuda.Stream() for _ in range(6)]
    nsamp = int(seconds * cfg.sample_rate)
    frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
    nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 64) / frame) - 2
    hi = 0x7F if mode == "1200" else 0x20
    words = torch.from_numpy(rng.integers(0x01 if mode == "rtty" else 0x20, hi, size=(nstreams, nwords), dtype=np.uint8)).cuda()
    lead = torch.from_numpy(rng.integers(0, 41, size=nstreams).astype(np.int32)).cuda()
    x, lens = M.synthesize_batch(ctxs[0], cfg, words, stride=(nsamp + 3) & ~3, leading_silence=lead)
    # (every third stream noisy: unequal stream durations, refinements)
    x[::3] += 0.18 * torch.randn_like(x[::3])
    torch.cuda.synchronize()
    want = ("bytes", "frames", "episodes")
    ref = M.results_to_host(M.demod_batch(ctxs[0], cfg, x, nsamples=lens, want=want, episodes_cap=16))
    assert int(ref["nframes"].sum()) > nstreams * nwords // 2
    for rep in range(3):
        outs = [M.demod_batch(c, cfg, x, nsamples=lens, want=want, episodes_cap=16, stream=s)
                for c, s in zip(ctxs, streams)]
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            r = M.results_to_host(o)
            for key in ("nframes", "nbytes", "nepisodes", "status"):
                assert np.array_equal(r[key], ref[key]), (mode, rep, k, key)
            for i in range(nstreams):
                nf, nb, ne = int(ref["nframes"][i]), int(ref["nbytes"][i]), min(16, int(ref["nepisodes"][i]))
                assert r["frames"][i, :nf].tobytes() == ref["frames"][i, :nf].tobytes(), (mode, rep, k, i)
                assert r["bytes"][i, :nb].tobytes() == ref["bytes"][i, :nb].tobytes(), (mode, rep, k, i)
                assert r["episodes"][i, :ne].tobytes() == ref["episodes"][i, :ne].tobytes(), (mode, rep, k, i)
            # what the call allocated is zero where the kernel wrote nothing
            assert not r["bytes"][np.arange(r["bytes"].shape[1])[None, :] >= ref["nbytes"][:, None]].any(), (mode, rep, k)
    del ctxs
print("STREAMS_OK")
'''


@pytest.mark.parametrize("queues", ["4", "8"])
def test_six_launches_in_flight_on_six_streams_equal_a_lone_launch(queues):
    env = dict(os.environ)
    env.update(GPU_MAX_HW_QUEUES=queues, MIFSK_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0 and b"STREAMS_OK" in r.stdout, r.stderr.decode()[-3000:]
