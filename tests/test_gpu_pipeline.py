"""mifsk_pipeline_* (include/mifsk.h "several batches in flight"): passes submitted through the
C ABI's pipeline -- lanes of context + stream + library-owned output sets -- give, pass for pass,
what a serial mifsk_demod_batch gives: counts, frame records, episodes, bytes, on all five
bench workloads (Bell-202 clean and under impairments, 12000 baud, NOAA SAME, RTTY) at reduced
size.  What it stands in for at the reference's call site: one file after another through the
batch entry (/root/reference/src/minimodem.c:1265,1373 via integration/minimodem-rx-batch.patch).
In a subprocess: GPU_MAX_HW_QUEUES is read when HIP starts, and the depth in effect depends on it."""
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

queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
asked = int(os.environ["MIFSK_TEST_DEPTH"])
pipe = M.Pipeline(0, depth=asked)
info = pipe.info()
assert info["depth_requested"] == asked and info["hw_queues"] == queues, info
assert info["depth"] == max(1, min(asked, queues - 1)) == pipe.depth, info
assert info["output_sets"] == 0
ctx = M.Context(0)
rng = np.random.default_rng(11)
torch.manual_seed(11)
WORK = (("1200", 192, 1.5, 0.0), ("1200", 192, 1.5, 0.25), ("12000", 256, 0.4, 0.05),
        ("same", 128, 2.0, 0.12), ("rtty", 48, 6.0, 0.08))
for mode, nstreams, seconds, sigma in WORK:
    cfg = M.rx_config(mode)
    nsamp = int(seconds * cfg.sample_rate)
    stride = (nsamp + 3) & ~3
    frame = (cfg.n_data_bits + cfg.nstartbits + cfg.nstopbits) * cfg.nsamples_per_bit
    nwords = int((nsamp - 6 * cfg.nsamples_per_bit - 64 - (16 * frame if cfg.do_rx_sync else 0)) / frame) - 2
    five = cfg.n_data_bits == 5
    batches = []
    for b in range(2):                      # two different batches, submitted alternately
        words = torch.from_numpy(rng.integers(1 if five else 0x20, 0x20 if five else 0x7F,
                                              size=(nstreams, nwords), dtype=np.uint8)).cuda()
        lead = torch.from_numpy(rng.integers(0, 1 if mode == "same" else 41, size=nstreams).astype(np.int32)).cuda()
        x, lens = M.synthesize_batch(ctx, cfg, words, stride=stride, leading_silence=lead,
                                     amplitude=0.5 if mode == "same" else 1.0)
        if sigma:
            x[::2] += sigma * torch.randn_like(x[::2])
        batches.append((x, lens))
    torch.cuda.synchronize()
    fcap = M.max_frames(cfg, stride)
    want = ("bytes", "frames", "episodes")
    refs = [M.results_to_host(M.demod_batch(ctx, cfg, x, nsamples=lens, want=want, frames_cap=fcap, episodes_cap=16))
            for x, lens in batches]
    assert all(int(r["nframes"].sum()) > nstreams * nwords // 3 for r in refs), mode
    pipe.outputs(nstreams, fcap, episodes_cap=16, want=want)
    assert pipe.info()["output_sets"] == pipe.depth
    npass = 3 * pipe.depth + 1
    got = {}
    tickets = []
    for t in range(npass):
        if t >= pipe.depth:                 # the set pass t writes is the one pass t - depth wrote: read it first
            old = tickets[t - pipe.depth]
            pipe.wait(old)
            got[old] = M.results_to_host(pipe.result(old))
        x, lens = batches[t & 1]
        tk = pipe.submit(cfg, x, nsamples=lens)
        assert tk == (tickets[-1] + 1 if tickets else pipe.next_ticket() - 1)
        tickets.append(tk)
    pipe.drain()
    for tk in tickets[-pipe.depth:]:
        got[tk] = M.results_to_host(pipe.result(tk))
    for t, tk in enumerate(tickets):
        r, ref = got[tk], refs[t & 1]
        for key in ("nframes", "nbytes", "nepisodes", "status"):
            assert np.array_equal(r[key], ref[key]), (mode, sigma, t, key)
        for i in range(nstreams):
            nf, nb, ne = int(ref["nframes"][i]), int(ref["nbytes"][i]), min(16, int(ref["nepisodes"][i]))
            assert r["frames"][i, :nf].tobytes() == ref["frames"][i, :nf].tobytes(), (mode, sigma, t, i)
            assert r["bytes"][i, :nb].tobytes() == ref["bytes"][i, :nb].tobytes(), (mode, sigma, t, i)
            assert r["episodes"][i, :ne].tobytes() == ref["episodes"][i, :ne].tobytes(), (mode, sigma, t, i)
# join: a consumer stream ordered behind a pass on the device
x, lens = batches[0]
side = torch.cuda.Stream()
tk = pipe.submit(cfg, x, nsamples=lens)
pipe.join(tk, side)
with torch.cuda.stream(side):
    total = pipe.result(tk)["nframes"].sum()
side.synchronize()
assert int(total) == int(refs[0]["nframes"].sum())
# a ticket that was never issued
try:
    pipe.wait(pipe.next_ticket() + 5)
    raise SystemExit("wait() accepted a ticket that was never issued")
except RuntimeError:
    pass
pipe.close()
print("PIPELINE_OK depth", info["depth"])
'''


@pytest.mark.parametrize("queues,depth", [("4", 3), ("4", 5), ("8", 4), ("8", 1)])
def test_pipelined_passes_equal_serial_launches_on_all_five_workloads(queues, depth):
    env = dict(os.environ)
    env.update(GPU_MAX_HW_QUEUES=queues, MIFSK_TEST_DEPTH=str(depth),
               MIFSK_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0 and b"PIPELINE_OK" in r.stdout, (r.stdout.decode()[-500:], r.stderr.decode()[-3000:])
