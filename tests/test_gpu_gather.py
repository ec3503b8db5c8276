"""mifsk_gather_* (csrc/mifsk_gather.cpp): decoded bytes to rank 0 behind the C ABI.  A test box
has ONE GPU, so the transport runs in loopback -- the one rank sends to itself through RCCL and
receives it as peer 0: the narrow staging copy, the grouped send / recv on the caller's stream,
the receive sets and their tickets.  (The same calls at world size > 1 post one receive per
peer instead; the shard arithmetic around them is covered on CPU: tests/test_distributed_cpu.py.)"""
import ctypes as C

import numpy as np
import pytest

import minimodem_amd as M
from minimodem_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _batch(torch, seed, n, cap):
    g = torch.Generator(device="cpu").manual_seed(seed)
    b = torch.randint(0, 256, (n, cap), dtype=torch.uint8, generator=g).cuda()
    nb = torch.randint(0, cap + 1, (n,), dtype=torch.int32, generator=g).cuda()
    return b, nb


def test_loopback_gather_delivers_the_columns_that_were_sent(torch_mod):
    torch = torch_mod
    g = M.NativeGatherer(None, 0, 1, cols=40, slots=3, loopback=True)
    info = g.info()
    assert info["world"] == 1 and info["loopback"] == 1 and info["communicator"] == 1 and info["slots"] == 3
    streams = [torch.cuda.Stream() for _ in range(3)]
    sent = []
    for k in range(3):                      # three gathers in flight on three streams, one set each
        b, nb = _batch(torch, 10 + k, 257, 64)
        streams[k].wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[k]):
            assert g.start(b, nb) == []
        sent.append((b, nb))
    torch.cuda.synchronize()
    for k in range(3):
        rb, rn = g.received(0, slot=k)
        assert rb.shape == (257, 40) and rn.shape == (257,)
        assert torch.equal(rb, sent[k][0][:, :40]) and torch.equal(rn, sent[k][1])
    # a fourth gather reuses set 0: gather 0's ticket is gone, the new one is there
    b, nb = _batch(torch, 99, 257, 64)
    g.start(b, nb)
    torch.cuda.synchronize()
    rb, rn = g.received(0)
    assert torch.equal(rb, b[:, :40]) and torch.equal(rn, nb)
    lib = _lib.load()
    pb, pn, rows, cols = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
    assert lib.mifsk_gather_received(g.handle, 0, 0, C.byref(pb), C.byref(pn), C.byref(rows), C.byref(cols)) == -22
    assert lib.mifsk_gather_received(g.handle, 7, 0, C.byref(pb), C.byref(pn), C.byref(rows), C.byref(cols)) == -22
    # full-width rows travel from where they are (no staging copy); a change of shape makes new sets
    b2, nb2 = _batch(torch, 5, 31, 48)
    g.cols = 48
    g.start(b2, nb2)
    torch.cuda.synchronize()
    rb, rn = g.received(0)
    assert rb.shape == (31, 48) and torch.equal(rb, b2) and torch.equal(rn, nb2)
    g.close()


def test_a_pipeline_pass_and_its_gather_on_the_lanes_stream(torch_mod):
    """bench.py's step: a pass of the library's pipeline, its bytes gathered on the lane's stream
    behind it, several in flight -- what arrives is what a lone launch decodes."""
    torch = torch_mod
    ctx = M.Context(0)
    cfg = M.rx_config("1200")
    rng = np.random.default_rng(3)
    rows = []
    for i in range(48):
        words = rng.integers(32, 127, size=30 + i % 7, dtype=np.uint8)
        rows.append(M.synthesize(cfg, words, leading_silence=int(rng.integers(0, 40))))
    width = (max(len(r) for r in rows) + 3) & ~3
    host = np.zeros((len(rows), width), np.float32)
    lens = np.zeros(len(rows), np.int32)
    for i, r in enumerate(rows):
        host[i, :len(r)] = r
        lens[i] = len(r)
    x, n = torch.from_numpy(host).cuda(), torch.from_numpy(lens).cuda()
    fc = int(M.max_frames(cfg, width))
    lone = M.results_to_host(M.demod_batch(ctx, cfg, x, nsamples=n, want=("bytes",), frames_cap=fc))
    cols = int(lone["nbytes"].max())
    pipe = M.Pipeline(0, depth=2)
    pipe.outputs(len(rows), fc, want=("bytes",))
    g = M.NativeGatherer(None, 0, 1, cols=cols, slots=pipe.depth, loopback=True)
    for i in range(5):
        tk = pipe.submit(cfg, x, nsamples=n)
        out = pipe.result(tk)
        with torch.cuda.stream(pipe.stream(tk)):
            g.start(out["bytes"], out["nbytes"])
    pipe.drain()
    torch.cuda.synchronize()
    for slot in range(pipe.depth):
        rb, rn = g.received(0, slot=slot)
        assert np.array_equal(rn.cpu().numpy(), lone["nbytes"])
        assert np.array_equal(rb.cpu().numpy(), lone["bytes"][:, :cols])
    g.close()
    pipe.close()
    ctx.close()


def test_world_of_one_without_loopback_makes_no_communicator(torch_mod):
    torch = torch_mod
    g = M.NativeGatherer(None, 0, 1)
    assert g.info()["communicator"] == 0
    b, nb = _batch(torch, 1, 8, 16)
    assert g.start(b, nb) == []
    with pytest.raises(RuntimeError):
        g.received(0)
    g.close()


def test_argument_errors(torch_mod):
    torch = torch_mod
    lib = _lib.load()
    h = C.c_void_p()
    ident = (C.c_ubyte * _lib.GATHER_ID_BYTES)()
    assert lib.mifsk_gather_unique_id(None) == -22
    assert lib.mifsk_gather_unique_id(ident) == 0 and any(bytes(ident))
    assert lib.mifsk_gather_create(C.byref(h), ident, 1, 1, -1, 2, 0) == -22           # rank outside the world
    assert lib.mifsk_gather_create(C.byref(h), None, 0, 2, -1, 2, 0) == -22            # a world of two needs the id
    assert lib.mifsk_gather_create(C.byref(h), ident, 0, 2, -1, 2, _lib.GATHER_LOOPBACK) == -22
    assert lib.mifsk_gather_create(C.byref(h), None, 0, 1, 99, 2, 0) == -19            # no such device
    g = M.NativeGatherer(None, 0, 1, loopback=True)
    b, nb = _batch(torch, 2, 8, 16)
    tk = C.c_uint64()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.mifsk_gather_start(g.handle, C.c_void_p(b.data_ptr()), 16, C.c_void_p(nb.data_ptr()), 8, 17, None,
                                  st, C.byref(tk)) == -22                              # more columns than the rows have
    assert lib.mifsk_gather_start(g.handle, None, 16, C.c_void_p(nb.data_ptr()), 8, 16, None, st, C.byref(tk)) == -22
    rows = (C.c_int * 1)(9)
    assert lib.mifsk_gather_start(g.handle, C.c_void_p(b.data_ptr()), 16, C.c_void_p(nb.data_ptr()), 8, 16, rows,
                                  st, C.byref(tk)) == -22                              # rows[rank] != nstreams
    g.close()
