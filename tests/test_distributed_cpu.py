"""The N>1 path on CPU: world_size-2 `gloo` processes exercise the stream
sharding and the gather of decoded bytes to rank 0 (minimodem_amd.shard_range /
gather_bytes, and ByteGatherer -- the class bench.py runs over RCCL on GPUs).  The demod
kernel itself needs a GPU and is covered by the -m gpu tests; here each rank's
"decoded bytes" come from the oracle so that the assembled result can be
checked against a single-process decode of the whole batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as O
import minimodem_amd as M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(nstreams):
    cfg = M.rx_config("1200")
    rng = np.random.default_rng(99)
    streams = []
    for i in range(nstreams):
        words = rng.integers(32, 127, size=20 + (i % 5), dtype=np.uint8)
        streams.append((M.synthesize(cfg, words, leading_silence=i % 7), bytes(words)))
    return streams


def _worker(rank, world, port, nstreams, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        streams = _make_batch(nstreams)
        lo, hi = M.shard_range(nstreams, rank, world)
        ocfg = O.oracle_config("1200")
        cap = 32
        local = torch.zeros((hi - lo, cap), dtype=torch.uint8)
        counts = torch.zeros(hi - lo, dtype=torch.int32)
        for j, i in enumerate(range(lo, hi)):
            b = O.oracle_rx_stream(ocfg, streams[i][0])["bytes"]
            local[j, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
            counts[j] = len(b)
        bufs, cnts = M.gather_bytes(local, counts, dst=0)
        if rank == 0:
            out = []
            for bb, cc in zip(bufs, cnts):
                for row, n in zip(bb.numpy(), cc.numpy()):
                    out.append(row[:n].tobytes())
            q.put(out)
        else:
            assert bufs is None and cnts is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nstreams", [7, 8])
def test_two_rank_shard_and_gather(nstreams):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nstreams, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [w for _, w in _make_batch(nstreams)]
    assert out == expected


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [M.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _pipelined_worker(rank, world, port, q):
    """bench.py's N>1 step structure: double-buffered outputs, the gather of step i
    overlapped with step i+1, waited on before its buffers are reused."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = M.ByteGatherer(dist, rank, world)
        bufs = [(torch.zeros((3, 8), dtype=torch.uint8), torch.zeros(3, dtype=torch.int32))
                for _ in range(2)]
        pending = [None, None]
        seen = []
        for step in range(5):
            b = step & 1
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None
            bufs[b][0].fill_(10 * rank + step)       # "decode" step into buffer b
            bufs[b][1].fill_(step + 1)
            pending[b] = g.start(*bufs[b])
            if rank == 0:                             # (the root checks what the previous step brought)
                for w in pending[b]:
                    w.wait()
                pending[b] = None
                seen.append([(int(g.received(r)[0][0, 0]), int(g.received(r)[1][0]))
                             for r in range(1, world)])
        for b in (0, 1):
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
        dist.barrier()
        if rank == 0:
            q.put(seen)
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_as_in_bench():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipelined_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert seen == [[(10 + step, step + 1)] for step in range(5)]


def _narrow_worker(rank, world, port, q):
    """ByteGatherer with `cols` (only the columns that can hold data are shipped, from a
    staging copy) and unequal shards (`rows`); then bench.py's agree(): one rank's failure
    becomes RankFailed on EVERY rank instead of a hang in the next collective."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        rows = [4, 3]
        g = M.ByteGatherer(dist, rank, world, cols=5, rows=rows)
        seen = []
        for step in range(3):
            b = torch.arange(rows[rank] * 16, dtype=torch.int32).reshape(rows[rank], 16).to(torch.uint8) + step
            n = torch.full((rows[rank],), step + 2, dtype=torch.int32)
            for w in g.start(b, n):
                w.wait()
            if rank == 0:
                rb, rn = g.received(1)
                seen.append((tuple(rb.shape), rb[2].tolist(), rn.tolist()))
        assert g.bytes_per_peer(rows[rank]) == rows[rank] * 5 + 4 * rows[rank]
        verdict = []
        for err in (None, RuntimeError("boom") if rank == 1 else None):
            try:
                bench.agree(torch, dist, err, "phase")
                verdict.append("ok")
            except bench.RankFailed as e:
                verdict.append("failed:" + ("mine" if "boom" in str(e) else "peer"))
        dist.barrier()
        q.put((rank, seen, verdict))
    finally:
        dist.destroy_process_group()


def test_narrow_gather_unequal_shards_and_failure_agreement():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_narrow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (seen, verdict)) for r, seen, verdict in (q.get(timeout=120), q.get(timeout=120)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen0, v0 = got[0]
    _, v1 = got[1]
    # rank 1's row 2 is 32 .. 47 (+ step); five columns arrive
    assert seen0 == [((3, 5), [32 + s, 33 + s, 34 + s, 35 + s, 36 + s], [s + 2] * 3) for s in range(3)]
    assert v0 == ["ok", "failed:peer"] and v1 == ["ok", "failed:mine"]


def test_bench_spawns_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` without torchrun: bench.py must start the two ranks itself
    (torch.distributed.run on 127.0.0.1) instead of asserting on WORLD_SIZE.  MIFSK_BENCH_DRYRUN
    runs the launch path without a GPU: rendezvous (gloo), sharding, reduce, one JSON line."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env["MIFSK_BENCH_DRYRUN"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 2 and d["total_streams"] == 2048


def _world8_worker(rank, world, port, q, depth=2):
    """bench.py's timed loop at the size of the node it is written for: EIGHT ranks, unequal
    shards (1027 streams: 129 129 129 128 ...), the narrow gather (`cols`), `depth` output sets
    with the gather of step i overlapped with the steps after it (bench.py --pipeline: passes in
    flight), the root fanning in from seven peers at once -- and one rank that fails in the
    middle of the loop, which must end in RankFailed on every rank (agree()) after every
    gather has still been joined."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        total, cap, cols, steps = 1027, 24, 9, 6
        spans = [M.shard_range(total, r, world) for r in range(world)]
        rows = [b - a for a, b in spans]
        lo, hi = spans[rank]
        g = M.ByteGatherer(dist, rank, world, cols=cols, rows=rows, slots=depth)
        bufs = [(torch.zeros((rows[rank], cap), dtype=torch.uint8), torch.zeros(rows[rank], dtype=torch.int32))
                for _ in range(depth)]
        pending = [None] * depth
        failure = None
        checked = 0

        def decode(step, b):
            # "stream gid decoded at step `step`": byte j of stream gid = (gid + 3 j + step) mod 251
            gid = torch.arange(lo, hi, dtype=torch.int64).reshape(-1, 1)
            j = torch.arange(cap, dtype=torch.int64).reshape(1, -1)
            bufs[b][0].copy_(((gid + 3 * j + step) % 251).to(torch.uint8))
            bufs[b][1].copy_(((torch.arange(lo, hi) + step) % (cols + 1)).to(torch.int32))

        def check(step):
            # what the root holds of every peer after step `step`'s gather has been waited on
            n = 0
            for r in range(1, world):
                rb, rn = g.received(r, step)
                plo, phi = spans[r]
                assert tuple(rb.shape) == (phi - plo, cols) and tuple(rn.shape) == (phi - plo,)
                gid = torch.arange(plo, phi, dtype=torch.int64).reshape(-1, 1)
                j = torch.arange(cols, dtype=torch.int64).reshape(1, -1)
                # (rank 5's launch of step 3 failed: it still joined the gather, with what its
                # buffer held -- the bytes of the step that last wrote that set)
                es = step - depth if (r == 5 and step == 3) else step
                assert torch.equal(rb, ((gid + 3 * j + es) % 251).to(torch.uint8)), (r, step)
                assert torch.equal(rn, ((torch.arange(plo, phi) + es) % (cols + 1)).to(torch.int32)), (r, step)
                n += phi - plo
            return n

        for step in range(steps):
            b = step % depth
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None
                if rank == 0:
                    checked += check(step - depth)
            try:
                if rank == 5 and step == 3:
                    raise RuntimeError("launch failed on rank 5")
                decode(step, b)
            except Exception as e:				# noqa: BLE001 -- this rank still joins every gather
                failure = failure or e
            pending[b] = g.start(*bufs[b])
        for b in [(steps + k) % depth for k in range(depth)]:	# oldest first
            if pending[b] is not None:
                for w in pending[b]:
                    w.wait()
                pending[b] = None
        dist.barrier()
        verdict = "ok"
        try:
            bench.agree(torch, dist, failure, "timed loop")
        except bench.RankFailed as e:
            verdict = "failed:" + ("mine" if "rank 5" in str(e) else "peer")
        q.put((rank, rows, checked, verdict, g.bytes_per_peer(rows[rank])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth", [2, 3])
def test_eight_ranks_pipelined_narrow_gather_unequal_shards_one_failing_rank(depth):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 8
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q, depth)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, rows, checked, verdict, bpp = q.get(timeout=300)
        got[rank] = (rows, checked, verdict, bpp)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = got[0][0]
    assert rows == [129, 129, 129, 128, 128, 128, 128, 128] and sum(rows) == 1027
    # the root verified every peer's rows of the gathers completed inside the loop (six steps:
    # steps 0..3 with two output sets, 0..2 with three)
    assert got[0][1] == (6 - depth) * (1027 - 129)
    for r in range(world):
        assert got[r][2] == ("failed:mine" if r == 5 else "failed:peer"), (r, got[r][2])
        assert got[r][3] == rows[r] * 9 + 4 * rows[r]


def test_bench_dry_run_eight_ranks_strong_scaling_configs3():
    """`bench.py --gpus 8 --scaling strong --config 12000` through its own launcher, without a
    GPU: the job is BASELINE configs[3]'s 65536 streams whatever the rank count, the shards are
    contiguous and cover it exactly."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env["MIFSK_BENCH_DRYRUN"] = "1"
    env["OMP_NUM_THREADS"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--scaling", "strong",
                        "--config", "12000"], env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 8 and d["total_streams"] == 65536 and d["scaling"] == "strong"
    assert d["shards"] == [[8192 * i, 8192 * (i + 1)] for i in range(8)]
