"""First contact with RCCL before the driver's 8-GPU run: the calls bench.py makes for N > 1
(process group on the `nccl` backend with a device id, all_reduce(MAX) of the timing, barrier,
and ByteGatherer's grouped isend/irecv of device tensors) executed on the one GPU a test box
has.  world_size is 1, so the peer-to-peer ops are a rank's send to ITSELF posted together with
the matching receive in one batch_isend_irecv group."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["MIFSK_ROOT"])
import minimodem_amd as M
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # bench.py: max step time over ranks
assert float(t.item()) == 1.25
dist.barrier()
# the gather's transport: grouped send/recv of (bytes, nbytes) device tensors
b = torch.arange(4 * 64, dtype=torch.int32, device="cuda").to(torch.uint8).reshape(4, 64)
n = torch.tensor([64, 3, 0, 17], dtype=torch.int32, device="cuda")
rb, rn = torch.empty_like(b), torch.empty_like(n)
ops = [dist.P2POp(dist.irecv, rb, 0), dist.P2POp(dist.irecv, rn, 0),
       dist.P2POp(dist.isend, b, 0), dist.P2POp(dist.isend, n, 0)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
assert torch.equal(rb, b) and torch.equal(rn, n)
# and the world-size-1 behaviour of the classes bench.py uses
g = M.ByteGatherer(dist, 0, 1)
assert g.start(b, n) == []
assert M.shard_range(1024, 0, 1) == (0, 1024)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_rccl_process_group_and_grouped_send_recv_on_device_tensors():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MIFSK_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"RCCL_OK" in r.stdout, r.stderr.decode()[-3000:]


def test_bench_line_for_every_config_small():
    """bench.py end to end on small batches of every BASELINE entry (one JSON line, payloads
    round-trip, roofline and CPU legs present)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cfg, streams in (("1200", 64), ("12000", 128), ("same", 64), ("rtty", 32)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--streams",
                            str(streams), "--steps", "2", "--warmup", "1"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        line = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["frac"] > 0
        ok, judged = (int(v) for v in line["payload_roundtrip_ok_streams"].split("/"))
        assert judged > 0 and ok >= judged * 0.9, line["payload_roundtrip_ok_streams"]
        assert line["cpu_port"]["mismatching_streams"] == 0
        assert line["cpu_baseline"]["mismatching_streams"] == 0


def test_bench_step_structure_of_n_gt_1_on_one_gpu_with_the_gather_sent_to_itself():
    """What the driver's scaling run executes at N > 1 -- process group on the `nccl` backend with a
    device id, the library's pipeline, the gather of every pass queued on its lane's stream with
    several in flight, agree(), the per-rank reductions -- at world size 1 with the gather forced
    on as a send of every pass's bytes to this rank itself (bench.py --gather-self).  The line
    must carry per_rank, RCCL's own rank count, and what came back must be what was sent."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29519", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # (the last one with the gather behind the C ABI: mifsk_gather_*, bench.py --native-gather)
    for cfg, streams, extra in (("1200", 512, []), ("12000", 1024, []), ("1200", 512, ["--native-gather"])):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--streams", str(streams),
                            "--steps", "8", "--warmup", "2", "--gather-self", "--no-cpu", "--no-h2d",
                            "--preheat-ms", "50"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           timeout=600, cwd=root, env=env)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        line = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == 1 and line["gather_self_ok"] is True
        assert line["gather_through"].startswith("mifsk_gather_" if extra else "torch.distributed")
        assert line["ranks"] == {"world_size": 1, "backend": "nccl"}
        pr = line["per_rank"]
        assert len(pr["kernel_ms_avg"]) == 1 and pr["kernel_ms_avg"][0] > 0
        assert pr["gather_bytes_per_peer_per_step"] > 0
        assert line["pipeline"]["passes_in_flight"] >= 1
        sets = line["pipeline"]["output_sets_equal_to_serial_launch"].split("/")
        assert sets[0] == sets[1]
        ok, judged = (int(v) for v in line["payload_roundtrip_ok_streams"].split("/"))
        assert ok == judged > 0
        assert line["value_serial"] > 0 and line["cold_ms_per_step"] > 0
