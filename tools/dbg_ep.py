import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import test_gpu_fullsize as T
import _oracle as O
import minimodem_amd as M, torch
ctx = M.Context(0)
cfg = M.rx_config("1200")
host, payloads = T._batch(M, cfg, 16, 1199, seed=42, nsamples=480000)
res = T._run(M, torch, ctx, cfg, host, want=("bytes","episodes","frames"))
ocfg = O.oracle_config("1200")
for i in range(6):
    ref = O.oracle_rx_stream(ocfg, host[i])
    e = res["episodes"][i,0]; r = ref["episodes"][0]
    fr = res["frames"][i,:1199]
    same = fr.tobytes()==ref["frames"].tobytes()
    c = fr["confidence"].astype(np.float32)
    seq = np.float32(0)
    for v in c: seq = np.float32(seq+v)
    print(i, "frames same", same, "ct gpu %r ref %r seqsum %r | at gpu %r ref %r" % (e["confidence_total"], r["confidence_total"], seq, e["amplitude_total"], r["amplitude_total"]))
    fl = fr["flags"]; print("   flagged frames:", np.nonzero(fl)[0][:12])
