"""what each build of the library would launch for the bench workloads (mifsk_demod_plan_ex)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import minimodem_amd as M
from minimodem_amd import _lib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for tag in sys.argv[1:]:
    _lib._lib = None
    _lib.LIB_PATH = os.path.join(ROOT, "minimodem_amd", "libmifsk.so" if tag == "main" else "libmifsk_%s.so" % tag)
    _lib.load()
    ctx = M.Context(0)
    for name in ("1200", "12000", "same", "rtty"):
        entry, mode, per_gpu, seconds, _, amp = bench.WORKLOADS[name]
        cfg = M.rx_config(mode)
        nsamp = bench.NSAMPLES if name == "1200" else int(seconds * cfg.sample_rate)
        print(tag, name, json.dumps(M.demod_plan(ctx, cfg, per_gpu, nsamples=(nsamp + 3) & ~3)))
