#!/usr/bin/env python3
"""tests/golden/refdata.npz = the reference's own test INPUT files
(/root/reference/tests/testdata-*: the texts its self-tests transmit and expect back), so that
tests/test_gpu_cli.py can run those self-tests on the GPU box, where /root/reference is absent.
    python tests/golden/make_refdata.py"""
import glob
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for p in sorted(glob.glob("/root/reference/tests/testdata-*")):
    with open(p, "rb") as f:
        out[os.path.basename(p).replace("-", "_").replace(".", "_")] = np.frombuffer(f.read(), np.uint8)
np.savez_compressed(os.path.join(HERE, "refdata.npz"), **out)
print({k: len(v) for k, v in out.items()})
