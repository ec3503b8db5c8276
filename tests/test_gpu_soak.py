"""A short run of the randomised parity soak (tools/soak.py) inside the GPU suite:
13 configurations x 24 seeded streams with noise, DC offset, clipping, rate slop,
truncation and double bursts; frames, episodes and bytes bit-identical to the
oracle.  (Round 1: seeds 1-5 with up to 256 streams per configuration, 6.5e5
frames, no mismatch.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [[], ["--ring"], ["--engine", "workgroup"]],
                         ids=["wave-flat", "wave-ring", "workgroup-flat"])
def test_soak_seed_7(variant):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--seed", "7",
                        "--streams", "24"] + variant, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    assert "0 mismatching streams" in out
