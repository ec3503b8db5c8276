"""The randomised parity soak (tools/soak.py) inside the GPU suite: 16 configurations x 48
seeded streams with noise, DC offset, clipping, rate slop, truncation and double bursts; frames,
episodes and bytes bit-identical to the oracle -- in one launch (both engines, RING
addressing), chained (random cuts, both engines), fed in slabs (both engines, RING slabs), and
without episode records.  Four seeds per variant here; tools/gpu/soak.sh runs more."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


VARIANTS = {
    "wave-flat": [], "wave-ring": ["--ring"], "workgroup-flat": ["--engine", "workgroup"],
    "wave-chained": ["--chain"], "workgroup-chained": ["--chain", "--engine", "workgroup"],
    "wave-slabs": ["--slabs", "4"], "workgroup-slabs": ["--slabs", "4", "--engine", "workgroup"],
    "ring-slabs": ["--slabs", "3", "--ring"], "no-episodes": ["--no-episodes"],
}


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7, 8, 9, 10])
@pytest.mark.parametrize("variant", sorted(VARIANTS))
def test_soak(variant, seed):
    """16 configurations x 48 seeded streams per (variant, seed), the oracle over every stream
    on all host cores; a hard time limit around the child (a hung kernel must not hang the suite)."""
    r = subprocess.run(["timeout", "-s", "KILL", "280", sys.executable, os.path.join(ROOT, "tools", "soak.py"),
                        "--seed", str(seed), "--streams", "48"] + VARIANTS[variant],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-2000:]
    assert "0 mismatching streams" in out
