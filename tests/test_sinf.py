"""minimodem_amd/csrc/mifsk_sinf.h -- the restatement of glibc's sinf that the device
transmitter uses for --lut=0 -- against the running C library, bit for bit (tools/sinf_check.c
compiled from the same header the device compiles).  The CPU suite sweeps every 509th
non-negative finite float plus the tone generator's whole working range densely; stride 1 (all
2.1e9 values, ~15 s) is `tools/sinf_check.c 1`."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sinf") / "sinf_check")
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-I", os.path.join(ROOT, "minimodem_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tools", "sinf_check.c"), "-lm"], check=True)
    return exe


def test_restated_sinf_equals_libm_on_a_strided_sweep(checker):
    """The restatement is pinned to THIS host's libm: glibc >= 2.28 on x86-64 with FMA (the
    library then runs its FMA build, and the header fuses where it does).  On a host without
    FMA, or with another C library, the comparison says nothing about the device code --
    skip rather than fail."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if " fma" not in flags:
        pytest.skip("host CPU without FMA: libm's sinf is another build than the one restated")
    r = subprocess.run([checker, "509"], stdout=subprocess.PIPE, timeout=300)
    if r.returncode != 0 and b"glibc" not in subprocess.run(["ldd", "--version"], stdout=subprocess.PIPE,
                                                              stderr=subprocess.STDOUT).stdout.lower():
        pytest.skip("not glibc: libm's sinf is not the algorithm restated in mifsk_sinf.h")
    assert r.returncode == 0, r.stdout.decode()
    assert b" 0 differ" in r.stdout
