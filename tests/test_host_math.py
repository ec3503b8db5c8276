"""The identity minimodem_amd/csrc/mifsk_devmath.h band_mag() relies on, against the running C
library: hypotf(x, y) -- what the reference's band_mag calls, /root/reference/src/fsk.c:107-114 --
equals (float)sqrt((double)x * x + (double)y * y) bit for bit, plus C's hypot(+-inf, NaN) = +inf
(tools/hypotf_check.c: 2 x 2^24 pseudo-random pairs, special values against a sweep; the pair
space is 2^62, `hypotf_check 29` compares 10^9 pairs in ~10 s)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hypotf") / "hypotf_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tools", "hypotf_check.c"), "-lm"], check=True)
    return exe


def test_libm_hypotf_is_the_rounded_double_sqrt_of_the_sum_of_squares(checker):
    r = subprocess.run([checker, "24"], stdout=subprocess.PIPE, timeout=300)
    if r.returncode != 0 and b"glibc" not in subprocess.run(["ldd", "--version"], stdout=subprocess.PIPE,
                                                              stderr=subprocess.STDOUT).stdout.lower():
        pytest.skip("not glibc: this host's hypotf is not the one the reference was pinned with")
    assert r.returncode == 0, r.stdout.decode()
    assert b" 0 differ" in r.stdout
