"""The reference's self-tests (tests/*.test + tests/self-test) run through the BATCH binding:
oracle/_ref/minimodem_mifsk_batch = minimodem's `--tx/--rx --file` command line over
mifsk_demod_batch_host + mifsk_stream_text (minimodem_amd/csrc/mifsk_cli.c) -- transmit with the
host transmitter, receive with the whole file handed to the MI355X as a batch of one.

/root/reference does not exist on the GPU box, so the driver script is restated here
(`self_test` below = tests/self-test: transmit the text file, receive it, `cmp` the text, read
the statistics line, for -P require "confidence=inf ... (rate perfect)") and the reference's
test input files travel as tests/golden/refdata.npz (tests/golden/make_refdata.py).  One case
per reference test, command lines as in the .test files."""
import os
import re
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle as O

pytestmark = pytest.mark.gpu

CLI = os.path.join(O.REF_DIR, "minimodem_mifsk_batch")
REFDATA = np.load(os.path.join(G.GOLDEN_DIR, "refdata.npz"))


def data(name):
    return REFDATA[name].tobytes()


def run_cli(args, stdin=None):
    r = subprocess.run([CLI] + args, input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, (args, r.stderr.decode("latin-1"))
    return r.stdout, r.stderr.decode("latin-1")


def self_test(tmp_path, text, tx_args, rx_args=None, perfect=False, expect=None):
    """tests/self-test: tx, rx, cmp, statistics; returns the statistics string."""
    rx_args = tx_args if rx_args is None else rx_args
    wav = str(tmp_path / "t.wav")
    run_cli(["--tx", "--file", wav] + tx_args, stdin=text)
    out, err = run_cli(["--rx", "--file", wav] + rx_args)
    assert out == (text if expect is None else expect)			# cmp "$textfile" $TMPF.out
    lines = err.split("\n")
    assert lines[0].startswith("### CARRIER ") and lines[1] == ""	# read xlitcarrier; read xlitblankline
    m = re.match(r"### NOCARRIER (.*) ###$", lines[2])
    assert m, err
    if perfect:
        assert re.search(r"confidence=inf .* \(rate perfect\)", err), err
    return m.group(1)


PERFECT = "1200 --samplerate 24000 -M 1200 -S 2400".split()

# (reference test, text file, tx args, rx args or None, -P)
CASES = [
    ("01-self-test-1200", "testdata_ascii_txt", ["1200"], None, False),
    ("02-self-test-300", "testdata_ascii_txt", ["300"], None, False),
    ("03-self-test-rtty", "testdata_baudot_txt", ["rtty"], None, False),
    ("05-self-test-12000", "testdata_ascii_txt", ["12000"], None, False),
    ("06-self-test-float-samples", "testdata_ascii_txt", ["--float-samples", "12000"], None, False),
    ("07-self-test-no-lut", "testdata_ascii_txt", ["1200", "--lut=0"], None, False),
    ("08-self-test-lut16", "testdata_ascii_txt", ["1200", "--lut=16"], None, False),
    ("09-self-test-lut16-float", "testdata_ascii_txt", ["1200", "--lut=16", "--float-samples"], None, False),
    ("10-verify-perfect", "testdata_ascii_txt", PERFECT, None, True),
    ("11-verify-perfect-nolut", "testdata_ascii_txt", PERFECT + ["--lut=0"], None, True),
    ("12-verify-perfect-lut16", "testdata_ascii_txt", PERFECT + ["--lut=16"], None, True),
    ("13-verify-perfect-nolut-float", "testdata_ascii_txt", PERFECT + ["--lut=0", "--float-samples"], None, True),
    ("14-verify-perfect-lut16-float", "testdata_ascii_txt", PERFECT + ["--lut=16", "--float-samples"], None, True),
    ("15-verify-perfect-float", "testdata_ascii_txt", PERFECT + ["--float-samples"], None, True),
    ("60-multibyte", "testdata_multibyte_txt", ["1200"], None, False),
    ("80-SAME", "testdata_ascii_txt", ["SAME"], None, False),
    ("81-ascii7", "testdata_ascii_txt", ["-7", "1200"], None, False),
    ("81-tdd", "testdata_baudot_txt", ["tdd"], None, False),
]


@pytest.fixture(scope="module", autouse=True)
def _have_cli():
    if not os.path.exists(CLI):
        pytest.skip("oracle/_ref/minimodem_mifsk_batch not built (make -C oracle)")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_reference_self_test_through_the_batch_binding(case, tmp_path):
    name, textfile, tx, rx, perfect = case
    self_test(tmp_path, data(textfile), tx, rx, perfect)


def test_04_self_test_0p5(tmp_path):
    """tests/04-self-test-0.5.test: "KAMAL" at 0.5 baud (6.1 M samples, 96000-sample windows)"""
    self_test(tmp_path, b"KAMAL\n", ["0.5"])


@pytest.mark.parametrize("fmt", [[], ["--float-samples"]], ids=["16-int", "17-float"])
def test_16_17_tx_consistent(fmt, tmp_path):
    """tests/16-verify-tx-consistent.test, 17-...-float: three transmissions, identical files"""
    files = []
    for i in range(3):
        wav = str(tmp_path / ("t%d.wav" % i))
        run_cli(["--tx", "--file", wav, "1200"] + fmt, stdin=data("testdata_ascii_txt"))
        with open(wav, "rb") as f:
            files.append(f.read())
    assert files[0] == files[1] == files[2] and len(files[0]) > 44


@pytest.mark.parametrize("tx_rate", [292, 299, 300, 301, 308])
def test_21_rate_slop(tx_rate, tmp_path):
    """tests/21-rate-slop.test: transmit at 300 +- 8 baud, receive at 300"""
    self_test(tmp_path, data("testdata_ascii_txt"), [str(tx_rate)], ["300"])


@pytest.mark.parametrize("fmt", [[], ["--float-samples"]], ids=["30-int", "31-float"])
@pytest.mark.parametrize("ampl", ["3.50", "1.00", "0.30", "0.01", "E"])
def test_30_31_amplitude(ampl, fmt, tmp_path):
    """tests/30-amplitude.test, 31-amplitude-float.test: the received amplitude follows the
    transmit volume to 0.01 (integer files clamp at ~1.0; `E` = FLT_EPSILON reads as 0.0)"""
    flags = ["1200"] + fmt
    stats = self_test(tmp_path, data("testdata_ascii_txt"), ["--volume", ampl] + flags, flags)
    rx_ampl = float(re.search(r"ampl=([0-9.]+)", stats).group(1))
    a = 0.0 if ampl == "E" else float(ampl)
    clamped = a > 1.0 and 1.00 < rx_ampl < 1.02
    assert clamped or a - 0.01 < rx_ampl < a + 0.01, (ampl, rx_ampl)


@pytest.mark.parametrize("flags", [["1200"], ["1200", "-M", "1200", "-S", "2400"]], ids=["40", "41-purefreqs"])
@pytest.mark.parametrize("noise", ["0.00", "0.05", "0.10", "0.50"])
def test_40_41_noise(noise, flags, tmp_path):
    """tests/40-noise.test, 41-noise-purefreqs.test"""
    self_test(tmp_path, data("testdata_ascii_txt"), flags + ["--volume", "0.5"],
              flags + ["--Xrxnoise", noise, "--rx-one"])


@pytest.mark.parametrize("which", ["mdmf", "sdmf"], ids=["70-callerid-mdmf", "71-callerid-sdmf"])
def test_70_71_callerid(which, tmp_path):
    """tests/70-callerid-mdmf.test, 71-callerid-sdmf.test: raw message bytes sent as 1200 --ascii,
    received with the caller-ID decoder"""
    self_test(tmp_path, data("testdata_callerid_%s_bytes" % which), ["1200", "--ascii"], ["callerid"],
              expect=data("testdata_callerid_%s_txt" % which))


def test_ring_exact_option(tmp_path):
    """the same program with the reference's stale-cell buffer semantics (--ring-exact)"""
    wav = str(tmp_path / "t.wav")
    run_cli(["--tx", "--file", wav, "rtty"], stdin=data("testdata_baudot_txt"))
    a = run_cli(["--rx", "--file", wav, "rtty"])
    b = run_cli(["--rx", "--file", wav, "--ring-exact", "rtty"])
    assert a[0] == b[0] == data("testdata_baudot_txt")
