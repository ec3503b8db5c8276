"""Drop-in proof on the GPU: the reference's OWN main() (src/minimodem.c and
friends, compiled unmodified into oracle/_ref/minimodem_mifsk with src/fsk.c
left out) linked against libmifsk.so, receiving the reference's own kind of
test input.  Every fsk_plan_new / fsk_find_frame / fsk_plan_destroy call that
main() makes lands in the HIP path through include/fsk.h's C ABI.

Expected values come from the goldens, i.e. from the reference program itself
(tests/golden/make_golden.py): decoded stdout byte for byte and the
"### NOCARRIER ndata=.. confidence=.. ampl=.. bps=.. (rate ..)" statistics line
character for character (that line is what the reference's -P tests grep)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle as O

pytestmark = pytest.mark.gpu

DROPIN = os.path.join(O.REF_DIR, "minimodem_mifsk")


def _write_wav(path, g):
    raw = g["stored"]           # the file as the reference wrote it (before any --Xrxnoise)
    if raw.dtype == np.int16:
        data, tag, bits = raw.astype("<i2").tobytes(), 1, 16
    else:
        data, tag, bits = raw.astype("<f4").tobytes(), 3, 32
    sr = g["sample_rate"]
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt ")
        f.write(struct.pack("<IHHIIHH", 16, tag, 1, sr, sr * bits // 8, bits // 8, bits))
        f.write(b"data" + struct.pack("<I", len(data)))
        f.write(data)


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/minimodem_mifsk not built")
@pytest.mark.parametrize("name", G.names())
def test_reference_main_over_libmifsk(name, tmp_path):
    g = G.load(name)
    z = np.load(os.path.join(G.GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    rx_args = [str(a) for a in z["rx_args"].tolist()]
    wav = str(tmp_path / "in.wav")
    _write_wav(wav, g)
    r = subprocess.run([DROPIN, "--rx", "--file", wav] + rx_args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == g["stdout"]
    lines = [l for l in r.stderr.decode().splitlines() if l.startswith("### NOCARRIER")]
    assert lines == g["nocarrier"]


RXBATCH = os.path.join(O.REF_DIR, "minimodem_mifsk_rxbatch")


@pytest.mark.skipif(not os.path.exists(RXBATCH), reason="oracle/_ref/minimodem_mifsk_rxbatch not built")
@pytest.mark.parametrize("name", G.names())
def test_reference_main_with_the_rx_batch_patch(name, tmp_path):
    """The reference's own main() with integration/minimodem-rx-batch.patch applied (oracle/Makefile
    applies it to a scratch copy at build time): `--rx --file` leaves main() before its receive
    loop and goes through libmifsk's batch entry -- whole file to the device, the loop on the
    device, frame bits through the databits post-pass.  Same command lines, same expectations
    as the unpatched drop-in above: the reference's stdout byte for byte, its NOCARRIER
    statistics line character for character, on every golden."""
    g = G.load(name)
    z = np.load(os.path.join(G.GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    rx_args = [str(a) for a in z["rx_args"].tolist()]
    wav = str(tmp_path / "in.wav")
    _write_wav(wav, g)
    r = subprocess.run([RXBATCH, "--rx", "--file", wav] + rx_args,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stdout == g["stdout"]
    lines = [l for l in r.stderr.decode().splitlines() if l.startswith("### NOCARRIER")]
    assert lines == g["nocarrier"]
