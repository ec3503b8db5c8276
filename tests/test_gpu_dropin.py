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


@pytest.mark.skipif(not os.path.exists(RXBATCH) or not O.have_ref(), reason="oracle/_ref not built")
def test_rx_batch_patch_leaves_files_it_does_not_read_to_the_reference_loop(tmp_path):
    """The batch entry reads mono PCM16 / float32 WAV; the reference opens whatever libsndfile
    opens (FLAC, AIFF, AU, 24-bit PCM ...: /root/reference/src/simpleaudio-sndfile.c:113-160).
    For anything else the patched main() must behave as if the patch were not there: the file goes
    to the reference's own receive loop -- here, with the WAV-only libsndfile shim under it, to the
    reference's own "cannot open" failure, word for word what the unpatched program says -- and
    not to an error message of the batch path's."""
    au = tmp_path / "in.au"
    au.write_bytes(b".snd" + struct.pack(">IIIII", 24, 8000, 3, 48000, 1) + bytes(8000))
    outs = []
    for exe in (O.MINIMODEM_REF, RXBATCH):
        r = subprocess.run([exe, "--rx", "--file", str(au), "1200"], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=120)
        outs.append((r.returncode, r.stdout, r.stderr))
    assert outs[0] == outs[1], outs
    assert b"not a mono PCM16" not in outs[1][2]
