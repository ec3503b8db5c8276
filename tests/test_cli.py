"""CPU half of the batch command-line program (minimodem_amd/csrc/mifsk_cli.c): its --tx --file
side (host transmitter + the Baudot / ASCII databits encoders) writes the very samples the
reference program writes, for every transmit command line of the reference's tests.  The
receive side needs the MI355X: tests/test_gpu_cli.py."""
import os
import subprocess

import numpy as np
import pytest

import _golden as G
import _oracle as O

CLI = os.path.join(O.REF_DIR, "minimodem_mifsk_batch")
pytestmark = pytest.mark.skipif(not (os.path.exists(CLI) and O.have_ref()),
                                reason="needs oracle/_ref (the reference built from its sources)")
REFDATA = np.load(os.path.join(G.GOLDEN_DIR, "refdata.npz"))

TX_LINES = [
    ("testdata_ascii_txt", "1200"), ("testdata_ascii_txt", "300"), ("testdata_baudot_txt", "rtty"),
    ("testdata_ascii_txt", "12000"), ("testdata_ascii_txt", "--float-samples 12000"),
    ("testdata_ascii_txt", "1200 --lut=0"), ("testdata_ascii_txt", "1200 --lut=16 --float-samples"),
    ("testdata_ascii_txt", "1200 --samplerate 24000 -M 1200 -S 2400 --lut=0 --float-samples"),
    ("testdata_ascii_txt", "--volume 3.50 1200"), ("testdata_ascii_txt", "--volume E 1200 --float-samples"),
    ("testdata_ascii_txt", "308"), ("testdata_multibyte_txt", "1200"), ("testdata_ascii_txt", "SAME"),
    ("testdata_ascii_txt", "-7 1200"), ("testdata_baudot_txt", "tdd"),
    ("testdata_callerid_mdmf_bytes", "1200 --ascii"), ("testdata_ascii_txt", "1200 -M 1200 -S 2400 --volume 0.5"),
]


@pytest.mark.parametrize("textfile,args", TX_LINES, ids=[a for _, a in TX_LINES])
def test_tx_writes_the_reference_samples(textfile, args, tmp_path):
    text = REFDATA[textfile].tobytes()
    mine, ref = str(tmp_path / "mine.wav"), str(tmp_path / "ref.wav")
    r = subprocess.run([CLI, "--tx", "--file", mine] + args.split(), input=text, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    O.ref_tx(text, args.split(), ref)
    sr1, x1 = O.read_wav(mine)
    sr2, x2 = O.read_wav(ref)
    assert sr1 == sr2 and len(x1) == len(x2) > 0
    assert np.array_equal(x1.view(np.uint32), x2.view(np.uint32))


def test_rx_without_a_gpu_fails_loudly(tmp_path):
    """no CPU receive path: without an MI355X the program says so and exits non-zero"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    wav = str(tmp_path / "t.wav")
    subprocess.run([CLI, "--tx", "--file", wav, "1200"], input=b"hello\n", check=True)
    r = subprocess.run([CLI, "--rx", "--file", wav, "1200"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and r.stdout == b""
