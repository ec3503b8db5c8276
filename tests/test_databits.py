"""Host post-pass: the databits decoders of include/mifsk.h against
vectors produced by the reference's own decoders (tests/golden/
make_databits_vectors.py), and the CARRIER / NOCARRIER / stdout text of whole
streams against what the reference printed for the golden recordings.  CPU only:
the frame bits come from the oracle here; tests/test_gpu_parity.py feeds the same
post-pass from the device's output."""
import os

import numpy as np
import pytest

import _golden as G
import _oracle as O
import minimodem_amd as M

VEC = np.load(os.path.join(G.GOLDEN_DIR, "databits_vectors.npz"))
NAMES = {"ascii8": "ascii8", "baudot": "baudot", "binary": "binary", "callerid": "callerid",
         "uic_ground": "uic-ground", "uic_train": "uic-train"}


@pytest.mark.parametrize("name", sorted(NAMES))
def test_decoder_matches_reference_vectors(name):
    d = M.DataBits(NAMES[name])
    bits, n, reset = VEC[name + "_bits"], VEC[name + "_n"], VEC[name + "_reset"]
    lens, ref = VEC[name + "_len"], VEC[name + "_out"].tobytes()
    pos = 0
    for i in range(bits.shape[0]):
        if reset[i]:
            d.reset()
        got = d.decode(int(bits[i]), int(n[i]))
        want = ref[pos:pos + int(lens[i])]
        assert got == want, (name, i, hex(int(bits[i])), got, want)
        pos += int(lens[i])
    assert pos == len(ref)


def test_decode_respects_output_size_and_null_reset():
    import ctypes as C
    lib = M._lib.load()
    h = C.c_void_p()
    assert lib.mifsk_databits_create(C.byref(h), M.DECODERS["binary"]) == 0
    buf = C.create_string_buffer(8)
    assert lib.mifsk_databits_decode(h, buf, 4, 0xFF, 8) == 4 and buf.raw[:4] == b"1111"
    assert lib.mifsk_databits_decode(h, None, 0, 0, 0) == 0          # reset convention
    lib.mifsk_databits_destroy(h)
    assert lib.mifsk_databits_create(C.byref(h), 17) != 0            # unknown decoder


def _print_filter(g):
    return "-p" in g["rx_args"] or "--print-filter" in g["rx_args"]


@pytest.mark.parametrize("name", G.names())
def test_stream_text_matches_reference_stdout_and_stderr(name):
    """oracle frames + episodes -> post-pass == the reference's stdout and its
    '### CARRIER' / '### NOCARRIER' lines, for every golden recording (ascii,
    baudot, caller-ID, binary output, print filter)."""
    g = G.load(name)
    cfg = M.rx_config(**g["cfg_kwargs"])
    r = O.oracle_rx_stream(O.oracle_config(**g["cfg_kwargs"]), g["samples"])
    out, err = M.stream_text(cfg, r["frames"]["bits"], r["episodes"], print_filter=_print_filter(g),
                             b_mark=r["carrier_band"])
    assert out == g["stdout"]
    lines = [l for l in err.splitlines() if l]
    assert [l for l in lines if l.startswith("### CARRIER")] == g["carrier"]
    assert [l for l in lines if l.startswith("### NOCARRIER")] == g["nocarrier"]
    # each NOCARRIER line is preceded by an empty line, as the reference prints it
    assert err.count("\n### NOCARRIER") == len(g["nocarrier"])
    q_out, q_err = M.stream_text(cfg, r["frames"]["bits"], r["episodes"], quiet=True,
                                 print_filter=_print_filter(g), b_mark=r["carrier_band"])
    assert q_out == out and q_err == ""


def test_stream_text_capacity_and_arguments():
    import ctypes as C
    lib = M._lib.load()
    cfg = M.rx_config("1200")
    bits = np.array([0x41, 0x42, 0x43], dtype=np.uint64)
    eps = np.zeros(1, dtype=M.EPISODE_DTYPE)
    eps["nframes"], eps["carrier_nsamples"] = 3, 1200
    eps["confidence_total"], eps["amplitude_total"], eps["end_reason"] = 12.0, 3.0, 2
    eps["b_mark"] = cfg.b_mark           # the band at acquisition travels with the episode
    out, err = M.stream_text(cfg, bits, eps)
    assert out == b"ABC" and err.startswith("### CARRIER 1200 @ 1200.0 Hz ###\n\n### NOCARRIER ndata=3 ")
    small = C.create_string_buffer(2)
    nout, nerr = C.c_size_t(), C.c_size_t()
    rc = lib.mifsk_stream_text(C.byref(cfg), bits.ctypes.data, 3, eps.ctypes.data, 1, 0,
                               small, 2, C.byref(nout), None, 0, C.byref(nerr))
    assert rc < 0 and nout.value == 3 and small.raw == b"AB" and nerr.value == len(err)
    assert lib.mifsk_stream_text(None, None, 0, None, 0, 0, None, 0, None, None, 0, None) < 0
