"""Loader for tests/golden/*.npz (made by tests/golden/make_golden.py from the
reference itself)."""
import glob
import os

import numpy as np

import _oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "t[0-9]*.npz")))


_INT_KEYS = {"sample_rate", "n_data_bits", "nstartbits", "binary_output", "inverted_freqs", "rx_one"}
_FLOAT_KEYS = {"mark_f", "space_f", "band_width", "nstopbits", "auto_carrier_threshold"}


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    kw = {}
    for k, v in zip(z["cfg_keys"].tolist(), z["cfg_vals"].tolist()):
        if k in _INT_KEYS:
            kw[k] = int(v)
        elif k in _FLOAT_KEYS:
            kw[k] = float(v)
        else:
            kw[k] = v
    s = z["samples"]
    if "samples_sha256" in z.files:
        # a recording too long to store (0.5 baud: 6.1 M samples): regenerated with the host
        # transmitter, which tests/test_tx_synth.py pins to the reference's WAV files, and checked
        # against the hash of what the reference wrote
        import hashlib
        import minimodem_amd as M
        is_s16 = str(z["samples_dtype"]) == "<i2"
        xr = M.synthesize(M.rx_config(**kw), z["payload"].tobytes(), s16=is_s16)
        s = np.rint(xr * 32768.0).astype("<i2") if is_s16 else xr.astype("<f4")
        assert s.shape[0] == int(z["samples_len"])
        assert hashlib.sha256(s.tobytes()).hexdigest() == str(z["samples_sha256"]), name
    if s.dtype == np.int16:
        x = s.astype(np.float32) / np.float32(32768.0)   # libsndfile S16 -> float
    else:
        x = s.astype(np.float32)
    rx_args = [str(a) for a in z["rx_args"].tolist()]
    rxnoise = float(rx_args[rx_args.index("--Xrxnoise") + 1]) if "--Xrxnoise" in rx_args else 0.0
    if rxnoise:
        # simpleaudio-sndfile.c:64-69 with rand()/RAND_MAX == 0 (integer division)
        x = x + (np.float32(0) - np.float32(0.5)) * (np.float32(rxnoise) * np.float32(2))
    return {
        "name": name,
        "stored": s,            # the samples in their on-disk encoding (int16 or float32)
        "rxnoise": rxnoise,
        "samples": x,
        "sample_rate": int(z["sample_rate"]),
        "payload": z["payload"].tobytes(),
        "stdout": z["stdout"].tobytes(),
        "nocarrier": [str(s) for s in z["nocarrier"].tolist()],
        "carrier": [str(s) for s in z["carrier"].tolist()],
        "rx_args": [str(s) for s in z["rx_args"].tolist()],
        "trace": z["trace"],
        "cfg_kwargs": kw,
    }


def raw_stdout(g):
    """The bytes the ascii8 decoder produced, before `--print-filter` (the golden
    stdout is what was printed; with the filter on, the unfiltered bytes are the
    transmitted payload)."""
    if "--print-filter" in g["rx_args"] or "-p" in g["rx_args"]:
        return g["payload"]
    return g["stdout"]
