#!/usr/bin/env python3
"""Generate tests/golden/databits_vectors.npz from the REFERENCE's own databits
decoders (oracle/_ref/libdatabits_ref.so = unmodified src/databits_*.c,
baudot.c, uic_codes.c; built by `make -C oracle` where /root/reference exists):

    python tests/golden/make_databits_vectors.py

For each decoder a deterministic call sequence (bits, n_databits, reset flag)
and what the reference returned for every call.  tests/test_databits.py replays
the sequences through mifsk_databits_* (include/mifsk.h)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libdatabits_ref.so")
REFTESTS = "/root/reference/tests"

DECODERS = ["ascii8", "baudot", "binary", "callerid", "uic_ground", "uic_train"]


def run(lib, name, calls):
    fn = getattr(lib, "databits_decode_" + name)
    fn.restype = C.c_uint
    fn.argtypes = [C.c_char_p, C.c_uint, C.c_ulonglong, C.c_uint]
    buf = C.create_string_buffer(8192)
    out, lens = bytearray(), []
    for bits, n, reset in calls:
        if reset:
            fn(None, 0, 0, 0)
        k = fn(buf, 4096, bits, n)
        lens.append(k)
        out += buf.raw[:k]
    return out, lens


def cid_message(kind, body):
    msg = bytes([kind, len(body)]) + body
    return msg + bytes([(-sum(msg)) & 0xFF])


def sequences():
    rng = np.random.default_rng(20260924)
    seq = {}
    seq["ascii8"] = [(int(v), 8, i % 17 == 0) for i, v in
                     enumerate(rng.integers(0, 2 ** 63, size=300, dtype=np.uint64))]
    seq["binary"] = [(int(v), int(n), False) for v, n in
                     zip(rng.integers(0, 2 ** 63, size=120, dtype=np.uint64),
                         rng.choice([1, 5, 7, 8, 39, 47, 64], size=120))]
    # baudot: a few calls before the first reset (the reference starts in the
    # "unknown" shift state), then random codes with resets sprinkled in
    b = [(int(v), 5, False) for v in rng.integers(0, 32, size=40)]
    b += [(int(v) | 0xE0, 5, i % 97 == 0) for i, v in enumerate(rng.integers(0, 32, size=1500))]
    seq["baudot"] = b
    # caller-ID: the reference's own two test messages, crafted MDMF/SDMF
    # messages covering every parameter type, garbage in between, back to back
    # (the message buffer keeps its bytes across messages)
    stream = bytearray()
    for f in ("testdata-callerid-mdmf.bytes", "testdata-callerid-sdmf.bytes"):
        with open(os.path.join(REFTESTS, f), "rb") as fh:
            stream += fh.read()
    stream += b"\x55\xaa\x00"
    params = [(1, b"09241337"), (2, b"8005551212"), (2, b"5551212"), (7, b"MI355X DEMOD"),
              (4, b"O"), (4, b"P"), (8, b"O"), (8, b"P"), (8, b"X"), (4, b"OP"),
              (0, b"zz"), (3, b""), (5, b"q"), (6, b"abc"), (7, b"with\x00nul"), (1, b"0102"),
              (2, b"123456789\x00")]
    for k in range(0, len(params), 3):
        body = b"".join(bytes([t, len(d)]) + d for t, d in params[k:k + 3])
        stream += cid_message(0x80, body)
    stream += cid_message(0x80, bytes([9, 1, 65]))          # bad parameter type
    stream += cid_message(0x80, bytes([7, 250]) + b"AB")    # parameter longer than the buffer
    stream += cid_message(0x04, b"09241337" + b"8005551212")
    stream += cid_message(0x04, b"09241337" + b"5551212")
    stream += cid_message(0x04, b"01020304")
    for _ in range(40):                                     # random short messages
        kind = int(rng.choice([0x80, 0x04]))
        n = int(rng.integers(8, 60))
        stream += cid_message(kind, bytes(int(v) for v in rng.integers(0, 128, size=n)))
        stream += bytes(int(v) for v in rng.integers(0, 256, size=int(rng.integers(0, 4))))
    seq["callerid"] = [(v, 8, i in (0, 700)) for i, v in enumerate(stream)]
    codes = [0x00, 0x02, 0x03, 0x04, 0x06, 0x08, 0x09, 0x0A, 0x0C, 0x55, 0x7F, 0xFF]
    u = []
    for c in codes:
        rev = int("{:08b}".format(c)[::-1], 2)
        u.append(((int(rng.integers(0, 2 ** 24)) | (rev << 24) | (int(rng.integers(0, 128)) << 32)), 39, False))
    u += [(int(v), 39, False) for v in rng.integers(0, 2 ** 39, size=80, dtype=np.uint64)]
    seq["uic_ground"] = u
    seq["uic_train"] = list(u)
    return seq


def main():
    lib = C.CDLL(LIB)
    data = {}
    for name, calls in sequences().items():
        out, lens = run(lib, name, calls)
        data[name + "_bits"] = np.array([c[0] for c in calls], dtype=np.uint64)
        data[name + "_n"] = np.array([c[1] for c in calls], dtype=np.uint32)
        data[name + "_reset"] = np.array([c[2] for c in calls], dtype=np.uint8)
        data[name + "_len"] = np.array(lens, dtype=np.uint32)
        data[name + "_out"] = np.frombuffer(bytes(out), dtype=np.uint8)
        print("%-12s %5d calls -> %6d bytes" % (name, len(calls), len(out)))
    path = os.path.join(HERE, "databits_vectors.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
