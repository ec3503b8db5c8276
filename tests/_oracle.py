"""ctypes bindings to the CHECKERS under oracle/ (test infrastructure only).

  * liboracle_fsk.so      -- the C restatement (oracle/fsk_oracle.c)
  * _ref/libfsk_ref.so    -- the reference's own src/fsk.c + FFT shim
  * _ref/minimodem_ref    -- the whole reference program + shims

Nothing in the product package imports this module.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
MINIMODEM_REF = os.path.join(REF_DIR, "minimodem_ref")

MAX_BITS = 64


class ModemArgs(C.Structure):
    _fields_ = [
        ("baudmode", C.c_char_p),
        ("sample_rate", C.c_uint),
        ("mark_f", C.c_float),
        ("space_f", C.c_float),
        ("band_width", C.c_float),
        ("n_data_bits", C.c_int),
        ("baudot", C.c_int),
        ("nstartbits", C.c_int),
        ("nstopbits", C.c_float),
        ("invert_start_stop", C.c_int),
        ("inverted_freqs", C.c_int),
        ("msb_first", C.c_int),
        ("have_sync_byte", C.c_int),
        ("sync_byte", C.c_longlong),
        ("confidence_threshold", C.c_float),
        ("search_limit", C.c_float),
        ("binary_output", C.c_int),
        ("binary_raw_nbits", C.c_int),
        ("rx_one", C.c_int),
        ("auto_carrier_threshold", C.c_float),
    ]


class RxConfig(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_uint),
        ("data_rate", C.c_float),
        ("mark_f", C.c_float),
        ("space_f", C.c_float),
        ("band_width", C.c_float),
        ("n_data_bits", C.c_uint),
        ("nstartbits", C.c_int),
        ("nstopbits", C.c_float),
        ("invert_start_stop", C.c_int),
        ("msb_first", C.c_int),
        ("do_rx_sync", C.c_int),
        ("sync_byte", C.c_ulonglong),
        ("decoder", C.c_int),
        ("rx_one", C.c_int),
        ("confidence_threshold", C.c_float),
        ("search_limit", C.c_float),
        ("auto_carrier_threshold", C.c_float),
        ("autodetect_shift", C.c_int),
        ("inverted_freqs", C.c_int),
        ("fftsize", C.c_int),
        ("nbands", C.c_uint),
        ("b_mark", C.c_uint),
        ("b_space", C.c_uint),
        ("frame_n_bits", C.c_uint),
        ("nsamples_per_bit", C.c_float),
        ("nsamples_overscan", C.c_uint),
        ("frame_nsamples", C.c_uint),
        ("expect_n_bits", C.c_uint),
        ("expect_nsamples", C.c_uint),
        ("expect_data", C.c_char * (MAX_BITS + 4)),
        ("expect_sync", C.c_char * (MAX_BITS + 4)),
        ("samplebuf_size", C.c_uint),
        ("try_first", C.c_uint * 2),
        ("try_max", C.c_uint * 2),
        ("try_step", C.c_uint * 2),
        ("try_step_fine", C.c_uint * 2),
        ("find_samples_per_bit", C.c_float),
        ("bit_nsamples", C.c_uint),
        ("bit_offset", C.c_uint * MAX_BITS),
    ]

    def as_dict(self):
        out = {}
        for name, _t in self._fields_:
            v = getattr(self, name)
            if isinstance(v, bytes):
                v = v.decode()
            elif hasattr(v, "__len__"):
                v = list(v)
            out[name] = v
        return out


class Frame(C.Structure):
    _fields_ = [
        ("bits", C.c_uint64),
        ("start", C.c_uint64),
        ("confidence", C.c_float),
        ("amplitude", C.c_float),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class Episode(C.Structure):
    _fields_ = [
        ("carrier_nsamples", C.c_uint64),
        ("first_frame", C.c_uint32),
        ("nframes", C.c_uint32),
        ("confidence_total", C.c_float),
        ("amplitude_total", C.c_float),
        ("end_reason", C.c_uint32),
        ("b_mark", C.c_uint32),
    ]


FRAME_DTYPE = np.dtype([("bits", "<u8"), ("start", "<u8"), ("confidence", "<f4"),
                        ("amplitude", "<f4"), ("flags", "<u4"), ("reserved", "<u4")])
EPISODE_DTYPE = np.dtype([("carrier_nsamples", "<u8"), ("first_frame", "<u4"),
                          ("nframes", "<u4"), ("confidence_total", "<f4"),
                          ("amplitude_total", "<f4"), ("end_reason", "<u4"),
                          ("b_mark", "<u4")])
assert FRAME_DTYPE.itemsize == C.sizeof(Frame) == 32
assert EPISODE_DTYPE.itemsize == C.sizeof(Episode) == 32


class RxResult(C.Structure):
    _fields_ = [
        ("frames", C.c_void_p), ("frames_cap", C.c_size_t), ("nframes", C.c_size_t),
        ("episodes", C.c_void_p), ("episodes_cap", C.c_size_t), ("nepisodes", C.c_size_t),
        ("bytes", C.c_void_p), ("bytes_cap", C.c_size_t), ("nbytes", C.c_size_t),
        ("carrier_band", C.c_int), ("carrier_b_space", C.c_uint),
        ("n_scan_windows", C.c_ulonglong),
        ("n_iterations", C.c_ulonglong),
        ("n_find_frame", C.c_ulonglong),
        ("n_positions", C.c_ulonglong),
    ]


class OPlan(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_float), ("f_mark", C.c_float), ("f_space", C.c_float),
        ("filter_bw", C.c_float), ("fftsize", C.c_int), ("nbands", C.c_uint),
        ("band_width", C.c_float), ("b_mark", C.c_uint), ("b_space", C.c_uint),
        ("tw_bit_nsamples", C.c_uint), ("tw_b_mark", C.c_uint), ("tw_b_space", C.c_uint),
        ("tw", C.c_void_p),
    ]


class RefPlan(C.Structure):
    """struct fsk_plan of the reference (src/fsk.h:30-46) as laid out on x86-64."""
    _fields_ = [
        ("sample_rate", C.c_float), ("f_mark", C.c_float), ("f_space", C.c_float),
        ("filter_bw", C.c_float), ("fftsize", C.c_int), ("nbands", C.c_uint),
        ("band_width", C.c_float), ("b_mark", C.c_uint), ("b_space", C.c_uint),
        ("fftplan", C.c_void_p), ("fftin", C.c_void_p), ("fftout", C.c_void_p),
    ]


_FIND_ARGS = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float,
              C.c_char_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_float),
              C.POINTER(C.c_uint)]


def build_oracle():
    """(Re)build the checkers.  The _ref targets are built only where
    /root/reference exists (this container); elsewhere the prebuilt files are used."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_oracle_lib = None


def oracle_lib():
    global _oracle_lib
    if _oracle_lib is None:
        path = os.path.join(ORACLE_DIR, "liboracle_fsk.so")
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        lib.ofsk_plan_new.restype = C.POINTER(OPlan)
        lib.ofsk_plan_new.argtypes = [C.c_float] * 4
        lib.ofsk_plan_destroy.argtypes = [C.c_void_p]
        lib.ofsk_find_frame.restype = C.c_float
        lib.ofsk_find_frame.argtypes = _FIND_ARGS
        lib.ofsk_frame_analyze.restype = C.c_float
        lib.ofsk_frame_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int,
                                           C.c_char_p, C.POINTER(C.c_ulonglong),
                                           C.POINTER(C.c_float)]
        lib.ofsk_bit_analyze.argtypes = [C.c_void_p, C.c_void_p, C.c_uint,
                                         C.POINTER(C.c_uint), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float)]
        lib.ofsk_bit_dft.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        lib.ofsk_bit_dft_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        lib.ofsk_detect_carrier.restype = C.c_int
        lib.ofsk_detect_carrier.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_float]
        lib.ofsk_set_tones_by_bandshift.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        lib.ofsk_last_n_positions.restype = C.c_uint
        lib.ofsk_twiddle.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
        lib.ofsk_modem_args_default.argtypes = [C.POINTER(ModemArgs)]
        lib.ofsk_rx_config_init.restype = C.c_int
        lib.ofsk_rx_config_init.argtypes = [C.POINTER(RxConfig), C.POINTER(ModemArgs)]
        lib.ofsk_rx_stream.restype = C.c_int
        lib.ofsk_rx_stream.argtypes = [C.POINTER(RxConfig), C.c_void_p, C.c_size_t, C.c_int,
                                       C.POINTER(RxResult)]
        _oracle_lib = lib
    return _oracle_lib


_ref_lib = None


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "libfsk_ref.so")) and os.path.exists(MINIMODEM_REF)


def ref_lib():
    """The reference's own fsk.c (unmodified) + FFT shim."""
    global _ref_lib
    if _ref_lib is None:
        lib = C.CDLL(os.path.join(REF_DIR, "libfsk_ref.so"))
        lib.fsk_plan_new.restype = C.POINTER(RefPlan)
        lib.fsk_plan_new.argtypes = [C.c_float] * 4
        lib.fsk_plan_destroy.argtypes = [C.c_void_p]
        lib.fsk_find_frame.restype = C.c_float
        lib.fsk_find_frame.argtypes = _FIND_ARGS
        lib.fsk_detect_carrier.restype = C.c_int
        lib.fsk_detect_carrier.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_float]
        lib.fsk_set_tones_by_bandshift.argtypes = [C.c_void_p, C.c_uint, C.c_int]
        lib.oracle_fft_execute_count.restype = C.c_ulonglong
        _ref_lib = lib
    return _ref_lib


def make_args(baudmode="1200", **kw):
    lib = oracle_lib()
    a = ModemArgs()
    lib.ofsk_modem_args_default(C.byref(a))
    a.baudmode = str(baudmode).encode()
    if "sync_byte" in kw:
        a.have_sync_byte = 1
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def oracle_config(baudmode="1200", **kw):
    lib = oracle_lib()
    a = make_args(baudmode, **kw)
    cfg = RxConfig()
    rc = lib.ofsk_rx_config_init(C.byref(cfg), C.byref(a))
    if rc != 0:
        raise ValueError("ofsk_rx_config_init -> %d" % rc)
    cfg._args = a  # keep baudmode bytes alive
    return cfg


def _find(lib_fn, plan, samples, frame_nsamples, first, tmax, step, limit, expect):
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    bits = C.c_ulonglong(0)
    ampl = C.c_float(0)
    start = C.c_uint(0)
    conf = lib_fn(plan, samples.ctypes.data, frame_nsamples, first, tmax, step,
                  C.c_float(limit), expect.encode() if isinstance(expect, str) else expect,
                  C.byref(bits), C.byref(ampl), C.byref(start))
    return float(np.float32(conf)), int(bits.value), float(np.float32(ampl.value)), int(start.value)


def oracle_find_frame(plan, samples, frame_nsamples, first, tmax, step, limit, expect):
    return _find(oracle_lib().ofsk_find_frame, plan, samples, frame_nsamples, first, tmax,
                 step, limit, expect)


def ref_find_frame(plan, samples, frame_nsamples, first, tmax, step, limit, expect):
    return _find(ref_lib().fsk_find_frame, plan, samples, frame_nsamples, first, tmax,
                 step, limit, expect)


def oracle_rx_stream(cfg, samples, ring_mode=False):
    """Run the restated receive loop over one stream.  Returns a dict of numpy arrays."""
    lib = oracle_lib()
    samples = np.ascontiguousarray(samples, dtype=np.float32)
    n = samples.shape[0]
    min_adv = max(1, int(cfg.frame_nsamples) - int(cfg.nsamples_overscan))
    cap = n // min_adv + 4
    frames = np.zeros(cap, dtype=FRAME_DTYPE)
    episodes = np.zeros(max(8, cap // 4), dtype=EPISODE_DTYPE)
    out_bytes = np.zeros(cap, dtype=np.uint8)
    res = RxResult()
    res.frames = frames.ctypes.data
    res.frames_cap = cap
    res.episodes = episodes.ctypes.data
    res.episodes_cap = episodes.shape[0]
    res.bytes = out_bytes.ctypes.data
    res.bytes_cap = cap
    rc = lib.ofsk_rx_stream(C.byref(cfg), samples.ctypes.data, n, 1 if ring_mode else 0,
                            C.byref(res))
    if rc != 0:
        raise RuntimeError("ofsk_rx_stream -> %d" % rc)
    assert res.nframes <= cap and res.nepisodes <= episodes.shape[0]
    return {
        "frames": frames[: res.nframes].copy(),
        "episodes": episodes[: res.nepisodes].copy(),
        "bytes": out_bytes[: res.nbytes].tobytes(),
        "carrier_band": int(res.carrier_band),
        "carrier_b_space": int(res.carrier_b_space),
        "n_scan_windows": int(res.n_scan_windows),
        "n_iterations": int(res.n_iterations),
        "n_find_frame": int(res.n_find_frame),
        "n_positions": int(res.n_positions),
    }


def oracle_batch_mismatches(cfg, samples, lens, res, ring_mode=False, chunk_bytes=1 << 30,
                            threads=None, what=("frames", "episodes", "bytes"), groups=None):
    """Whole-batch parity: the restated receive loop over EVERY stream of a batch, on all host
    cores (ofsk_rx_stream is re-entrant and ctypes releases the GIL), compared with a GPU
    result `res` (numpy arrays from results_to_host: frames / nframes, episodes / nepisodes,
    bytes / nbytes -- whichever of `what` it holds).  `samples` is a [nstreams, stride] numpy
    array or torch tensor (device tensors are brought over in chunks of `chunk_bytes`); `lens`
    the stream lengths (None: the row width).  Returns (indices of mismatching streams,
    {group: frames equal / streams} if `groups` -- a function of the stream index -- is given,
    seconds spent in the oracle)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    oracle_lib()
    nstreams, stride = int(samples.shape[0]), int(samples.shape[1])
    if lens is None:
        lens = np.full(nstreams, stride, np.int64)
    elif not isinstance(lens, np.ndarray):
        lens = lens.cpu().numpy()
    lens = lens.astype(np.int64)
    threads = threads or (os.cpu_count() or 1)
    per = max(1, int(chunk_bytes // max(1, stride * 4)))
    bad, by_group, t_oracle = [], {}, 0.0

    def one(args):
        i, xi = args
        ref = oracle_rx_stream(cfg, xi, ring_mode=ring_mode)
        ok = True
        if "frames" in what and "frames" in res:
            nf = int(res["nframes"][i])
            ok = ok and nf == len(ref["frames"]) and \
                res["frames"][i, :min(nf, res["frames"].shape[1])].tobytes() == ref["frames"].tobytes()
        elif "nframes" in res:
            ok = ok and int(res["nframes"][i]) == len(ref["frames"])
        if "episodes" in what and "episodes" in res:
            ne = int(res["nepisodes"][i])
            m = min(ne, res["episodes"].shape[1])
            ok = ok and ne == len(ref["episodes"]) and \
                res["episodes"][i, :m].tobytes() == ref["episodes"][:m].tobytes()
        if "bytes" in what and "bytes" in res:
            ok = ok and res["bytes"][i, :int(res["nbytes"][i])].tobytes() == ref["bytes"]
        return i, ok

    with ThreadPoolExecutor(max_workers=threads) as ex:
        for c0 in range(0, nstreams, per):
            c1 = min(nstreams, c0 + per)
            host = samples[c0:c1]
            if not isinstance(host, np.ndarray):
                host = host.cpu().numpy()
            t0 = time.perf_counter()
            for i, ok in ex.map(one, [(c0 + j, host[j, :int(lens[c0 + j])]) for j in range(c1 - c0)]):
                if not ok:
                    bad.append(i)
                if groups is not None:
                    g = by_group.setdefault(groups(i), [0, 0])
                    g[0] += bool(ok)
                    g[1] += 1
            t_oracle += time.perf_counter() - t0
            del host
    return bad, by_group, t_oracle


# ---------------------------------------------------------------------------
# WAV helpers + the reference binary
# ---------------------------------------------------------------------------

def read_wav(path):
    """Return (sample_rate, float32 mono samples) with libsndfile's S16 normalisation."""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        ln = int.from_bytes(data[pos + 4:pos + 8], "little")
        body = data[pos + 8:pos + 8 + ln]
        if cid == b"fmt ":
            tag = int.from_bytes(body[0:2], "little")
            ch = int.from_bytes(body[2:4], "little")
            sr = int.from_bytes(body[4:8], "little")
            bits = int.from_bytes(body[14:16], "little")
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            tag, ch, sr, bits = fmt
            assert ch == 1
            if tag == 1 and bits == 16:
                x = np.frombuffer(body[: len(body) // 2 * 2], dtype="<i2").astype(np.float32)
                x = x / np.float32(32768.0)
            elif tag == 3 and bits == 32:
                x = np.frombuffer(body[: len(body) // 4 * 4], dtype="<f4").copy()
            else:
                raise ValueError("unsupported wav encoding")
            return sr, x
        pos += 8 + ln + (ln & 1)
    raise ValueError("no data chunk")


def write_wav(path, x, sample_rate, s16):
    """mono WAV: PCM16 from an int16 array, or IEEE float32"""
    import struct
    if s16:
        data, tag, bits = np.asarray(x, "<i2").tobytes(), 1, 16
    else:
        data, tag, bits = np.asarray(x, "<f4").tobytes(), 3, 32
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt ")
        f.write(struct.pack("<IHHIIHH", 16, tag, 1, sample_rate, sample_rate * bits // 8,
                            bits // 8, bits))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def ref_tx(payload, tx_args, path):
    """minimodem_ref --tx --file path <tx_args>  < payload"""
    subprocess.run([MINIMODEM_REF, "--tx", "--file", path] + list(tx_args),
                   input=payload, check=True)


def ref_rx(path, rx_args):
    """minimodem_ref --rx --file path <rx_args> -> (stdout bytes, stderr text)"""
    r = subprocess.run([MINIMODEM_REF, "--rx", "--file", path] + list(rx_args),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    return r.stdout, r.stderr.decode()


def tmp_wav():
    fd, path = tempfile.mkstemp(suffix=".wav", prefix="mifsk-")
    os.close(fd)
    return path


def format_nocarrier(cfg, ep):
    """The reference's "### NOCARRIER ..." line for one episode
    (src/minimodem.c:253-291), from the episode totals."""
    f32 = np.float32
    nframes = int(ep["nframes"])
    frame_n_bits = f32(cfg.frame_n_bits)
    nbits = f32(nframes) * frame_n_bits
    sr = f32(cfg.sample_rate)
    cns = int(ep["carrier_nsamples"])
    with np.errstate(all="ignore"):
        rate = nbits * sr / f32(cns)
        conf = f32(ep["confidence_total"]) / f32(nframes)
        ampl = f32(ep["amplitude_total"]) / f32(nframes)
        s = "### NOCARRIER ndata=%u confidence=%.3f ampl=%.3f bps=%.2f" % (
            nframes, float(conf), float(ampl), float(rate))
        lhs = int(np.uint64(f32(nbits * sr + f32(0.5))))
        rhs = int(np.uint64(f32(f32(cfg.data_rate) * f32(cns))))
        if lhs == rhs:
            s += " (rate perfect) ###"
        else:
            skew = (rate - f32(cfg.data_rate)) / f32(cfg.data_rate)
            s += " (%.1f%% %s) ###" % (float(abs(skew) * f32(100.0)),
                                         "slow" if np.signbit(skew) else "fast")
    return s
