"""minimodem_amd -- MI355X-native batched FSK demodulator behind minimodem's
fsk.h API.

The signal path lives in libmifsk.so (hand-written HIP for gfx950, C ABI in
include/fsk.h + include/mifsk.h).  This package is the thin host-side mirror
used by the tests, the benchmark and Python callers: it marshals arguments,
uses PyTorch only for device memory / streams / torch.distributed, and raises
if the native library is missing -- there is no fallback implementation.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib
from ._lib import ModemArgs, RxConfig  # noqa: F401

__all__ = ["build", "rx_config", "Context", "demod_batch", "find_frame_batch",
           "LegacyPlan", "synthesize", "FRAME_DTYPE", "EPISODE_DTYPE",
           "gather_bytes", "shard_range"]

_HERE = os.path.dirname(os.path.abspath(__file__))

FRAME_DTYPE = np.dtype([("bits", "<u8"), ("start", "<u8"), ("confidence", "<f4"),
                        ("amplitude", "<f4"), ("flags", "<u4"), ("reserved", "<u4")])
EPISODE_DTYPE = np.dtype([("carrier_nsamples", "<u8"), ("first_frame", "<u4"),
                          ("nframes", "<u4"), ("confidence_total", "<f4"),
                          ("amplitude_total", "<f4"), ("end_reason", "<u4"),
                          ("b_mark", "<u4")])
SEARCH_DTYPE = np.dtype([("sample_offset", "<u8"), ("navail", "<u4"), ("try_first", "<u4"),
                         ("try_max", "<u4"), ("try_step", "<u4"), ("search_limit", "<f4"),
                         ("use_sync_string", "<u4")])
RESULT_DTYPE = np.dtype([("bits", "<u8"), ("confidence", "<f4"), ("amplitude", "<f4"),
                         ("frame_start", "<u4"), ("n_positions", "<u4")])
assert SEARCH_DTYPE.itemsize == 32 and RESULT_DTYPE.itemsize == 24
NCOUNTERS = 32
COUNTER_NAMES = {0: "iterations", 1: "batches", 2: "stages", 3: "bulk_frames", 4: "refines",
                 5: "cache_hits", 6: "positions", 7: "lattice_batches", 8: "cyc_total",
                 9: "cyc_scan", 10: "cyc_wait", 11: "cyc_confidence", 12: "cyc_bulk", 13: "w_stage", 14: "w_correlate",
                 15: "w_barrier", 16: "cyc_general", 17: "cyc_restart", 18: "cyc_scan1", 19: "cyc_scan2", 20: "cyc_replay_scan", 21: "cyc_scan_wait",
                 24: "conf_fallbacks", 25: "seg_second_looks"}


def build(force=False):
    """Compile libmifsk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force and os.path.exists(_lib.LIB_PATH):
        os.unlink(_lib.LIB_PATH)
    subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc")], check=True)
    return _lib.LIB_PATH


def rx_config(baudmode="1200", **opts):
    """The --rx command line as a derived configuration
    (mifsk_rx_config_init; reference src/minimodem.c:819-965,1037-1131)."""
    lib = _lib.load()
    a = ModemArgs()
    lib.mifsk_modem_args_default(C.byref(a))
    a.baudmode = str(baudmode).encode()
    if "sync_byte" in opts:
        a.have_sync_byte = 1
    for k, v in opts.items():
        if not hasattr(a, k):
            raise TypeError("unknown modem option %r" % k)
        setattr(a, k, v)
    cfg = RxConfig()
    rc = lib.mifsk_rx_config_init(C.byref(cfg), C.byref(a))
    if rc != 0:
        raise ValueError("mifsk_rx_config_init failed: %d" % rc)
    cfg._keepalive = a
    return cfg


class Context:
    """A device context (twiddle tables etc.); one per process/GPU."""

    def __init__(self, device=-1):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.mifsk_ctx_create(C.byref(h), int(device))
        if rc != 0:
            raise RuntimeError("mifsk_ctx_create failed: %d (no gfx950 HIP device? "
                               "this package has no CPU path)" % rc)
        self.handle = h

    @property
    def device_name(self):
        return self._lib.mifsk_ctx_device_name(self.handle).decode()

    def selftest_sqrt(self, nvalues, seed=1):
        """mifsk_selftest_sqrt: the kernels' short square root against the exact sequence on
        `nvalues` sums of squares -> (accepted and differing [must be 0], sent to the exact
        sequence by the guard, differing without the guard, evaluated)."""
        counts = (C.c_uint64 * 4)()
        rc = self._lib.mifsk_selftest_sqrt(self.handle, int(seed), int(nvalues), counts)
        if rc != 0:
            raise RuntimeError("mifsk_selftest_sqrt failed: %d" % rc)
        return tuple(int(c) for c in counts)

    def close(self):
        if self.handle:
            self._lib.mifsk_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch():
    import torch
    return torch


def _on(torch, stream):
    """Allocations and fills that a launch on `stream` depends on are queued ON that stream
    (torch orders a fill on its current stream only: on another one it would race the kernel)."""
    return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()


def _stream_ptr(torch, stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _check_nsamples(torch, nsamples, nstreams):
    """Per-stream lengths cross the C ABI as a raw uint32 pointer: reject what the
    kernels would silently misread (a CPU tensor, int64 lengths, a wrong shape)."""
    if nsamples is None:
        return
    assert nsamples.is_cuda and nsamples.dtype in (torch.int32, torch.uint32), \
        "nsamples must be a CUDA int32/uint32 tensor"
    assert nsamples.dim() == 1 and nsamples.shape[0] == nstreams and nsamples.is_contiguous()


def max_frames(cfg, nsamples):
    return int(_lib.load().mifsk_max_frames(C.byref(cfg), int(nsamples)))


def demod_batch(ctx, cfg, samples, nsamples=None, want=("bytes", "episodes"),
                frames_cap=None, episodes_cap=8, stream=None, out=None, ring_exact=False,
                engine=None, force_engine=False):
    """Run the receive loop over a batch of streams resident in HBM.

    samples : torch.float32 CUDA tensor [nstreams, stride] (stride % 4 == 0)
    nsamples: optional torch.int32/uint32 CUDA tensor [nstreams]; default: stride
    Returns a dict of CUDA tensors (no synchronisation is performed).
    `out` may be a dict returned by a previous call with the same shapes, to
    reuse its buffers (nothing is allocated inside the timed region then).
    ring_exact: MIFSK_IO_RING_EXACT (the reference's stale-cell buffer semantics).
    engine: None (the library chooses), "wave" (one wavefront per stream,
    MIFSK_IO_ENGINE_WAVE) or "workgroup" (one workgroup of a master and 2-3 worker waves per stream,
    MIFSK_IO_ENGINE_WORKGROUP); force_engine=True makes None mean "wave".
    """
    torch = _torch()
    lib = _lib.load()
    assert samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 2
    assert samples.stride(1) == 1
    nstreams, width = samples.shape
    _check_nsamples(torch, nsamples, nstreams)     # (the kernels clamp lengths to the row width)
    # (a lone row's stride is arbitrary in torch -- numpy's x[None, :] has stride 0 -- so its
    # width stands in for it: the kernels read whole float4s, the row must hold them)
    assert nstreams > 1 or width % 4 == 0, "a lone row must be padded to a multiple of 4 samples"
    stride = samples.stride(0) if nstreams > 1 else int(width)
    n_uniform = int(width)
    if frames_cap is None:
        frames_cap = max_frames(cfg, n_uniform)
    dev = samples.device
    if out is None:
        # (allocated and zero-filled ON the launch stream: a fill queued on torch's current
        # stream would race a kernel launched on another one)
        with _on(torch, stream):
            out = {}
            out["nframes"] = torch.zeros(nstreams, dtype=torch.int32, device=dev)
            out["status"] = torch.zeros(nstreams, dtype=torch.int32, device=dev)
            if "bytes" in want:
                out["bytes"] = torch.zeros((nstreams, frames_cap), dtype=torch.uint8, device=dev)
                out["nbytes"] = torch.zeros(nstreams, dtype=torch.int32, device=dev)
            if "bits" in want:
                out["bits"] = torch.zeros((nstreams, frames_cap), dtype=torch.int64, device=dev)
            if "frames" in want:
                out["frames"] = torch.zeros((nstreams, frames_cap, FRAME_DTYPE.itemsize),
                                            dtype=torch.uint8, device=dev)
            if "counters" in want:
                out["counters"] = torch.zeros((nstreams, NCOUNTERS), dtype=torch.int64, device=dev)
            if "carrier_band" in want or cfg.auto_carrier_threshold > 0:
                out["carrier_band"] = torch.full((nstreams,), -1, dtype=torch.int32, device=dev)
            if "episodes" in want:
                out["episodes"] = torch.zeros((nstreams, episodes_cap, EPISODE_DTYPE.itemsize),
                                              dtype=torch.uint8, device=dev)
                out["nepisodes"] = torch.zeros(nstreams, dtype=torch.int32, device=dev)
        if stream is not None:
            # The tensors belong to `stream` in torch's caching allocator.  Whoever reads them on
            # another stream must order that read behind `stream` (an event, wait_stream); marking
            # them used on the current stream as well keeps the allocator from handing a dropped
            # tensor's block back out -- and its zero-fill onto `stream` -- under such a reader.
            cur = torch.cuda.current_stream()
            for t in out.values():
                t.record_stream(cur)

    def ptr(name):
        t = out.get(name)
        return C.c_void_p(t.data_ptr()) if t is not None else None

    io = _lib.DemodIO()
    io.d_samples = samples.data_ptr()
    io.stream_stride = stride
    io.d_nsamples = nsamples.data_ptr() if nsamples is not None else None
    io.nsamples = n_uniform
    io.nstreams = nstreams
    io.d_bytes = ptr("bytes")
    io.d_nbytes = ptr("nbytes")
    io.d_bits = ptr("bits")
    io.d_frames = ptr("frames")
    io.d_nframes = ptr("nframes")
    io.frames_cap = frames_cap
    io.d_episodes = ptr("episodes")
    io.d_nepisodes = ptr("nepisodes")
    io.episodes_cap = episodes_cap
    io.d_status = ptr("status")
    io.d_counters = ptr("counters")
    io.d_carrier_band = ptr("carrier_band")
    if engine is None and force_engine:
        engine = "wave"
    io.flags = (_lib.IO_RING_EXACT if ring_exact else 0) | \
        (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | \
        (_lib.IO_ENGINE_WAVE if engine == "wave" else 0)
    rc = lib.mifsk_demod_batch(ctx.handle, C.byref(cfg), C.byref(io), _stream_ptr(torch, stream))
    if rc != 0:
        raise RuntimeError("mifsk_demod_batch failed: %d" % rc)
    return out


def demod_plan(ctx, cfg, nstreams, ring_exact=False, engine=None, nsamples=None):
    """mifsk_demod_plan[_ex]: what demod_batch would launch (kernel instantiation, engine,
    workgroup size, dynamic LDS per workgroup, workgroups per CU by LDS, LATTICE mode, and --
    for rows of `nsamples` samples; None: long ones -- the groups x time chunks a large batch
    is cut into)."""
    info = _lib.LaunchInfo()
    flags = (_lib.IO_RING_EXACT if ring_exact else 0) | \
        (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | \
        (_lib.IO_ENGINE_WAVE if engine == "wave" else 0)
    rc = _lib.load().mifsk_demod_plan_ex(ctx.handle, C.byref(cfg), int(nstreams),
                                         0xFFFFFFFF if nsamples is None else int(nsamples), flags, C.byref(info))
    if rc != 0:
        raise RuntimeError("mifsk_demod_plan failed: %d" % rc)
    d = {k: getattr(info, k) for k, _ in info._fields_}
    d["kernel"] = d["kernel"].decode()
    d["engine"] = "workgroup" if d["engine"] == _lib.IO_ENGINE_WORKGROUP else "wave"
    return d


def results_to_host(out):
    """Copy a demod_batch() result to numpy (synchronises)."""
    res = {}
    for k, t in out.items():
        a = t.cpu().numpy()
        if k == "frames":
            a = a.view(FRAME_DTYPE).reshape(a.shape[0], a.shape[1])
        elif k == "episodes":
            a = a.view(EPISODE_DTYPE).reshape(a.shape[0], a.shape[1])
        elif k == "bits":
            a = a.view(np.uint64)
        res[k] = a
    return res


def find_frame_batch(ctx, cfg, samples, problems, stream=None):
    """N independent fsk_find_frame() problems over a flat CUDA float buffer.
    `problems` is a numpy array of SEARCH_DTYPE; returns numpy RESULT_DTYPE."""
    torch = _torch()
    lib = _lib.load()
    assert samples.is_cuda and samples.dtype == torch.float32
    problems = np.ascontiguousarray(problems, dtype=SEARCH_DTYPE)
    n = problems.shape[0]
    with _on(torch, stream):
        d_prob = torch.from_numpy(problems.view(np.uint8).reshape(-1).copy()).to(samples.device)
        d_res = torch.zeros(n * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=samples.device)
    rc = lib.mifsk_find_frame_batch(ctx.handle, C.byref(cfg), C.c_void_p(samples.data_ptr()),
                                    C.c_void_p(d_prob.data_ptr()), C.c_void_p(d_res.data_ptr()),
                                    n, _stream_ptr(torch, stream))
    if rc != 0:
        raise RuntimeError("mifsk_find_frame_batch failed: %d" % rc)
    return d_res.cpu().numpy().view(RESULT_DTYPE)


class LegacyPlan:
    """The reference's fsk.h API (include/fsk.h), called through the C ABI with
    host buffers exactly as src/minimodem.c calls it."""

    def __init__(self, sample_rate, f_mark, f_space, filter_bw):
        self._lib = _lib.load()
        self.p = self._lib.fsk_plan_new(sample_rate, f_mark, f_space, filter_bw)
        if not self.p:
            raise OSError(C.get_errno(), "fsk_plan_new failed")

    def __getattr__(self, name):
        return getattr(self.p.contents, name)

    def find_frame(self, samples, frame_nsamples, try_first, try_max, try_step, limit, expect):
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        bits, ampl, start = C.c_ulonglong(0), C.c_float(0), C.c_uint(0)
        conf = self._lib.fsk_find_frame(self.p, samples.ctypes.data, frame_nsamples, try_first,
                                        try_max, try_step, C.c_float(limit),
                                        expect.encode() if isinstance(expect, str) else expect,
                                        C.byref(bits), C.byref(ampl), C.byref(start))
        return float(conf), int(bits.value), float(ampl.value), int(start.value)

    def detect_carrier(self, samples, min_mag_threshold):
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        return int(self._lib.fsk_detect_carrier(self.p, samples.ctypes.data, samples.shape[0],
                                                C.c_float(min_mag_threshold)))

    def set_tones_by_bandshift(self, b_mark, b_shift):
        self._lib.fsk_set_tones_by_bandshift(self.p, b_mark, b_shift)

    def close(self):
        if self.p:
            self._lib.fsk_plan_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def synthesize(cfg, words, lut=4096, amplitude=1.0, leading_silence=0, s16=False):
    """FSK audio for `words` exactly as `minimodem --tx --file` writes it
    (host-side generator, csrc/mifsk_tx.cpp).  Returns float32 numpy."""
    lib = _lib.load()
    words = np.ascontiguousarray(np.frombuffer(bytes(words), dtype=np.uint8)
                                 if isinstance(words, (bytes, bytearray)) else words,
                                 dtype=np.uint8)
    n = lib.mifsk_tx_synthesize(C.byref(cfg), words.ctypes.data, words.shape[0], lut,
                                C.c_float(amplitude), leading_silence, 1 if s16 else 0,
                                None, 0)
    if n < 0:
        raise ValueError("mifsk_tx_synthesize failed: %d" % n)
    out = np.zeros(n, dtype=np.float32)
    lib.mifsk_tx_synthesize(C.byref(cfg), words.ctypes.data, words.shape[0], lut,
                            C.c_float(amplitude), leading_silence, 1 if s16 else 0,
                            out.ctypes.data, n)
    return out


# ---------------------------------------------------------------------------
# multi-GPU: streams are independent, so ranks own contiguous stream ranges and
# the only exchange is a gather of decoded bytes (RCCL over xGMI on GPUs; the
# same code runs over gloo on CPU tensors in the tests).
# ---------------------------------------------------------------------------

def shard_range(nstreams, rank, world):
    """Contiguous, balanced [lo, hi) stream range of `rank`."""
    q, r = divmod(nstreams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_bytes(local_bytes, local_nbytes, dst=0, group=None):
    """Gather every rank's decoded bytes ([n_local, cap] uint8 + [n_local] int32
    counts) to rank `dst`.  Returns (bytes, nbytes) lists on dst, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return [local_bytes], [local_nbytes]
    torch = _torch()
    shape = torch.tensor([local_bytes.shape[0], local_bytes.shape[1]], dtype=torch.int64,
                         device=local_bytes.device)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    if rank == dst:
        bufs = [torch.empty((int(s[0]), int(s[1])), dtype=torch.uint8, device=local_bytes.device)
                for s in shapes]
        cnts = [torch.empty(int(s[0]), dtype=torch.int32, device=local_bytes.device)
                for s in shapes]
        bufs[dst] = local_bytes
        cnts[dst] = local_nbytes
        reqs = []
        for r in range(world):
            if r == dst:
                continue
            reqs.append(dist.P2POp(dist.irecv, bufs[r], r, group))
            reqs.append(dist.P2POp(dist.irecv, cnts[r], r, group))
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
        return bufs, cnts
    reqs = [dist.P2POp(dist.isend, local_bytes.contiguous(), dst, group),
            dist.P2POp(dist.isend, local_nbytes.contiguous(), dst, group)]
    for w in dist.batch_isend_irecv(reqs):
        w.wait()
    return None, None


class _DevArray:
    """A device array the library owns, for torch.as_tensor (zero copy)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2, "strides": None}


class Pipeline:
    """mifsk_pipeline_*: `depth` batches in flight behind the C ABI -- lanes of a context, a HIP
    stream and a completion event each, and (outputs(...)) one set of output arrays per lane,
    all owned by the library.  submit() is mifsk_demod_batch as the pipeline's next pass and
    returns its ticket; pass t runs on lane t % depth.  Results are those of demod_batch.

        pipe = M.Pipeline(device, depth=3)
        pipe.outputs(nstreams, frames_cap, want=("bytes",))
        t = pipe.submit(cfg, samples)            # ... more submits: they overlap on the device
        pipe.wait(t); out = pipe.result(t)       # dict of torch views of that pass's set
    """

    def __init__(self, device=-1, depth=3):
        self._lib = _lib.load()
        h = C.c_void_p()
        rc = self._lib.mifsk_pipeline_create(C.byref(h), int(device), int(depth))
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_create failed: %d" % rc)
        self.handle = h
        info = _lib.PipelineInfo()
        self._lib.mifsk_pipeline_info_get(self.handle, C.byref(info))
        self.depth, self.depth_requested, self.hw_queues = int(info.depth), int(info.depth_requested), int(info.hw_queues)
        self._views = {}
        self._shape = None

    def info(self):
        info = _lib.PipelineInfo()
        self._lib.mifsk_pipeline_info_get(self.handle, C.byref(info))
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    def outputs(self, nstreams, frames_cap, episodes_cap=8, want=("bytes",)):
        flags = (_lib.WANT_BYTES if "bytes" in want else 0) | (_lib.WANT_BITS if "bits" in want else 0) | \
            (_lib.WANT_FRAMES if "frames" in want else 0) | (_lib.WANT_EPISODES if "episodes" in want else 0)
        rc = self._lib.mifsk_pipeline_outputs_alloc(self.handle, int(nstreams), int(frames_cap),
                                                    int(episodes_cap), flags)
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_outputs_alloc failed: %d" % rc)
        self._views = {}
        self._shape = (int(nstreams), int(frames_cap), int(episodes_cap))

    def result(self, ticket):
        """torch views (no copy, no synchronisation) of the output set pass `ticket` writes"""
        lane = int(ticket) % self.depth
        if lane in self._views:
            return self._views[lane]
        torch = _torch()
        io = _lib.DemodIO()
        rc = self._lib.mifsk_pipeline_outputs_get(self.handle, int(ticket), C.byref(io))
        if rc != 0:
            raise RuntimeError("no output sets: call outputs() first (%d)" % rc)
        ns, fc, ec = self._shape
        dev = torch.device("cuda", torch.cuda.current_device())

        def view(ptr, shape, typestr, dtype):
            return torch.as_tensor(_DevArray(ptr, shape, typestr), device=dev).view(dtype) if ptr else None
        out = {"nframes": view(io.d_nframes, (ns,), "<i4", torch.int32),
               "nbytes": view(io.d_nbytes, (ns,), "<i4", torch.int32),
               "status": view(io.d_status, (ns,), "<i4", torch.int32)}
        if io.d_bytes:
            out["bytes"] = view(io.d_bytes, (ns, fc), "|u1", torch.uint8)
        if io.d_bits:
            out["bits"] = view(io.d_bits, (ns, fc), "<i8", torch.int64)
        if io.d_frames:
            out["frames"] = view(io.d_frames, (ns, fc, FRAME_DTYPE.itemsize), "|u1", torch.uint8)
        if io.d_episodes:
            out["episodes"] = view(io.d_episodes, (ns, int(io.episodes_cap), EPISODE_DTYPE.itemsize), "|u1", torch.uint8)
            out["nepisodes"] = view(io.d_nepisodes, (ns,), "<i4", torch.int32)
        self._views[lane] = out
        return out

    def submit(self, cfg, samples, nsamples=None, after="current", ring_exact=False, engine=None):
        """`after`: the torch stream the batch was produced on ("current": torch's current
        stream), or None when nothing has to be waited for."""
        torch = _torch()
        assert samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 2 and samples.stride(1) == 1
        nstreams, width = samples.shape
        _check_nsamples(torch, nsamples, nstreams)
        assert nstreams > 1 or width % 4 == 0
        io = _lib.DemodIO()
        io.d_samples = samples.data_ptr()
        io.stream_stride = samples.stride(0) if nstreams > 1 else int(width)
        io.d_nsamples = nsamples.data_ptr() if nsamples is not None else None
        io.nsamples = int(width)
        io.nstreams = nstreams
        io.flags = (_lib.IO_RING_EXACT if ring_exact else 0) | \
            (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | \
            (_lib.IO_ENGINE_WAVE if engine == "wave" else 0)
        if after == "current":
            after = torch.cuda.current_stream()
        prod = _lib.PIPELINE_NO_PRODUCER if after is None else C.c_void_p(after.cuda_stream)
        t = C.c_uint64(0)
        rc = self._lib.mifsk_pipeline_submit(self.handle, C.byref(cfg), C.byref(io), prod, C.byref(t))
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_submit failed: %d" % rc)
        return int(t.value)

    def next_ticket(self):
        return int(self._lib.mifsk_pipeline_next_ticket(self.handle))

    def wait(self, ticket):
        rc = self._lib.mifsk_pipeline_wait(self.handle, int(ticket))
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_wait(%d) failed: %d" % (ticket, rc))

    def join(self, ticket, stream=None):
        """torch stream `stream` (default: the current one) waits for pass `ticket` on the device"""
        torch = _torch()
        rc = self._lib.mifsk_pipeline_join(self.handle, int(ticket), _stream_ptr(torch, stream))
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_join(%d) failed: %d" % (ticket, rc))

    def drain(self):
        rc = self._lib.mifsk_pipeline_drain(self.handle)
        if rc != 0:
            raise RuntimeError("mifsk_pipeline_drain failed: %d" % rc)

    def stream(self, ticket):
        """the lane's HIP stream as a torch stream (what a gather of that pass's results is queued on)"""
        torch = _torch()
        return torch.cuda.ExternalStream(int(self._lib.mifsk_pipeline_stream(self.handle, int(ticket))))

    def close(self):
        if self.handle:
            self._views = {}
            self._lib.mifsk_pipeline_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ByteGatherer:
    """Overlapped gather of decoded bytes to rank 0 for a pipelined caller
    (bench.py): start(bytes, nbytes) posts one grouped send/recv -- every peer
    sends on its own link to the root (RCCL over xGMI on GPUs, gloo in the CPU
    tests) -- and returns the work handles; the caller waits on them before it
    reuses the buffers it passed.  On the root, received(r) is what peer r >= 1
    sent in the most recently STARTED gather, once its handles have been waited
    on; the root keeps `slots` sets of receive buffers (default two) and cycles
    through them like the caller does with its outputs, so that with that many
    gathers in flight an older one's data is not overwritten by a newer one's
    arrival (received(r, slot) names a set: the k-th start() fills set k % slots).

    cols: how many columns of the [nstreams, frames_cap] byte buffer can hold
    data at all (the caller knows its streams' lengths: mifsk_max_frames of the
    longest) -- only those are shipped, from a narrow staging copy made on the
    caller's stream (`slots` of them, used in turn like the caller's buffers).
    rows: streams per rank when the shards differ in size (default: all ranks
    hold as many as this one).
    loopback (world size 1 only): the one rank sends to ITSELF and receives it in the same
    group -- the whole transport (narrow staging copy, grouped isend/irecv, receive sets) on one
    GPU, which is how the N > 1 step structure is exercised where only one GPU is there;
    received(0) is then what it sent itself."""

    def __init__(self, dist, rank, world, cols=None, rows=None, slots=2, loopback=False):
        self.dist, self.rank, self.world = dist, rank, world
        self.cols, self.rows = cols, rows
        self.loopback = bool(loopback) and world == 1
        self.slots = max(2, int(slots))
        self._rx = [None] * self.slots
        self._tx = [None] * self.slots
        self._k = 0
        self._last = 0

    def bytes_per_peer(self, nstreams):
        """what one peer sends per step (bytes + counts)"""
        return int(nstreams) * (int(self.cols) if self.cols else 0) + 4 * int(nstreams)

    def start(self, local_bytes, local_nbytes):
        dist = self.dist
        if self.world == 1 and not self.loopback:
            return []
        cols = local_bytes.shape[1] if self.cols is None else min(int(self.cols), local_bytes.shape[1])
        self.cols = cols
        if self.loopback:
            torch = _torch()
            slot = self._k % self.slots
            self._k += 1
            self._last = slot
            if self._rx[slot] is None:
                self._rx[slot] = [(torch.empty((local_bytes.shape[0], cols), dtype=local_bytes.dtype, device=local_bytes.device),
                                   torch.empty_like(local_nbytes))]
                self._tx[slot] = torch.empty((local_bytes.shape[0], cols), dtype=local_bytes.dtype,
                                             device=local_bytes.device)
            self._tx[slot].copy_(local_bytes[:, :cols])
            rb, rn = self._rx[slot][0]
            ops = [dist.P2POp(dist.irecv, rb, 0), dist.P2POp(dist.irecv, rn, 0),
                   dist.P2POp(dist.isend, self._tx[slot], 0), dist.P2POp(dist.isend, local_nbytes, 0)]
            return dist.batch_isend_irecv(ops)
        if self.rank == 0:
            slot = self._k % self.slots
            self._k += 1
            self._last = slot
            if self._rx[slot] is None:
                torch = _torch()
                rows = self.rows or [local_bytes.shape[0]] * self.world
                self._rx[slot] = [(torch.empty((rows[r], cols), dtype=local_bytes.dtype, device=local_bytes.device),
                                   torch.empty((rows[r],), dtype=local_nbytes.dtype, device=local_nbytes.device))
                                  for r in range(1, self.world)]
            rx = self._rx[slot]
            ops = []
            for r in range(1, self.world):
                ops.append(dist.P2POp(dist.irecv, rx[r - 1][0], r))
                ops.append(dist.P2POp(dist.irecv, rx[r - 1][1], r))
        else:
            tx = local_bytes
            if cols != local_bytes.shape[1]:
                b = self._k % self.slots
                self._k += 1
                if self._tx[b] is None:
                    self._tx[b] = _torch().empty((local_bytes.shape[0], cols), dtype=local_bytes.dtype,
                                                 device=local_bytes.device)
                self._tx[b].copy_(local_bytes[:, :cols])
                tx = self._tx[b]
            ops = [dist.P2POp(dist.isend, tx, 0), dist.P2POp(dist.isend, local_nbytes, 0)]
        return dist.batch_isend_irecv(ops)

    def received(self, r, slot=None):
        return self._rx[self._last if slot is None else slot % self.slots][0 if self.loopback else r - 1]


class NativeGatherer:
    """mifsk_gather_*: the same gather behind the C ABI (csrc/mifsk_gather.cpp: RCCL opened with
    dlopen, a communicator of the library's own).  Interface of ByteGatherer -- start(bytes,
    nbytes) enqueues ONE gather on torch's current stream and returns no handles (it is complete
    when that stream reaches the point behind the call); received(r, slot) on rank 0 is what peer
    r sent in the gather that filled set `slot` (default: the most recently started one), as torch
    views of the library's receive set.

    `dist`: a torch.distributed module whose default group carries rank 0's id to the others
    (None at world size 1); the communicator itself is made by the library."""

    def __init__(self, dist, rank, world, cols=None, rows=None, slots=2, loopback=False, device=-1):
        torch = _torch()
        self._lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self.cols, self.rows = cols, rows
        self.loopback = bool(loopback) and world == 1
        self.slots = max(2, int(slots))
        self._tickets = []
        ident = None
        if world > 1 or self.loopback:
            buf = (C.c_ubyte * _lib.GATHER_ID_BYTES)()
            if rank == 0:
                rc = self._lib.mifsk_gather_unique_id(buf)
                if rc != 0:
                    raise RuntimeError("mifsk_gather_unique_id failed: %d" % rc)
            if world > 1:
                # (through the job's own process group: on the device for RCCL, on the host for gloo)
                dev = "cuda" if str(dist.get_backend()) == "nccl" else "cpu"
                t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
                dist.broadcast(t, src=0)
                buf = (C.c_ubyte * _lib.GATHER_ID_BYTES)(*t.cpu().tolist())
            ident = buf
        h = C.c_void_p()
        rc = self._lib.mifsk_gather_create(C.byref(h), ident, self.rank, self.world, int(device), self.slots,
                                           _lib.GATHER_LOOPBACK if self.loopback else 0)
        if rc != 0:
            raise RuntimeError("mifsk_gather_create failed: %d" % rc)
        self.handle = h

    def info(self):
        gi = _lib.GatherInfo()
        self._lib.mifsk_gather_info_get(self.handle, C.byref(gi))
        return {k: int(getattr(gi, k)) for k, _ in gi._fields_}

    def bytes_per_peer(self, nstreams):
        return int(nstreams) * (int(self.cols) if self.cols else 0) + 4 * int(nstreams)

    def start(self, local_bytes, local_nbytes):
        torch = _torch()
        cols = local_bytes.shape[1] if self.cols is None else min(int(self.cols), local_bytes.shape[1])
        self.cols = cols
        assert local_bytes.dtype == torch.uint8 and local_nbytes.dtype == torch.int32 and local_bytes.stride(1) == 1
        rows = None
        if self.rows is not None and not self.loopback:
            rows = (C.c_int * self.world)(*[int(r) for r in self.rows])
        tk = C.c_uint64()
        rc = self._lib.mifsk_gather_start(self.handle, C.c_void_p(local_bytes.data_ptr()), int(local_bytes.stride(0)),
                                          C.c_void_p(local_nbytes.data_ptr()), int(local_bytes.shape[0]), cols, rows,
                                          _stream_ptr(torch, torch.cuda.current_stream()), C.byref(tk))
        if rc != 0:
            raise RuntimeError("mifsk_gather_start failed: %d" % rc)
        self._tickets.append(int(tk.value))
        del self._tickets[:-self.slots]
        return []

    def received(self, r, slot=None):
        torch = _torch()
        tk = self._tickets[-1]
        if slot is not None:
            tk = [t for t in self._tickets if t % self.slots == slot % self.slots][-1]
        pb, pn, rows, cols = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_int()
        rc = self._lib.mifsk_gather_received(self.handle, tk, 0 if self.loopback else int(r), C.byref(pb), C.byref(pn),
                                             C.byref(rows), C.byref(cols))
        if rc != 0:
            raise RuntimeError("mifsk_gather_received failed: %d" % rc)
        dev = torch.device("cuda", torch.cuda.current_device())
        b = torch.as_tensor(_DevArray(pb.value, (rows.value, cols.value), "|u1"), device=dev)
        n = torch.as_tensor(_DevArray(pn.value, (rows.value,), "<i4"), device=dev)
        return b, n

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mifsk_gather_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:		# noqa: BLE001
            pass


DECODERS = {"ascii8": 0, "baudot": 1, "binary": 2, "callerid": 3, "uic-ground": 4, "uic-train": 5}
TEXT_PRINT_FILTER = 1
TEXT_QUIET = 2


class DataBits:
    """One databits decoder with its state (include/mifsk.h, reference
    src/databits.h:49-92): decode(bits, n_databits) -> bytes, reset()."""

    def __init__(self, decoder):
        self._lib = _lib.load()
        self.h = C.c_void_p()
        kind = DECODERS[decoder] if isinstance(decoder, str) else int(decoder)
        rc = self._lib.mifsk_databits_create(C.byref(self.h), kind)
        if rc != 0:
            raise ValueError("mifsk_databits_create -> %d" % rc)
        self._buf = C.create_string_buffer(4096)

    def reset(self):
        self._lib.mifsk_databits_reset(self.h)

    def decode(self, bits, n_databits):
        n = self._lib.mifsk_databits_decode(self.h, self._buf, 4096, int(bits), int(n_databits))
        return self._buf.raw[:n]

    def __del__(self):
        try:
            if self.h:
                self._lib.mifsk_databits_destroy(self.h)
                self.h = None
        except Exception:
            pass


def stream_text(cfg, bits, episodes, print_filter=False, quiet=False, b_mark=None):
    """What minimodem writes for one stream: (stdout bytes, stderr str) from the
    frame data bits (numpy uint64, loop order) and its episodes (EPISODE_DTYPE);
    host post-pass mifsk_stream_text (reference src/minimodem.c:253-291,1336-1461).
    The "### CARRIER ... @ f Hz" line is printed from each episode's own b_mark
    (`b_mark` is accepted for callers of the older signature and ignored)."""
    lib = _lib.load()
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    episodes = np.ascontiguousarray(episodes, dtype=EPISODE_DTYPE)
    flags = (TEXT_PRINT_FILTER if print_filter else 0) | (TEXT_QUIET if quiet else 0)
    out_cap = 64 + 320 * max(1, bits.shape[0])
    err_cap = 256 + 512 * max(1, episodes.shape[0])
    out = C.create_string_buffer(out_cap)
    err = C.create_string_buffer(err_cap)
    nout, nerr = C.c_size_t(0), C.c_size_t(0)
    rc = lib.mifsk_stream_text(C.byref(cfg), bits.ctypes.data, bits.shape[0],
                               episodes.ctypes.data, episodes.shape[0], flags,
                               out, out_cap, C.byref(nout), err, err_cap, C.byref(nerr))
    if rc != 0:
        raise RuntimeError("mifsk_stream_text -> %d" % rc)
    return out.raw[:nout.value], err.raw[:nerr.value].decode("latin-1")


def wav_parse(data):
    """RIFF/WAVE header of a `--rx --file` input (mifsk_wav_parse): dict with
    sample_rate, channels, bits_per_sample, is_float, data_offset, nframes."""
    info = _lib.WavInfo()
    rc = _lib.load().mifsk_wav_parse(data, len(data), C.byref(info))
    if rc != 0:
        raise ValueError("mifsk_wav_parse -> %d" % rc)
    return {k: getattr(info, k) for k, _ in info._fields_}


def ingest_s16(ctx, pcm, nsamples=None, rxnoise=0.0, stride=None, stream=None):
    """libsndfile's S16 -> float conversion (+ the --Xrxnoise term) on the device:
    pcm is a torch.int16 CUDA tensor [nstreams, width]; returns float32
    [nstreams, stride] (stride defaults to width rounded up to a multiple of 4)."""
    torch = _torch()
    assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.dim() == 2 and pcm.stride(1) == 1
    nstreams, width = pcm.shape
    _check_nsamples(torch, nsamples, nstreams)
    if stride is None:
        stride = (width + 3) & ~3
    out = torch.empty((nstreams, stride), dtype=torch.float32, device=pcm.device)
    rc = _lib.load().mifsk_ingest_s16(
        ctx.handle, C.c_void_p(pcm.data_ptr()), pcm.stride(0) if nstreams > 1 else width,
        C.c_void_p(out.data_ptr()), stride,
        C.c_void_p(nsamples.data_ptr()) if nsamples is not None else None, int(width), nstreams,
        C.c_float(rxnoise), _stream_ptr(torch, stream))
    if rc != 0:
        raise RuntimeError("mifsk_ingest_s16 -> %d" % rc)
    return out


def ingest_rxnoise(ctx, samples, rxnoise, nsamples=None, stream=None):
    """The --Xrxnoise term applied in place to float32 CUDA samples [nstreams, stride]."""
    torch = _torch()
    assert samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 2
    _check_nsamples(torch, nsamples, samples.shape[0])
    rc = _lib.load().mifsk_ingest_rxnoise_f32(
        ctx.handle, C.c_void_p(samples.data_ptr()), samples.stride(0),
        C.c_void_p(nsamples.data_ptr()) if nsamples is not None else None, int(samples.shape[1]),
        samples.shape[0], C.c_float(rxnoise), _stream_ptr(torch, stream))
    if rc != 0:
        raise RuntimeError("mifsk_ingest_rxnoise_f32 -> %d" % rc)
    return samples


def synthesize_batch(ctx, cfg, words, nwords=None, lut=4096, amplitude=1.0, leading_silence=0,
                     s16=False, stride=None, stream=None):
    """The transmitter for a whole batch on the device (mifsk_tx_synthesize_batch):
    words is a torch.uint8 CUDA tensor [nstreams, max_words]; nwords / leading_silence
    may be int32 CUDA tensors [nstreams] or scalars.  Returns (samples float32
    [nstreams, stride], lengths int32 [nstreams]); rows are zero after their length."""
    torch = _torch()
    assert words.is_cuda and words.dtype == torch.uint8 and words.dim() == 2
    nstreams, width = words.shape
    nw_t = nwords if torch.is_tensor(nwords) else None
    nw_u = int(width if nwords is None else (0 if nw_t is not None else nwords))
    ls_t = leading_silence if torch.is_tensor(leading_silence) else None
    ls_u = 0 if ls_t is not None else int(leading_silence)
    if stride is None:
        max_lead = int(ls_t.max()) if ls_t is not None and nstreams else ls_u
        max_words = int(nw_t.max()) if nw_t is not None and nstreams else nw_u
        one = np.zeros(max(1, max_words), np.uint8)
        n = _lib.load().mifsk_tx_synthesize(C.byref(cfg), one.ctypes.data, max_words, lut,
                                            C.c_float(amplitude), max_lead, 0, None, 0)
        stride = (max(int(n), 4) + 3) & ~3
    with _on(_torch(), stream):
        out = torch.empty((nstreams, stride), dtype=torch.float32, device=words.device)
        lens = torch.zeros(nstreams, dtype=torch.int32, device=words.device)
    rc = _lib.load().mifsk_tx_synthesize_batch(
        ctx.handle, C.byref(cfg), C.c_void_p(words.data_ptr()),
        words.stride(0) if nstreams > 1 else width,
        C.c_void_p(nw_t.data_ptr()) if nw_t is not None else None, nw_u, nstreams, int(lut),
        C.c_float(amplitude), C.c_void_p(ls_t.data_ptr()) if ls_t is not None else None, ls_u,
        1 if s16 else 0, C.c_void_p(out.data_ptr()), stride, C.c_void_p(lens.data_ptr()),
        _stream_ptr(torch, stream))
    if rc != 0:
        raise RuntimeError("mifsk_tx_synthesize_batch -> %d" % rc)
    return out, lens


_PINNED = {}


def host_alloc(shape, dtype=np.float32):
    """A numpy array in page-locked host memory (mifsk_host_alloc): input rows the host entry
    points copy by DMA from where they are.  Release with host_free()."""
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape)) * dtype.itemsize
    ptr = _lib.load().mifsk_host_alloc(max(nbytes, 1))
    if not ptr:
        raise MemoryError("mifsk_host_alloc(%d)" % nbytes)
    buf = (C.c_char * max(nbytes, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _PINNED[arr.ctypes.data] = ptr
    return arr


def host_free(arr):
    ptr = _PINNED.pop(arr.ctypes.data, None)
    if ptr:
        _lib.load().mifsk_host_free(ptr)


def max_episodes(cfg, nsamples):
    return int(_lib.load().mifsk_max_episodes(C.byref(cfg), int(nsamples)))


def demod_batch_host(ctx, cfg, samples, nsamples=None, frames_cap=None, episodes_cap=8,
                     ring_exact=False, want=("bytes", "bits", "frames", "episodes"), rxnoise=0.0,
                     stats=False, engine=None):
    """mifsk_demod_batch_host[_ex]: the whole batch from HOST memory in one call -- chunks of
    streams cross PCIe on a copy stream while the chunk before is demodulated and the one
    before that is copied back.  samples is a float32 numpy array [nstreams, stride], or
    int16 (PCM16 as a WAV file holds it: MIFSK_IO_HOST_S16, converted on the device); rows
    from host_alloc() are copied by DMA from where they are.  Returns a dict of numpy arrays
    (plus "stats" when asked).  `ctx` may be a list of Contexts (one per GPU):
    mifsk_demod_batch_host_multi shards the streams over them inside this one process."""
    lib = _lib.load()
    s16 = samples.dtype == np.int16
    if not (s16 or (samples.dtype == np.float32 and samples.flags.c_contiguous)):
        samples = np.ascontiguousarray(samples, dtype=np.float32)
    assert samples.ndim == 2 and samples.flags.c_contiguous
    nstreams, stride = samples.shape
    if frames_cap is None:
        frames_cap = max_frames(cfg, stride)
    res = {"nframes": np.zeros(nstreams, np.uint32), "status": np.zeros(nstreams, np.uint32),
           "carrier_band": np.full(nstreams, -1, np.int32)}
    if "bytes" in want:
        res["bytes"] = np.zeros((nstreams, frames_cap), np.uint8)
        res["nbytes"] = np.zeros(nstreams, np.uint32)
    if "bits" in want:
        res["bits"] = np.zeros((nstreams, frames_cap), np.uint64)
    if "frames" in want:
        res["frames"] = np.zeros((nstreams, frames_cap), FRAME_DTYPE)
    if "episodes" in want:
        res["episodes"] = np.zeros((nstreams, episodes_cap), EPISODE_DTYPE)
        res["nepisodes"] = np.zeros(nstreams, np.uint32)

    def ptr(name):
        return res[name].ctypes.data if name in res else None

    io = _lib.DemodIO()
    io.d_samples = samples.ctypes.data
    io.stream_stride = stride
    if nsamples is not None:
        nsamples = np.ascontiguousarray(nsamples, dtype=np.uint32)
        io.d_nsamples = nsamples.ctypes.data
    io.nsamples = stride
    io.nstreams = nstreams
    io.d_bytes = ptr("bytes")
    io.d_nbytes = ptr("nbytes")
    io.d_bits = ptr("bits")
    io.d_frames = ptr("frames")
    io.d_nframes = ptr("nframes")
    io.frames_cap = frames_cap
    io.d_episodes = ptr("episodes")
    io.d_nepisodes = ptr("nepisodes")
    io.episodes_cap = episodes_cap
    io.d_status = ptr("status")
    io.d_carrier_band = ptr("carrier_band")
    io.flags = (_lib.IO_RING_EXACT if ring_exact else 0) | (_lib.IO_HOST_S16 if s16 else 0) | \
        (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | \
        (_lib.IO_ENGINE_WAVE if engine == "wave" else 0)
    if isinstance(ctx, (list, tuple)):
        assert not s16 and not rxnoise
        handles = (C.c_void_p * len(ctx))(*[c.handle for c in ctx])
        rc = lib.mifsk_demod_batch_host_multi(handles, len(ctx), C.byref(cfg), C.byref(io))
    else:
        st = _lib.HostStats()
        rc = lib.mifsk_demod_batch_host_ex(ctx.handle, C.byref(cfg), C.byref(io), C.c_float(rxnoise),
                                           C.byref(st))
        if stats:
            res["stats"] = {k: getattr(st, k) for k, _ in st._fields_ if k != "reserved"}
    if rc != 0:
        raise RuntimeError("mifsk_demod_batch_host failed: %d" % rc)
    return res


def demod_files(ctx, paths, baudmode="1200", rxnoise=0.0, ring_exact=False, want_frames=False,
                engine=None, **opts):
    """mifsk_demod_files: `minimodem --rx --file F <baudmode>` for a list of WAV files (PCM16 or
    float32, any mix of lengths and sample rates) as batches on the device: raw samples are
    pread() into pinned memory by worker threads, cross PCIe as they are in the file and are
    converted there.  Returns (list of per-file dicts, stats dict); a file that could not be
    decoded has {"error": -errno}.  Each dict carries the RxConfig it was decoded with."""
    lib = _lib.load()
    a = ModemArgs()
    lib.mifsk_modem_args_default(C.byref(a))
    a.baudmode = str(baudmode).encode()
    if "sync_byte" in opts:
        a.have_sync_byte = 1
    for k, v in opts.items():
        if not hasattr(a, k):
            raise TypeError("unknown modem option %r" % k)
        setattr(a, k, v)
    enc = [os.fsencode(p) for p in paths]
    arr = (C.c_char_p * max(1, len(enc)))(*enc)
    flags = (_lib.IO_RING_EXACT if ring_exact else 0) | (_lib.FILES_WANT_FRAMES if want_frames else 0) | \
        (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | \
        (_lib.IO_ENGINE_WAVE if engine == "wave" else 0)
    h = C.c_void_p()
    rc = lib.mifsk_demod_files(ctx.handle, C.byref(a), arr, len(enc), C.c_float(rxnoise), flags, C.byref(h))
    if rc != 0:
        if h:
            lib.mifsk_files_free(h)
        raise RuntimeError("mifsk_demod_files failed: %d" % rc)
    out = []
    try:
        for i in range(lib.mifsk_files_count(h)):
            fr = lib.mifsk_files_get(h, i).contents
            d = {"error": fr.error, "path": paths[i],
                 "info": {k: getattr(fr.info, k) for k, _ in fr.info._fields_}}
            if fr.error == 0:
                cfg = RxConfig()
                C.memmove(C.byref(cfg), fr.cfg, C.sizeof(RxConfig))
                d["cfg"] = cfg
                d["status"] = fr.status
                d["carrier_band"] = fr.carrier_band
                d["bits"] = np.ctypeslib.as_array((C.c_uint64 * max(1, fr.nframes)).from_address(fr.bits))[:fr.nframes].copy()
                d["bytes"] = bytes((C.c_uint8 * max(1, fr.nbytes)).from_address(fr.bytes))[:fr.nbytes]
                d["episodes"] = np.frombuffer(
                    (C.c_char * (max(1, fr.nepisodes) * EPISODE_DTYPE.itemsize)).from_address(fr.episodes),
                    dtype=EPISODE_DTYPE)[:fr.nepisodes].copy()
                if fr.frames:
                    d["frames"] = np.frombuffer(
                        (C.c_char * (max(1, fr.nframes) * FRAME_DTYPE.itemsize)).from_address(fr.frames),
                        dtype=FRAME_DTYPE)[:fr.nframes].copy()
            out.append(d)
        st = lib.mifsk_files_stats(h).contents
        stats = {k: getattr(st, k) for k, _ in st._fields_ if k != "reserved"}
    finally:
        lib.mifsk_files_free(h)
    return out, stats


# ---------------------------------------------------------------------------
# streams that arrive in pieces (mifsk_demod_slab)
# ---------------------------------------------------------------------------

STATE_DTYPE = np.dtype([("base", "<u8"), ("rp", "<u8"), ("carrier_nsamples", "<u8"),
                        ("nframes_total", "<u8"), ("advance", "<u4"), ("flags", "<u4"),
                        ("confidence_total", "<f4"), ("amplitude_total", "<f4"),
                        ("nframes_decoded", "<u4"), ("noconfidence", "<u4"),
                        ("track_amplitude", "<f4"), ("peak_confidence", "<f4"),
                        ("carrier_band", "<i4"), ("first_band", "<i4"), ("b_mark", "<u4"),
                        ("ep_b_mark", "<u4"), ("ep_first", "<u4"), ("nbytes_total", "<u4"),
                        ("nepisodes_total", "<u4"), ("status", "<u4")])
assert STATE_DTYPE.itemsize == 96
STATE_FINISHED = 4


class SlabSession:
    """A batch of streams fed in pieces (mifsk_demod_slab; reference: the half-buffer refills of
    src/minimodem.c:1144-1174).  feed(new_samples, final) appends each stream's new samples
    behind what the receive loop has not passed yet, runs the loop as far as the data allows
    and returns the frames / bytes / episodes that this made; the loop's state lives in device
    memory between calls, the unconsumed tail of every stream on the host.  Whatever the cuts,
    the concatenated output is the single call's, bit for bit."""

    def __init__(self, ctx, cfg, nstreams, episodes_cap=16, engine=None, ring_exact=False):
        """engine: None (the library chooses, as demod_batch does), "wave" or "workgroup"; may
        be changed between feeds (self.engine): the state record is the same for both.
        ring_exact: the reference's buffer semantics (mifsk_demod_slab_ring): its samplebuf
        cells persist in device memory between the feeds."""
        torch = _torch()
        self.ctx, self.cfg, self.n = ctx, cfg, int(nstreams)
        self.engine = engine
        self.ring = None
        if ring_exact:
            nf = int(_lib.load().mifsk_ring_floats(C.byref(cfg)))
            assert nf > 0
            self.ring = torch.zeros((self.n, nf), dtype=torch.float32, device="cuda")
        self.episodes_cap = episodes_cap
        self.state = torch.zeros((self.n, STATE_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
        self.origin = np.zeros(self.n, np.uint64)
        self.tail = [np.zeros(0, np.float32) for _ in range(self.n)]

    def feed(self, new, final=False, stream=None):
        torch = _torch()
        lib = _lib.load()
        assert len(new) == self.n
        for i, x in enumerate(new):
            if x is not None and len(x):
                self.tail[i] = np.concatenate([self.tail[i], np.asarray(x, np.float32)])
        if stream is not None:
            # (state and ring were made -- and on the first feed zero-filled -- on the stream that
            # was current then: this stream starts behind it)
            stream.wait_stream(torch.cuda.current_stream())
        width = (max([len(t) for t in self.tail] + [4]) + 3) & ~3
        host = np.zeros((self.n, width), np.float32)
        lens = np.zeros(self.n, np.int32)
        for i, t in enumerate(self.tail):
            host[i, :len(t)] = t
            lens[i] = len(t)
        with _on(torch, stream):
            d = torch.from_numpy(host).cuda()
            dl = torch.from_numpy(lens).cuda()
            do = torch.from_numpy(self.origin.view(np.int64).copy()).cuda()
            fc = max_frames(self.cfg, width)
            dev = d.device
            out = {"nframes": torch.zeros(self.n, dtype=torch.int32, device=dev),
                   "status": torch.zeros(self.n, dtype=torch.int32, device=dev),
                   "bytes": torch.zeros((self.n, fc), dtype=torch.uint8, device=dev),
                   "nbytes": torch.zeros(self.n, dtype=torch.int32, device=dev),
                   "bits": torch.zeros((self.n, fc), dtype=torch.int64, device=dev),
                   "frames": torch.zeros((self.n, fc, FRAME_DTYPE.itemsize), dtype=torch.uint8, device=dev),
                   "episodes": torch.zeros((self.n, self.episodes_cap, EPISODE_DTYPE.itemsize),
                                           dtype=torch.uint8, device=dev),
                   "nepisodes": torch.zeros(self.n, dtype=torch.int32, device=dev),
                   "carrier_band": torch.full((self.n,), -1, dtype=torch.int32, device=dev)}
        io = _lib.DemodIO()
        io.d_samples = d.data_ptr()
        io.stream_stride = d.stride(0) if self.n > 1 else width
        io.d_nsamples = dl.data_ptr()
        io.nsamples = width
        io.nstreams = self.n
        io.d_bytes = out["bytes"].data_ptr()
        io.d_nbytes = out["nbytes"].data_ptr()
        io.d_bits = out["bits"].data_ptr()
        io.d_frames = out["frames"].data_ptr()
        io.d_nframes = out["nframes"].data_ptr()
        io.frames_cap = fc
        io.d_episodes = out["episodes"].data_ptr()
        io.d_nepisodes = out["nepisodes"].data_ptr()
        io.episodes_cap = self.episodes_cap
        io.d_status = out["status"].data_ptr()
        io.d_carrier_band = out["carrier_band"].data_ptr()
        io.flags = (_lib.IO_ENGINE_WORKGROUP if self.engine == "workgroup" else 0) | \
            (_lib.IO_ENGINE_WAVE if self.engine == "wave" else 0)
        if self.ring is not None:
            io.flags = _lib.IO_RING_EXACT
            rc = lib.mifsk_demod_slab_ring(self.ctx.handle, C.byref(self.cfg), C.byref(io),
                                           C.c_void_p(self.state.data_ptr()), C.c_void_p(do.data_ptr()),
                                           C.c_void_p(self.ring.data_ptr()), 1 if final else 0,
                                           _stream_ptr(torch, stream))
        else:
            rc = lib.mifsk_demod_slab(self.ctx.handle, C.byref(self.cfg), C.byref(io),
                                      C.c_void_p(self.state.data_ptr()), C.c_void_p(do.data_ptr()),
                                      1 if final else 0, _stream_ptr(torch, stream))
        if rc != 0:
            raise RuntimeError("mifsk_demod_slab failed: %d" % rc)
        torch.cuda.synchronize()
        res = results_to_host(out)
        st = self.state.cpu().numpy().view(STATE_DTYPE).reshape(self.n)
        for i in range(self.n):
            drop = int(st["base"][i]) - int(self.origin[i])
            if drop > 0:
                self.tail[i] = self.tail[i][drop:]
                self.origin[i] = st["base"][i]
        res["state"] = st.copy()
        return res


class Session:
    """mifsk_session_*: SlabSession's job behind the C ABI (csrc/mifsk_session.cpp) -- the
    unconsumed tails, origins, loop state, RING cells and output arrays are the library's.
    feed(new, final) returns one dict per stream of what THIS feed made of it (numpy copies)."""

    def __init__(self, ctx, cfg, nstreams, engine=None, ring_exact=False, want_frames=True):
        self._lib = _lib.load()
        self.n = int(nstreams)
        flags = (_lib.IO_ENGINE_WORKGROUP if engine == "workgroup" else 0) | (_lib.IO_ENGINE_WAVE if engine == "wave" else 0) \
            | (_lib.IO_RING_EXACT if ring_exact else 0) | (_lib.SESSION_WANT_FRAMES if want_frames else 0)
        h = C.c_void_p()
        rc = self._lib.mifsk_session_create(C.byref(h), ctx.handle, C.byref(cfg), self.n, flags)
        if rc != 0:
            raise RuntimeError("mifsk_session_create failed: %d" % rc)
        self.handle = h

    def feed(self, new, final=False):
        assert len(new) == self.n
        keep = [np.ascontiguousarray(x, dtype=np.float32) if x is not None and len(x) else None for x in new]
        ptrs = (C.c_void_p * self.n)(*[k.ctypes.data if k is not None else None for k in keep])
        cnts = (C.c_uint32 * self.n)(*[len(k) if k is not None else 0 for k in keep])
        rc = self._lib.mifsk_session_feed(self.handle, ptrs, cnts, 1 if final else 0)
        if rc != 0:
            raise RuntimeError("mifsk_session_feed failed: %d" % rc)
        out = []
        for i in range(self.n):
            r = self._lib.mifsk_session_get(self.handle, i).contents

            def arr(ptr, count, dtype):
                if not ptr or not count:
                    return np.zeros(0, dtype)
                nb = count * np.dtype(dtype).itemsize
                return np.frombuffer(C.string_at(ptr, nb), dtype=dtype).copy()
            out.append({"frames": arr(r.frames, r.nframes, FRAME_DTYPE), "bits": arr(r.bits, r.nframes, np.uint64),
                        "bytes": arr(r.bytes, r.nbytes, np.uint8).tobytes(),
                        "episodes": arr(r.episodes, r.nepisodes, EPISODE_DTYPE), "status": int(r.status),
                        "carrier_band": int(r.carrier_band), "consumed": int(r.consumed), "finished": bool(r.finished),
                        "pending": int(self._lib.mifsk_session_pending(self.handle, i))})
        return out

    def close(self):
        if getattr(self, "handle", None):
            self._lib.mifsk_session_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:		# noqa: BLE001
            pass


def scan_plan(cfg, kind):
    """mifsk_scan_plan_get: the shared-segment plan of scan `kind` (2 * fine + carrier) as a dict
    of numpy arrays, or None when that scan correlates every window by itself."""
    sp = _lib.ScanPlan()
    rc = _lib.load().mifsk_scan_plan_get(C.byref(cfg), int(kind), C.byref(sp))
    if rc != 0:
        raise RuntimeError("mifsk_scan_plan_get -> %d" % rc)
    if not sp.valid:
        return None
    n, w = sp.nseg, sp.nwin
    return {"nseg": n, "npass": sp.npass, "nwin": w, "span_hi": sp.span_hi,
            "pass_len": list(sp.pass_len), "pass_min": list(sp.pass_min), "bound_c": sp.bound_c,
            "seg_rel": np.array(sp.seg_rel[:n]), "seg_len": np.array(sp.seg_len[:n]),
            "slot_seg": np.array(sp.slot_seg[:64 * sp.npass]),
            "win_first": np.array(sp.win_first[:w]), "win_count": np.array(sp.win_count[:w])}
