"""ctypes binding of libmifsk.so (the C ABI declared in include/fsk.h and
include/mifsk.h).  The library is built in-tree by `minimodem_amd.build()` /
`__graft_entry__.build()`; if it is missing or cannot be loaded this module
raises -- there is no Python or CPU fallback for the signal path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIFSK_LIBRARY selects another build of the same library (e.g. libmifsk_prof.so)
LIB_PATH = os.environ.get("MIFSK_LIBRARY") or os.path.join(_HERE, "libmifsk.so")
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


class Search(C.Structure):
    _fields_ = [
        ("sample_offset", C.c_uint64),
        ("navail", C.c_uint32),
        ("try_first", C.c_uint32),
        ("try_max", C.c_uint32),
        ("try_step", C.c_uint32),
        ("search_limit", C.c_float),
        ("use_sync_string", C.c_uint32),
    ]


class SearchResult(C.Structure):
    _fields_ = [
        ("bits", C.c_uint64),
        ("confidence", C.c_float),
        ("amplitude", C.c_float),
        ("frame_start", C.c_uint32),
        ("n_positions", C.c_uint32),
    ]


class DemodIO(C.Structure):
    _fields_ = [
        ("d_samples", C.c_void_p),
        ("stream_stride", C.c_size_t),
        ("d_nsamples", C.c_void_p),
        ("nsamples", C.c_uint32),
        ("nstreams", C.c_int),
        ("d_bytes", C.c_void_p),
        ("d_nbytes", C.c_void_p),
        ("d_bits", C.c_void_p),
        ("d_frames", C.c_void_p),
        ("d_nframes", C.c_void_p),
        ("frames_cap", C.c_size_t),
        ("d_episodes", C.c_void_p),
        ("d_nepisodes", C.c_void_p),
        ("episodes_cap", C.c_size_t),
        ("d_status", C.c_void_p),
        ("d_counters", C.c_void_p),
        ("d_carrier_band", C.c_void_p),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]

class LaunchInfo(C.Structure):
    _fields_ = [
        ("kernel", C.c_char * 64),
        ("engine", C.c_uint32), ("workgroup_size", C.c_uint32),
        ("lds_bytes_per_workgroup", C.c_uint32), ("workgroups_per_cu", C.c_uint32),
        ("lattice_mode", C.c_uint32), ("frames_per_block", C.c_uint32),
        ("compute_units", C.c_uint32),
        ("chain_groups", C.c_uint32), ("chain_chunks", C.c_uint32),
    ]


class PipelineInfo(C.Structure):
    _fields_ = [("depth_requested", C.c_uint32), ("depth", C.c_uint32),
                ("hw_queues", C.c_uint32), ("output_sets", C.c_uint32)]


class GatherInfo(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("device", C.c_int),
                ("slots", C.c_uint32), ("loopback", C.c_uint32), ("communicator", C.c_uint32)]


class SessionResult(C.Structure):
    _fields_ = [("nframes", C.c_uint32), ("nbytes", C.c_uint32), ("nepisodes", C.c_uint32),
                ("status", C.c_uint32), ("carrier_band", C.c_int32), ("finished", C.c_uint32),
                ("bits", C.c_void_p), ("bytes", C.c_void_p), ("frames", C.c_void_p),
                ("episodes", C.c_void_p), ("consumed", C.c_uint64)]


SESSION_WANT_FRAMES = 0x1000
GATHER_ID_BYTES = 128
GATHER_LOOPBACK = 1
WANT_BYTES, WANT_BITS, WANT_FRAMES, WANT_EPISODES = 1, 2, 4, 8
PIPELINE_NO_PRODUCER = C.c_void_p(-1)
IO_RING_EXACT = 1
IO_ENGINE_WORKGROUP = 2
IO_ENGINE_WAVE = 4
IO_HOST_S16 = 0x100
FILES_WANT_FRAMES = 0x1000


class HostStats(C.Structure):
    _fields_ = [("seconds_total", C.c_double), ("seconds_staging", C.c_double),
                ("bytes_h2d", C.c_uint64), ("bytes_d2h", C.c_uint64),
                ("chunks", C.c_uint32), ("streams", C.c_uint32),
                ("source_pinned", C.c_uint32), ("reserved", C.c_uint32)]


class FskPlan(C.Structure):
    """struct fsk_plan, include/fsk.h (layout of reference src/fsk.h:30-46)."""
    _fields_ = [
        ("sample_rate", C.c_float), ("f_mark", C.c_float), ("f_space", C.c_float),
        ("filter_bw", C.c_float), ("fftsize", C.c_int), ("nbands", C.c_uint),
        ("band_width", C.c_float), ("b_mark", C.c_uint), ("b_space", C.c_uint),
        ("fftplan", C.c_void_p), ("fftin", C.c_void_p), ("fftout", C.c_void_p),
    ]


class WavInfo(C.Structure):
    _fields_ = [("sample_rate", C.c_uint), ("channels", C.c_uint), ("bits_per_sample", C.c_uint),
                ("is_float", C.c_int), ("data_offset", C.c_size_t), ("nframes", C.c_size_t)]


class ScanPlan(C.Structure):
    _fields_ = [("valid", C.c_uint32), ("nseg", C.c_uint32), ("npass", C.c_uint32), ("nwin", C.c_uint32),
                ("span_hi", C.c_uint32), ("pass_len", C.c_uint32 * 2), ("pass_min", C.c_uint32 * 2),
                ("bound_c", C.c_float), ("seg_rel", C.c_uint32 * 128), ("seg_len", C.c_uint16 * 128),
                ("slot_seg", C.c_uint16 * 128), ("win_first", C.c_uint16 * 128),
                ("win_count", C.c_uint16 * 128), ("p_slot", C.c_uint32 * 128), ("p_win", C.c_uint32 * 128),
                ("p_slot_seg", C.c_uint8 * 128)]


class FileResult(C.Structure):
    _fields_ = [("error", C.c_int), ("info", WavInfo), ("cfg", C.POINTER(RxConfig)),
                ("nframes", C.c_uint32), ("nbytes", C.c_uint32), ("nepisodes", C.c_uint32),
                ("status", C.c_uint32), ("carrier_band", C.c_int32), ("reserved", C.c_uint32),
                ("bits", C.c_void_p), ("bytes", C.c_void_p), ("frames", C.c_void_p),
                ("episodes", C.c_void_p)]


assert C.sizeof(FskPlan) == 64 and FskPlan.fftplan.offset == 40
assert C.sizeof(Search) == 32 and C.sizeof(SearchResult) == 24

# every symbol include/*.h declares
EXPORTS = [
    "fsk_plan_new", "fsk_plan_destroy", "fsk_find_frame", "fsk_detect_carrier",
    "fsk_set_tones_by_bandshift",
    "mifsk_modem_args_default", "mifsk_rx_config_init", "mifsk_max_frames",
    "mifsk_stream_padding", "mifsk_ctx_create", "mifsk_ctx_destroy",
    "mifsk_ctx_device_name", "mifsk_abi_version", "mifsk_abi_sizeof", "mifsk_find_frame_batch", "mifsk_demod_plan_ex",
    "mifsk_demod_batch", "mifsk_demod_batch_host",
    "mifsk_tx_synthesize",
    "mifsk_databits_create", "mifsk_databits_destroy", "mifsk_databits_reset",
    "mifsk_databits_decode", "mifsk_databits_encode", "mifsk_shard_range",
    "mifsk_demod_batch_host_multi", "mifsk_demod_plan", "mifsk_stream_text",
    "mifsk_wav_parse", "mifsk_ingest_s16", "mifsk_ingest_rxnoise_f32",
    "mifsk_tx_synthesize_batch",
    "mifsk_demod_batch_host_ex", "mifsk_host_alloc", "mifsk_host_free", "mifsk_max_episodes",
    "mifsk_demod_files", "mifsk_files_count", "mifsk_files_get", "mifsk_files_stats",
    "mifsk_files_free", "mifsk_demod_slab", "mifsk_scan_plan_get",
    "mifsk_demod_slab_ring", "mifsk_ring_floats", "mifsk_selftest_sqrt",
    "mifsk_pipeline_create", "mifsk_pipeline_destroy", "mifsk_pipeline_info_get",
    "mifsk_pipeline_outputs_alloc", "mifsk_pipeline_outputs_get", "mifsk_pipeline_submit",
    "mifsk_pipeline_next_ticket", "mifsk_pipeline_wait", "mifsk_pipeline_join",
    "mifsk_pipeline_drain", "mifsk_pipeline_stream", "mifsk_pipeline_ctx",
    "mifsk_gather_unique_id", "mifsk_gather_create", "mifsk_gather_destroy", "mifsk_gather_info_get",
    "mifsk_gather_start", "mifsk_gather_received",
    "mifsk_session_create", "mifsk_session_destroy", "mifsk_session_feed", "mifsk_session_get",
    "mifsk_session_pending",
]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  minimodem_amd has no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mifsk_modem_args_default.argtypes = [C.POINTER(ModemArgs)]
    lib.mifsk_rx_config_init.restype = C.c_int
    lib.mifsk_rx_config_init.argtypes = [C.POINTER(RxConfig), C.POINTER(ModemArgs)]
    lib.mifsk_max_frames.restype = C.c_size_t
    lib.mifsk_max_frames.argtypes = [C.POINTER(RxConfig), C.c_size_t]
    lib.mifsk_stream_padding.restype = C.c_size_t
    lib.mifsk_stream_padding.argtypes = [C.POINTER(RxConfig)]
    lib.mifsk_ctx_create.restype = C.c_int
    lib.mifsk_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.mifsk_ctx_destroy.argtypes = [C.c_void_p]
    lib.mifsk_ctx_device_name.restype = C.c_char_p
    lib.mifsk_ctx_device_name.argtypes = [C.c_void_p]
    lib.mifsk_abi_version.restype = C.c_int
    lib.mifsk_abi_sizeof.restype = C.c_size_t
    lib.mifsk_abi_sizeof.argtypes = [C.c_char_p]
    lib.mifsk_find_frame_batch.restype = C.c_int
    lib.mifsk_find_frame_batch.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.mifsk_demod_batch.restype = C.c_int
    lib.mifsk_demod_batch.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO),
                                      C.c_void_p]
    lib.mifsk_demod_batch_host.restype = C.c_int
    lib.mifsk_demod_batch_host.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO)]
    lib.fsk_plan_new.restype = C.POINTER(FskPlan)
    lib.fsk_plan_new.argtypes = [C.c_float] * 4
    lib.fsk_plan_destroy.argtypes = [C.c_void_p]
    lib.fsk_find_frame.restype = C.c_float
    lib.fsk_find_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint,
                                   C.c_uint, C.c_float, C.c_char_p,
                                   C.POINTER(C.c_ulonglong), C.POINTER(C.c_float),
                                   C.POINTER(C.c_uint)]
    lib.fsk_detect_carrier.restype = C.c_int
    lib.fsk_detect_carrier.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_float]
    lib.fsk_set_tones_by_bandshift.argtypes = [C.c_void_p, C.c_uint, C.c_int]
    lib.mifsk_tx_synthesize.restype = C.c_long
    lib.mifsk_tx_synthesize.argtypes = [C.POINTER(RxConfig), C.c_void_p, C.c_size_t,
                                        C.c_uint, C.c_float, C.c_uint, C.c_int,
                                        C.c_void_p, C.c_size_t]
    lib.mifsk_databits_create.restype = C.c_int
    lib.mifsk_databits_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.mifsk_databits_destroy.argtypes = [C.c_void_p]
    lib.mifsk_databits_reset.argtypes = [C.c_void_p]
    lib.mifsk_databits_decode.restype = C.c_uint
    lib.mifsk_databits_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_ulonglong, C.c_uint]
    lib.mifsk_stream_text.restype = C.c_int
    lib.mifsk_stream_text.argtypes = [C.POINTER(RxConfig), C.c_void_p, C.c_uint32, C.c_void_p,
                                      C.c_uint32, C.c_uint, C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t,
                                      C.POINTER(C.c_size_t)]
    lib.mifsk_wav_parse.restype = C.c_int
    lib.mifsk_wav_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(WavInfo)]
    lib.mifsk_ingest_s16.restype = C.c_int
    lib.mifsk_ingest_s16.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_void_p]
    lib.mifsk_ingest_rxnoise_f32.restype = C.c_int
    lib.mifsk_ingest_rxnoise_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                             C.c_uint32, C.c_int, C.c_float, C.c_void_p]
    lib.mifsk_tx_synthesize_batch.restype = C.c_int
    lib.mifsk_tx_synthesize_batch.argtypes = [
        C.c_void_p, C.POINTER(RxConfig), C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_int,
        C.c_uint, C.c_float, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p,
        C.c_void_p]
    lib.mifsk_demod_plan.restype = C.c_int
    lib.mifsk_demod_plan.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.c_int, C.c_uint,
                                     C.POINTER(LaunchInfo)]
    lib.mifsk_demod_plan_ex.restype = C.c_int
    lib.mifsk_demod_plan_ex.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.c_int, C.c_uint32, C.c_uint,
                                        C.POINTER(LaunchInfo)]
    lib.mifsk_demod_batch_host_multi.restype = C.c_int
    lib.mifsk_demod_batch_host_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RxConfig),
                                                 C.POINTER(DemodIO)]
    lib.mifsk_databits_encode.restype = C.c_uint
    lib.mifsk_databits_encode.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_char]
    lib.mifsk_shard_range.restype = None
    lib.mifsk_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mifsk_demod_batch_host_ex.restype = C.c_int
    lib.mifsk_demod_batch_host_ex.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO),
                                              C.c_float, C.POINTER(HostStats)]
    lib.mifsk_host_alloc.restype = C.c_void_p
    lib.mifsk_host_alloc.argtypes = [C.c_size_t]
    lib.mifsk_host_free.restype = None
    lib.mifsk_host_free.argtypes = [C.c_void_p]
    lib.mifsk_max_episodes.restype = C.c_size_t
    lib.mifsk_max_episodes.argtypes = [C.POINTER(RxConfig), C.c_size_t]
    lib.mifsk_demod_files.restype = C.c_int
    lib.mifsk_demod_files.argtypes = [C.c_void_p, C.POINTER(ModemArgs), C.POINTER(C.c_char_p), C.c_int,
                                      C.c_float, C.c_uint, C.POINTER(C.c_void_p)]
    lib.mifsk_files_count.restype = C.c_int
    lib.mifsk_files_count.argtypes = [C.c_void_p]
    lib.mifsk_files_get.restype = C.POINTER(FileResult)
    lib.mifsk_files_get.argtypes = [C.c_void_p, C.c_int]
    lib.mifsk_files_stats.restype = C.POINTER(HostStats)
    lib.mifsk_files_stats.argtypes = [C.c_void_p]
    lib.mifsk_files_free.restype = None
    lib.mifsk_files_free.argtypes = [C.c_void_p]
    lib.mifsk_scan_plan_get.restype = C.c_int
    lib.mifsk_scan_plan_get.argtypes = [C.POINTER(RxConfig), C.c_int, C.POINTER(ScanPlan)]
    lib.mifsk_demod_slab.restype = C.c_int
    lib.mifsk_demod_slab.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO), C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p]
    lib.mifsk_demod_slab_ring.restype = C.c_int
    lib.mifsk_demod_slab_ring.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO), C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.mifsk_ring_floats.restype = C.c_size_t
    lib.mifsk_ring_floats.argtypes = [C.POINTER(RxConfig)]
    lib.mifsk_pipeline_create.restype = C.c_int
    lib.mifsk_pipeline_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int]
    lib.mifsk_pipeline_destroy.restype = None
    lib.mifsk_pipeline_destroy.argtypes = [C.c_void_p]
    lib.mifsk_pipeline_info_get.restype = C.c_int
    lib.mifsk_pipeline_info_get.argtypes = [C.c_void_p, C.POINTER(PipelineInfo)]
    lib.mifsk_pipeline_outputs_alloc.restype = C.c_int
    lib.mifsk_pipeline_outputs_alloc.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_uint]
    lib.mifsk_pipeline_outputs_get.restype = C.c_int
    lib.mifsk_pipeline_outputs_get.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(DemodIO)]
    lib.mifsk_pipeline_submit.restype = C.c_int
    lib.mifsk_pipeline_submit.argtypes = [C.c_void_p, C.POINTER(RxConfig), C.POINTER(DemodIO), C.c_void_p,
                                          C.POINTER(C.c_uint64)]
    lib.mifsk_pipeline_next_ticket.restype = C.c_uint64
    lib.mifsk_pipeline_next_ticket.argtypes = [C.c_void_p]
    lib.mifsk_pipeline_wait.restype = C.c_int
    lib.mifsk_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64]
    lib.mifsk_pipeline_join.restype = C.c_int
    lib.mifsk_pipeline_join.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    lib.mifsk_pipeline_drain.restype = C.c_int
    lib.mifsk_pipeline_drain.argtypes = [C.c_void_p]
    lib.mifsk_pipeline_stream.restype = C.c_void_p
    lib.mifsk_pipeline_stream.argtypes = [C.c_void_p, C.c_uint64]
    lib.mifsk_pipeline_ctx.restype = C.c_void_p
    lib.mifsk_pipeline_ctx.argtypes = [C.c_void_p, C.c_uint64]
    lib.mifsk_gather_unique_id.restype = C.c_int
    lib.mifsk_gather_unique_id.argtypes = [C.c_void_p]
    lib.mifsk_gather_create.restype = C.c_int
    lib.mifsk_gather_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]
    lib.mifsk_gather_destroy.restype = None
    lib.mifsk_gather_destroy.argtypes = [C.c_void_p]
    lib.mifsk_gather_info_get.restype = C.c_int
    lib.mifsk_gather_info_get.argtypes = [C.c_void_p, C.POINTER(GatherInfo)]
    lib.mifsk_gather_start.restype = C.c_int
    lib.mifsk_gather_start.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int,
                                       C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_uint64)]
    lib.mifsk_gather_received.restype = C.c_int
    lib.mifsk_gather_received.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mifsk_session_create.restype = C.c_int
    lib.mifsk_session_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(RxConfig), C.c_int, C.c_uint]
    lib.mifsk_session_destroy.restype = None
    lib.mifsk_session_destroy.argtypes = [C.c_void_p]
    lib.mifsk_session_feed.restype = C.c_int
    lib.mifsk_session_feed.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int]
    lib.mifsk_session_get.restype = C.POINTER(SessionResult)
    lib.mifsk_session_get.argtypes = [C.c_void_p, C.c_int]
    lib.mifsk_session_pending.restype = C.c_size_t
    lib.mifsk_session_pending.argtypes = [C.c_void_p, C.c_int]
    lib.mifsk_selftest_sqrt.restype = C.c_int
    lib.mifsk_selftest_sqrt.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]
    _lib = lib
    return lib
