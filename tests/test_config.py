"""Host logic (no GPU): the product's derived receive configuration
(csrc/mifsk_config.cpp) against SURVEY Appendix A (the reference's own
arithmetic) and against the oracle's independent derivation."""
import ctypes as C

import pytest

import _oracle as O
import minimodem_amd as M
from minimodem_amd import _lib

# mode kwargs -> (fftsize, b_mark, b_space, bit_nsamples, expect_data, expect_nsamples,
#                 frame_nsamples, overscan, samplebuf, (nocarrier max, step),
#                 (carrier first, max, step, fine))         -- SURVEY.md Appendix A
APPENDIX_A = [
    (dict(baudmode="1200"), 240, 6, 11, 40, "10dddddddd1", 440, 400, 20, 4000, (60, 20), (20, 50, 16, 6)),
    (dict(baudmode="300"), 960, 25, 21, 160, "10dddddddd1", 1760, 1600, 80, 4000, (240, 80), (80, 200, 66, 25)),
    (dict(baudmode="rtty"), 4800, 159, 142, 1056, "10ddddd1", 8448, 7393, 528, 19026, (1584, 528), (528, 1320, 440, 165)),
    (dict(baudmode="tdd"), 4800, 140, 180, 1056, "10ddddd1", 8448, 8449, 528, 19026, (1584, 528), (528, 1320, 440, 165)),
    (dict(baudmode="12000"), 240, 33, 83, 4, "10dddddddd1", 44, 40, 2, 4000, (6, 2), (2, 5, 1, 1)),
    (dict(baudmode="same"), 92, 4, 3, 92, "dddddddd", 737, 737, 46, 4000, (138, 46), (46, 115, 38, 14)),
    (dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400), 120, 6, 12, 20,
     "10dddddddd1", 220, 200, 10, 2000, (30, 10), (10, 25, 8, 3)),
    (dict(baudmode="0.5"), 96000, 3170, 2830, 96000, "10dddddddd1", 1056000, 960000, 48000,
     2304000, (144000, 48000), (48000, 120000, 40000, 15000)),
    (dict(baudmode="uic-ground"), 240, 7, 9, 80, "11110010" + "d" * 39, 3760, 3760, 40, 8000,
     (120, 40), (40, 100, 33, 12)),
]


@pytest.mark.parametrize("row", APPENDIX_A, ids=[r[0]["baudmode"] + str(r[0].get("sample_rate", "")) for r in APPENDIX_A])
def test_appendix_a(row):
    kw, fftsize, bm, bs, bitn, expect, en, fn, osc, buf, nc, ca = row
    for cfg in (M.rx_config(**kw), O.oracle_config(**kw)):
        assert cfg.fftsize == fftsize
        assert (cfg.b_mark, cfg.b_space) == (bm, bs)
        assert cfg.bit_nsamples == bitn
        assert cfg.expect_data.decode() == expect
        assert cfg.expect_nsamples == en
        assert cfg.frame_nsamples == fn
        assert cfg.nsamples_overscan == osc
        assert cfg.samplebuf_size == buf
        assert (cfg.try_max[0], cfg.try_step[0]) == nc
        assert (cfg.try_first[1], cfg.try_max[1], cfg.try_step[1], cfg.try_step_fine[1]) == ca


MODES = [
    dict(baudmode="1200"), dict(baudmode="300"), dict(baudmode="rtty"), dict(baudmode="tdd"),
    dict(baudmode="same"), dict(baudmode="callerid"), dict(baudmode="uic-train"),
    dict(baudmode="uic-ground"), dict(baudmode="V.21"), dict(baudmode="12000"),
    dict(baudmode="0.5"), dict(baudmode="110"), dict(baudmode="1000", sample_rate=44100),
    dict(baudmode="1200", n_data_bits=7), dict(baudmode="1200", inverted_freqs=1),
    dict(baudmode="300", nstopbits=2.0), dict(baudmode="1200", sync_byte=0x7E),
    dict(baudmode="1200", msb_first=1, invert_start_stop=1),
    dict(baudmode="1200", binary_raw_nbits=16), dict(baudmode="rtty", n_data_bits=8),
    dict(baudmode="1200", confidence_threshold=3.0, search_limit=1.0),
    dict(baudmode="2400", mark_f=2400, space_f=1200, band_width=100),
    dict(baudmode="1200", sample_rate=8000), dict(baudmode="1200", rx_one=1),
]


@pytest.mark.parametrize("kw", MODES, ids=[str(i) for i in range(len(MODES))])
def test_product_config_equals_oracle_config(kw):
    a = M.rx_config(**kw).as_dict()
    b = O.oracle_config(**kw).as_dict()
    assert a == b


def test_invalid_modes_rejected():
    for kw in (dict(baudmode="bogus"), dict(baudmode="1200", mark_f=30000.0),
               dict(baudmode="1200", binary_raw_nbits=70)):
        with pytest.raises(ValueError):
            M.rx_config(**kw)
        with pytest.raises(ValueError):
            O.oracle_config(**kw)


def _declared_functions():
    """every function name include/*.h declares (a declarator followed by `(` at the start of a
    declaration: `int  mifsk_x( ...`, `const char *mifsk_y( ...`, `fsk_plan *` newline `fsk_z(`)"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for h in ("fsk.h", "mifsk.h"):
        text = open(os.path.join(root, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b((?:mifsk|fsk)_\w+)\s*\(", text))
    return names - {"mifsk_h"}


def test_library_exports_every_declared_symbol():
    """The library loads and exports every symbol the two headers declare, and the binding's
    list is that list (no compute calls: this runs without a GPU)."""
    lib = _lib.load()
    declared = _declared_functions()
    assert len(declared) > 40
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.mifsk_abi_version() == 8


def test_scan_plans_cover_their_windows():
    """Shared-segment plans (host side, no GPU): every window of a long-window scan is tiled
    exactly by consecutive segments, every segment sits in exactly one lane of one pass, the
    packed copies say what the plain arrays say."""
    import numpy as np
    for mode, kw in (("rtty", {}), ("rtty", dict(sample_rate=44100)), ("50", {}), ("30", {}),
                     ("rtty", dict(sample_rate=96000))):
        cfg = M.rx_config(mode, **kw)
        B, nb = int(cfg.bit_nsamples), int(cfg.expect_n_bits)
        for kind in range(4):
            p = M.scan_plan(cfg, kind)
            assert p is not None, (mode, kw, kind)
            first = cfg.try_first[kind & 1]
            mx = cfg.try_max[kind & 1]
            step = (cfg.try_step_fine if kind & 2 else cfg.try_step)[kind & 1]
            U = (mx - first - 1) // step + 1
            D = min(U - 1, first // step)
            cands = [first] + [first + ((i + 1) // 2) * step if i & 1 else first - ((i + 1) // 2) * step
                               for i in range(1, 2 * D + 1)] + [first + (i - D) * step for i in range(2 * D + 1, U + D)]
            assert p["nwin"] == len(cands) * nb
            rel, ln = p["seg_rel"].astype(np.int64), p["seg_len"].astype(np.int64)
            assert np.all(np.diff(rel) >= ln[:-1])                     # position order, no overlap
            for w in range(p["nwin"]):
                a = cands[w // nb] + int(cfg.bit_offset[w % nb])
                f, c = int(p["win_first"][w]), int(p["win_count"][w])
                assert rel[f] == a and rel[f + c - 1] + ln[f + c - 1] == a + B
                assert np.all(rel[f:f + c - 1] + ln[f:f + c - 1] == rel[f + 1:f + c])
            slots = p["slot_seg"]
            used = slots[slots != 0xFFFF]
            assert sorted(used.tolist()) == list(range(p["nseg"]))     # each segment exactly once
            for ps in range(p["npass"]):
                mine = slots[64 * ps:64 * ps + 64]
                mine = mine[mine != 0xFFFF]
                assert ln[mine].max() == p["pass_len"][ps] and ln[mine].min() == p["pass_min"][ps]
            assert p["bound_c"] >= B + 100
        # the plan that holds the carrier-held coarse scan's windows AND the fine scan's (kind 4):
        # coarse windows first, each tiled exactly by its segments
        u = M.scan_plan(cfg, 4)
        c1, c3 = M.scan_plan(cfg, 1), M.scan_plan(cfg, 3)
        assert u is not None and u["nwin"] == c1["nwin"] + c3["nwin"]
        urel, uln = u["seg_rel"].astype(np.int64), u["seg_len"].astype(np.int64)
        for kind, off in ((1, 0), (3, c1["nwin"])):
            pk = M.scan_plan(cfg, kind)
            for w in range(pk["nwin"]):
                a = int(pk["seg_rel"][pk["win_first"][w]])              # the window's start
                f, c = int(u["win_first"][off + w]), int(u["win_count"][off + w])
                assert urel[f] == a and urel[f + c - 1] + uln[f + c - 1] == a + B
                assert np.all(urel[f:f + c - 1] + uln[f:f + c - 1] == urel[f + 1:f + c])
    assert M.scan_plan(M.rx_config("1200"), 1) is None                  # short windows: no plan


def test_struct_layouts_match_between_bindings():
    # the ctypes mirror against the library's own sizeof() of every public struct
    import minimodem_amd as M
    lib = _lib.load()
    mirror = {"mifsk_modem_args": _lib.ModemArgs, "mifsk_rx_config": _lib.RxConfig, "mifsk_search": _lib.Search,
              "mifsk_search_result": _lib.SearchResult, "mifsk_demod_io": _lib.DemodIO,
              "mifsk_launch_info": _lib.LaunchInfo, "mifsk_pipeline_info": _lib.PipelineInfo, "mifsk_gather_info": _lib.GatherInfo, "mifsk_session_result": _lib.SessionResult, "mifsk_scan_plan": _lib.ScanPlan,
              "mifsk_host_stats": _lib.HostStats, "mifsk_wav_info": _lib.WavInfo,
              "mifsk_file_result": _lib.FileResult, "fsk_plan": _lib.FskPlan}
    for name, t in mirror.items():
        assert lib.mifsk_abi_sizeof(name.encode()) == C.sizeof(t), name
    assert lib.mifsk_abi_sizeof(b"mifsk_frame") == M.FRAME_DTYPE.itemsize
    assert lib.mifsk_abi_sizeof(b"mifsk_episode") == M.EPISODE_DTYPE.itemsize
    assert lib.mifsk_abi_sizeof(b"mifsk_stream_state") == M.STATE_DTYPE.itemsize
    assert lib.mifsk_abi_sizeof(b"no such struct") == 0
    assert C.sizeof(_lib.RxConfig) == C.sizeof(O.RxConfig)
    assert C.sizeof(_lib.ModemArgs) == C.sizeof(O.ModemArgs)


def test_stream_padding_is_the_search_reach():
    """mifsk_stream_padding = last candidate position + last bit window, rounded to 4."""
    lib = M._lib.load()
    for mode in ("1200", "300", "rtty", "same", "12000"):
        cfg = M.rx_config(mode)
        reach = max(cfg.try_max[0], cfg.try_max[1])
        last = cfg.bit_offset[cfg.expect_n_bits - 1] + cfg.bit_nsamples
        assert lib.mifsk_stream_padding(C.byref(cfg)) == (reach + last + 3) & ~3


def test_gather_entry_checks_its_arguments_before_it_touches_a_device():
    """mifsk_gather_create (include/mifsk.h): -EINVAL for a rank outside the world, a world of
    several without the id, loopback outside a world of one -- decided before any HIP or RCCL
    call; with sound arguments the device decides (-ENODEV on a box without one)."""
    import torch
    lib = _lib.load()
    h = C.c_void_p()
    ident = (C.c_ubyte * _lib.GATHER_ID_BYTES)()
    assert lib.mifsk_gather_create(None, ident, 0, 1, -1, 2, 0) == -22
    assert lib.mifsk_gather_create(C.byref(h), ident, 2, 2, -1, 2, 0) == -22
    assert lib.mifsk_gather_create(C.byref(h), None, 0, 2, -1, 2, 0) == -22
    assert lib.mifsk_gather_create(C.byref(h), ident, 0, 2, -1, 2, _lib.GATHER_LOOPBACK) == -22
    assert lib.mifsk_gather_create(C.byref(h), ident, 0, 1, -1, 2, 2) == -22
    assert lib.mifsk_gather_unique_id(None) == -22
    assert lib.mifsk_gather_start(None, None, 0, None, 0, 0, None, None, None) == -22
    assert lib.mifsk_gather_received(None, 0, 0, None, None, None, None) == -22
    if not torch.cuda.is_available():
        assert lib.mifsk_gather_create(C.byref(h), None, 0, 1, -1, 2, 0) == -19 and not h.value
