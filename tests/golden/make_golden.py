#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref).

Run in the build container (needs oracle/_ref, i.e. /root/reference present):

    python tests/golden/make_golden.py

For every case (one per family of reference tests, tests/*.test) it
  1. transmits a short payload with `minimodem_ref --tx --file x.wav ...`
  2. receives it with `minimodem_ref --rx --file x.wav ...`
  3. drives the reference's own fsk_find_frame() (libfsk_ref.so = unmodified
     src/fsk.c + FFT shim) over a deterministic set of search windows
and stores: the audio samples (exactly as written by the reference TX), the
decoded stdout bytes, the "### NOCARRIER" statistics lines, and the
find_frame call trace (arguments + outputs).  The oracle restatement and the
HIP path are both checked against these files; the files are small on purpose
(short payloads) -- the full 500-byte reference payloads are exercised live
against oracle/_ref by tests/test_oracle_vs_reference.py where it exists.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O  # noqa: E402

ASCII_PAYLOAD = (b"The quick brown fox jumps over the lazy dog 0123456789 "
                 b"!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~ MI355X fsk\n")
BAUDOT_PAYLOAD = b"THE QUICK BROWN FOX JUMPS OVER THE LAZY DOG 0123456789\n"



def _ref_bytes(name):
    path = os.path.join("/root/reference/tests", name)
    if not os.path.exists(path):
        return b""
    with open(path, "rb") as f:
        return f.read()


# name, payload, tx args, rx args, oracle_config kwargs
CASES = [
    ("t01_1200", ASCII_PAYLOAD, ["1200"], ["1200"], dict(baudmode="1200")),
    ("t02_300", ASCII_PAYLOAD[:48], ["300"], ["300"], dict(baudmode="300")),
    ("t03_rtty", BAUDOT_PAYLOAD[:24], ["rtty"], ["rtty"], dict(baudmode="rtty")),
    ("t05_12000", ASCII_PAYLOAD, ["12000"], ["12000"], dict(baudmode="12000")),
    ("t06_1200_float", ASCII_PAYLOAD, ["1200", "--float-samples"], ["1200"],
     dict(baudmode="1200")),
    ("t07_1200_nolut", ASCII_PAYLOAD, ["1200", "--lut=0"], ["1200"], dict(baudmode="1200")),
    ("t08_1200_lut16", ASCII_PAYLOAD, ["1200", "--lut=16"], ["1200"], dict(baudmode="1200")),
    ("t10_perfect", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    ("t13_perfect_nolut_float", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400 --lut=0 --float-samples".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    ("t14_perfect_lut16_float", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400 --lut=16 --float-samples".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    ("t21_slop_308", ASCII_PAYLOAD[:48], ["308"], ["300"], dict(baudmode="300")),
    ("t21_slop_292", ASCII_PAYLOAD[:48], ["292"], ["300"], dict(baudmode="300")),
    ("t30_ampl_0p3", ASCII_PAYLOAD, ["--volume", "0.30", "1200"], ["1200"],
     dict(baudmode="1200")),
    ("t60_7bit", ASCII_PAYLOAD, ["1200", "-7"], ["1200", "-7"],
     dict(baudmode="1200", n_data_bits=7)),
    ("t80_same", ASCII_PAYLOAD, ["same"], ["same"], dict(baudmode="same")),
    ("t81_tdd", BAUDOT_PAYLOAD[:24], ["tdd"], ["tdd"], dict(baudmode="tdd")),
    # tests/70-callerid-mdmf.test, 71-callerid-sdmf.test: the reference's own message bytes
    ("t70_callerid_mdmf", _ref_bytes("testdata-callerid-mdmf.bytes"), ["1200", "--ascii"],
     ["callerid"], dict(baudmode="callerid")),
    ("t71_callerid_sdmf", _ref_bytes("testdata-callerid-sdmf.bytes"), ["1200", "--ascii"],
     ["callerid"], dict(baudmode="callerid")),
    # output modes of the databits post-pass
    ("t90_binary_output", ASCII_PAYLOAD[:40], ["1200"], ["1200", "--binary-output"],
     dict(baudmode="1200", binary_output=1)),
    # --auto-carrier (minimodem.c:1179-1220, fsk.c:543-598): standard tones, moved tones behind
    # leading silence that is not a whole number of scan windows, inverted tones, long windows
    ("t50_auto_300", ASCII_PAYLOAD[:32], ["300"], ["--auto-carrier", "300"],
     dict(baudmode="300", auto_carrier_threshold=0.001)),
    ("t50_auto_300_moved_lead", ASCII_PAYLOAD[:32], ["300", "-M", "1570", "-S", "1370"],
     ["--auto-carrier", "300"], dict(baudmode="300", auto_carrier_threshold=0.001), dict(lead=10007)),
    ("t50_auto_1200_lead", ASCII_PAYLOAD[:40], ["1200"], ["--auto-carrier", "1200"],
     dict(baudmode="1200", auto_carrier_threshold=0.001), dict(lead=30001)),
    ("t50_auto_300_inverted", ASCII_PAYLOAD[:32], ["300", "-M", "1070", "-S", "1270"],
     ["--auto-carrier", "-i", "300"],
     dict(baudmode="300", auto_carrier_threshold=0.001, inverted_freqs=1), dict(lead=333)),
    ("t50_auto_rtty_lead", BAUDOT_PAYLOAD[:12], ["rtty"], ["--auto-carrier", "rtty"],
     dict(baudmode="rtty", auto_carrier_threshold=0.001), dict(lead=5000)),
    # tests/40-noise.test: --Xrxnoise adds the constant -factor (integer division of rand())
    ("t40_rxnoise_0p10", ASCII_PAYLOAD[:40], ["--volume", "0.5", "1200"],
     ["--Xrxnoise", "0.10", "1200"], dict(baudmode="1200")),
    ("t40_rxnoise_0p50_float", ASCII_PAYLOAD[:40], ["--volume", "0.5", "1200", "--float-samples"],
     ["--Xrxnoise", "0.50", "1200"], dict(baudmode="1200")),
    ("t91_print_filter", b"tab\there \x01\x02 bell\x07 del\x7f high\xe9\xff nl\n cr\r end",
     ["1200"], ["1200", "--print-filter"], dict(baudmode="1200")),
    # ---- round 2: the reference tests that had no recording of their own --------------
    # tests/04-self-test-0.5.test: 96000-sample bit windows, 6.1 M samples.  Stored as payload +
    # SHA-256 of the reference's samples; the loader regenerates them with the host transmitter
    # (bit-exact with the reference's WAV: tests/test_tx_synth.py) and checks the hash.
    ("t04_0p5", b"KAMAL\n", ["0.5"], ["0.5"], dict(baudmode="0.5"), dict(regen=True)),
    ("t09_1200_lut16_float", ASCII_PAYLOAD, ["1200", "--lut=16", "--float-samples"], ["1200"],
     dict(baudmode="1200")),
    ("t11_perfect_nolut", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400 --lut=0".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    ("t12_perfect_lut16", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400 --lut=16".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    ("t15_perfect_float", ASCII_PAYLOAD,
     "1200 --samplerate 24000 -M 1200 -S 2400 --float-samples".split(),
     "1200 --samplerate 24000 -M 1200 -S 2400".split(),
     dict(baudmode="1200", sample_rate=24000, mark_f=1200, space_f=2400)),
    # tests/30-amplitude.test / 31-amplitude-float.test: clipping S16 at volume 3.5, unclipped
    # float at 3.5, the 0.01 floor
    ("t30_ampl_3p50", ASCII_PAYLOAD[:48], ["--volume", "3.50", "1200"], ["1200"], dict(baudmode="1200")),
    ("t30_ampl_0p01", ASCII_PAYLOAD[:48], ["--volume", "0.01", "1200"], ["1200"], dict(baudmode="1200")),
    ("t31_ampl_float_3p50", ASCII_PAYLOAD[:48], ["--volume", "3.50", "1200", "--float-samples"],
     ["1200"], dict(baudmode="1200")),
    ("t31_ampl_float_0p01", ASCII_PAYLOAD[:48], ["--volume", "0.01", "1200", "--float-samples"],
     ["1200"], dict(baudmode="1200")),
    # tests/40-noise.test and 41-noise-purefreqs.test as they are run: --volume 0.5, --rx-one
    ("t40_rxnoise_0p05_rxone", ASCII_PAYLOAD[:40], ["1200", "--volume", "0.5"],
     ["1200", "--Xrxnoise", "0.05", "--rx-one"], dict(baudmode="1200", rx_one=1)),
    ("t41_purefreqs_0p50_rxone", ASCII_PAYLOAD[:40], ["1200", "-M", "1200", "-S", "2400", "--volume", "0.5"],
     ["1200", "-M", "1200", "-S", "2400", "--Xrxnoise", "0.50", "--rx-one"],
     dict(baudmode="1200", mark_f=1200, space_f=2400, rx_one=1)),
    ("t41_purefreqs_0p00_rxone", ASCII_PAYLOAD[:40], ["1200", "-M", "1200", "-S", "2400", "--volume", "0.5"],
     ["1200", "-M", "1200", "-S", "2400", "--Xrxnoise", "0.00", "--rx-one"],
     dict(baudmode="1200", mark_f=1200, space_f=2400, rx_one=1)),
    # --auto-carrier re-detection (minimodem.c:1297: the band is dropped after 21 searches without
    # confidence): noise above the detection threshold before the signal -- the reference locks onto
    # a noise band first and has to recover --, and two bursts on different tone pairs
    ("t51_auto_300_noise_lead", ASCII_PAYLOAD[:24], None, ["--auto-carrier", "300"],
     dict(baudmode="300", auto_carrier_threshold=0.001),
     dict(compose=[("noise", 9000, 0.02, 7), ("tx", ["300"], ASCII_PAYLOAD[:24]), ("silence", 3000)])),
    ("t51_auto_300_two_bursts", ASCII_PAYLOAD[:24] + ASCII_PAYLOAD[24:40], None, ["--auto-carrier", "300"],
     dict(baudmode="300", auto_carrier_threshold=0.001),
     dict(compose=[("silence", 2000), ("tx", ["300"], ASCII_PAYLOAD[:24]), ("silence", 20011),
                   ("tx", ["300", "-M", "1870", "-S", "1670"], ASCII_PAYLOAD[24:40]), ("silence", 5000)])),
    ("t51_auto_1200_two_bursts_noise", ASCII_PAYLOAD[:30] + ASCII_PAYLOAD[30:60], None, ["--auto-carrier", "1200"],
     dict(baudmode="1200", auto_carrier_threshold=0.001),
     dict(compose=[("noise", 4000, 0.004, 3), ("tx", ["1200"], ASCII_PAYLOAD[:30]), ("noise", 9000, 0.004, 4),
                   ("tx", ["1200", "-M", "1600", "-S", "2600"], ASCII_PAYLOAD[30:60]), ("silence", 2500)])),
]


def find_frame_trace(cfg, x, max_calls=24):
    """Call the reference's fsk_find_frame on windows laid out the way main()
    lays them out (coarse no-carrier, coarse carrier, fine), at deterministic
    offsets spread over the stream."""
    lib = O.ref_lib()
    plan = lib.fsk_plan_new(float(cfg.sample_rate), cfg.mark_f, cfg.space_f, cfg.band_width)
    assert plan
    pad = np.zeros(int(cfg.expect_nsamples) + 2 * int(cfg.try_max[0]) + 64, np.float32)
    xp = np.concatenate([x, pad])
    n = x.shape[0]
    span = int(cfg.expect_nsamples) + int(cfg.try_max[0])
    rng = np.random.default_rng(20260923)
    offs = sorted(set(int(v) for v in rng.integers(0, max(1, n - span), size=max_calls)))
    rows = []
    for i, off in enumerate(offs):
        mode = i % 3
        if mode == 0:      # coarse, no carrier, sync string
            ci, step, limit, expect = 0, cfg.try_step[0], cfg.search_limit, cfg.expect_sync
        elif mode == 1:    # coarse, carrier, data string
            ci, step, limit, expect = 1, cfg.try_step[1], cfg.search_limit, cfg.expect_data
        else:              # fine rescan
            ci, step, limit, expect = 1, cfg.try_step_fine[1], float("inf"), cfg.expect_data
        first, tmax = int(cfg.try_first[ci]), int(cfg.try_max[ci])
        conf, bits, ampl, start = O.ref_find_frame(
            plan, xp[off:], int(cfg.expect_nsamples), first, tmax, int(step),
            limit, expect)
        rows.append((off, first, tmax, int(step), limit, 1 if expect == cfg.expect_sync and
                     cfg.expect_sync != cfg.expect_data else 0, conf, bits, ampl, start))
    lib.fsk_plan_destroy(plan)
    dt = np.dtype([("offset", "<u8"), ("first", "<u4"), ("max", "<u4"), ("step", "<u4"),
                   ("limit", "<f4"), ("use_sync", "<u4"), ("confidence", "<f4"),
                   ("bits", "<u8"), ("amplitude", "<f4"), ("start", "<u4")])
    return np.array(rows, dtype=dt)


def compose_wav(parts, wav):
    """A recording made of reference transmissions, silence and (seeded) noise, written as S16
    the way the reference writes its files."""
    chunks, sr = [], None
    for part in parts:
        if part[0] == "tx":
            tmp = O.tmp_wav()
            try:
                O.ref_tx(part[2], part[1], tmp)
                with open(tmp, "rb") as f:
                    raw = f.read()
                sr0, _ = O.read_wav(tmp)
            finally:
                os.unlink(tmp)
            assert sr in (None, sr0)
            sr = sr0
            chunks.append(np.frombuffer(raw[44:], dtype="<i2").copy())
        elif part[0] == "silence":
            chunks.append(np.zeros(part[1], "<i2"))
        elif part[0] == "noise":
            rng = np.random.default_rng(part[3])
            chunks.append(np.clip(np.rint(rng.normal(0, part[2], part[1]) * 32768), -32768, 32767).astype("<i2"))
        else:
            raise ValueError(part[0])
    O.write_wav(wav, np.concatenate(chunks), sr, True)


def main():
    import hashlib
    only = set(sys.argv[1:])
    assert O.have_ref(), "oracle/_ref missing: run `make -C oracle` where /root/reference exists"
    for case in CASES:
        name, payload, tx, rx, kw = case[:5]
        if only and name not in only:
            continue
        extra = case[5] if len(case) > 5 else {}
        wav = O.tmp_wav()
        try:
            if extra.get("compose"):
                compose_wav(extra["compose"], wav)
                tx = ["(composed)"]
            else:
                O.ref_tx(payload, tx, wav)
            if extra.get("lead"):       # leading silence: rewrite the reference's file
                with open(wav, "rb") as f:
                    raw0 = f.read()
                sr0, x0 = O.read_wav(wav)
                if "--float-samples" in tx:
                    body = np.concatenate([np.zeros(extra["lead"], "<f4"), x0.astype("<f4")])
                else:
                    body = np.concatenate([np.zeros(extra["lead"], "<i2"),
                                           np.frombuffer(raw0[44:], dtype="<i2")])
                O.write_wav(wav, body, sr0, "--float-samples" not in tx)
            out, err = O.ref_rx(wav, rx)
            with open(wav, "rb") as f:
                raw = f.read()
            sr, x = O.read_wav(wav)
        finally:
            os.unlink(wav)
        is_float = "--float-samples" in tx
        # store the samples in their on-disk encoding (S16 files stay 2 B/sample)
        if is_float:
            stored = x.astype("<f4")
        else:
            stored = np.frombuffer(raw[44:], dtype="<i2").copy()
            assert np.array_equal(stored.astype(np.float32) / np.float32(32768.0), x)
        cfg = O.oracle_config(**kw)
        assert cfg.sample_rate == sr
        x_rx = x
        if "--Xrxnoise" in rx:      # what the receive loop sees (simpleaudio-sndfile.c:64-69)
            f = np.float32(float(rx[rx.index("--Xrxnoise") + 1]))
            x_rx = x + (np.float32(0) - np.float32(0.5)) * (f * np.float32(2))
        trace = find_frame_trace(cfg, x_rx)
        stats = [l for l in err.splitlines() if l.startswith("### NOCARRIER")]
        carriers = [l for l in err.splitlines() if l.startswith("### CARRIER")]
        path = os.path.join(HERE, name + ".npz")
        regen = {}
        if extra.get("regen"):
            # too long to store: keep the hash and the transmitter's arguments instead
            import minimodem_amd as M
            mine = M.synthesize(M.rx_config(**kw), payload, s16=not is_float)
            assert np.array_equal(mine, x), "host transmitter differs from the reference's WAV"
            regen = dict(samples_sha256=np.array(hashlib.sha256(stored.tobytes()).hexdigest()),
                         samples_len=np.int64(stored.shape[0]),
                         samples_dtype=np.array(stored.dtype.str))
            trace = trace[:6]
            stored = stored[:0]
        np.savez_compressed(
            path, samples=stored, sample_rate=np.int64(sr), **regen,
            payload=np.frombuffer(payload, np.uint8),
            stdout=np.frombuffer(out, np.uint8),
            nocarrier=np.array(stats), carrier=np.array(carriers),
            trace=trace,
            tx_args=np.array(tx), rx_args=np.array(rx),
            cfg_keys=np.array(list(kw.keys())),
            cfg_vals=np.array([str(v) for v in kw.values()]))
        print("%-28s %8d samples  %4d bytes out  %s  (%d KiB)" % (
            name, x.shape[0], len(out), stats[-1] if stats else "-",
            os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
