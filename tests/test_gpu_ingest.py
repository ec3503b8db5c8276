"""Input side: S16 -> float conversion and the --Xrxnoise term on the
device (mifsk_ingest_s16 / mifsk_ingest_rxnoise_f32), checked bit-for-bit against the
host conversion the goldens were recorded with (libsndfile semantics: value / 32768),
and end to end: on-disk samples -> device ingest -> demod -> post-pass == what the
reference printed for the same file and options."""
import numpy as np
import pytest

import _golden as G
import minimodem_amd as M


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU: the ingest kernels have no CPU fallback")
    return torch, M.Context(0)


def _ingest(torch, ctx, g):
    s = g["stored"]
    if s.dtype == np.int16:
        pcm = torch.from_numpy(np.ascontiguousarray(s)[None, :]).cuda()
        return M.ingest_s16(ctx, pcm, rxnoise=g["rxnoise"])
    x = np.zeros((1, (s.shape[0] + 3) & ~3), np.float32)
    x[0, :s.shape[0]] = s
    d = torch.from_numpy(x).cuda()
    n = torch.tensor([s.shape[0]], dtype=torch.int32).cuda()
    return M.ingest_rxnoise(ctx, d, g["rxnoise"], nsamples=n) if g["rxnoise"] else d


@pytest.mark.gpu
@pytest.mark.parametrize("name", G.names())
def test_ingest_then_demod_reproduces_reference_output(gpu, name):
    torch, ctx = gpu
    g = G.load(name)
    n = g["samples"].shape[0]
    d = _ingest(torch, ctx, g)
    got = d.cpu().numpy()[0]
    assert got[:n].tobytes() == g["samples"].tobytes()          # bit-exact conversion
    assert not got[n:].any()                                    # row padding is zero
    cfg = M.rx_config(**g["cfg_kwargs"])
    nv = torch.tensor([n], dtype=torch.int32).cuda()
    res = M.results_to_host(M.demod_batch(ctx, cfg, d, nsamples=nv, want=("bits", "episodes")))
    out, err = M.stream_text(cfg, res["bits"][0, :int(res["nframes"][0])],
                             res["episodes"][0, :int(res["nepisodes"][0])],
                             print_filter="--print-filter" in g["rx_args"])
    assert out == g["stdout"]
    assert [l for l in err.splitlines() if l.startswith("### NOCARRIER")] == g["nocarrier"]


@pytest.mark.gpu
def test_ingest_s16_batch_ragged_unaligned(gpu):
    """Ragged lengths, a width that is not a multiple of 8, extreme values."""
    torch, ctx = gpu
    rng = np.random.default_rng(7)
    for width in (4096, 4099, 17):
        pcm = rng.integers(-32768, 32768, size=(5, width), dtype=np.int16)
        pcm[0, :4] = (-32768, 32767, 0, -1)
        lens = np.array([width, width - 1, width // 2, 1, 0], dtype=np.int32)
        d = M.ingest_s16(ctx, torch.from_numpy(pcm).cuda(), nsamples=torch.from_numpy(lens).cuda(),
                         rxnoise=0.05)
        got = d.cpu().numpy()
        term = (np.float32(0) - np.float32(0.5)) * (np.float32(0.05) * np.float32(2))
        for s in range(5):
            want = pcm[s, :lens[s]].astype(np.float32) / np.float32(32768.0) + term
            assert got[s, :lens[s]].tobytes() == want.tobytes()
            assert not got[s, lens[s]:].any()


def test_wav_parse_accepts_what_the_reference_writes():
    """CPU-only part: header parsing of PCM16 / float32 mono files."""
    import struct
    def wav(fmt_tag, bits, ch, frames):
        data = b"\x00" * (frames * ch * bits // 8)
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, 48000, 48000 * ch * bits // 8, ch * bits // 8, bits)
        return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<I", 16)
                + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + b"data"
                + struct.pack("<I", len(data)) + data)
    i = M.wav_parse(wav(1, 16, 1, 100))
    assert (i["sample_rate"], i["is_float"], i["nframes"], i["data_offset"]) == (48000, 0, 100, 56)
    i = M.wav_parse(wav(3, 32, 1, 7))
    assert (i["is_float"], i["bits_per_sample"], i["nframes"]) == (1, 32, 7)
    for bad in (wav(1, 16, 2, 4), wav(1, 24, 1, 4), b"RIFFxxxxWAVE", b"not a wav file at all"):
        with pytest.raises(ValueError):
            M.wav_parse(bad)
    # a file cut short yields the frames that are there
    assert M.wav_parse(wav(1, 16, 1, 100)[:-50])["nframes"] == 75
