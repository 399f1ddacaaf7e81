"""GPU parity tests for CRI ADX (BASELINE config 3 shape at reduced channel count): the HIP
path through the C ABI must be bit-exact with the CPU oracle for encode and decode."""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth
from vgaudio_amd.criadx import CriAdxCodec, CriAdxFormat, CriAdxParameters, CriAdxType
from vgaudio_amd.gcadpcm import Pcm16Format

pytestmark = pytest.mark.gpu

CASES = [dict(Type=3), dict(Type=4), dict(Type=2, Filter=0), dict(Type=2, Filter=1), dict(Type=2, Filter=2),
         dict(Type=2, Filter=3), dict(Type=3, Version=3), dict(Type=4, Version=3), dict(Type=3, FrameSize=34),
         dict(Type=3, Padding=70), dict(Type=4, Padding=33), dict(Type=2, Filter=2, Padding=64),
         dict(Type=3, SampleRate=22050),
         # padded (looping) streams through the time-piece kernels (round 6): a padding inside the first frame, of exactly
         # one frame, reaching into the second, a multiple of the load width and not
         dict(Type=3, Padding=1), dict(Type=3, Padding=8), dict(Type=3, Padding=31), dict(Type=4, Version=3, Padding=32),
         dict(Type=3, Version=3, Padding=24), dict(Type=2, Filter=1, Padding=63)]


def _op(kw):
    m = dict(Type="type", Filter="filter", Version="version", FrameSize="frame_size", Padding="padding",
             SampleRate="sample_rate", History="history", HighpassFrequency="highpass_frequency")
    return po.adx_params(**{m[k]: v for k, v in kw.items()})


def _edge_channels(n, rng):
    t = np.arange(n)
    return [synth.generate(1, n)[0], synth.sine(n), np.zeros(n, np.int16),
            np.where((t // 5) % 2 == 0, 32767, -32768).astype(np.int16),
            rng.integers(-32768, 32768, n).astype(np.int16), rng.integers(-4, 5, n).astype(np.int16),
            np.full(n, -32768, np.int16), np.where(t % 61 == 0, 32767, 0).astype(np.int16)]


@pytest.mark.parametrize("kw", CASES, ids=[str(c) for c in CASES])
@pytest.mark.parametrize("n", [32 * 300, 32 * 300 + 13])
def test_encode_decode_match_oracle(kw, n):
    rng = np.random.default_rng(n)
    chans = _edge_channels(n, rng)
    cfg = CriAdxParameters(**kw)
    enc = CriAdxCodec.Encode(chans, cfg)
    for i, pcm in enumerate(chans):
        p = _op(kw)
        want = po.adx_encode(pcm, p)
        assert len(enc[i]) == len(want) and (enc[i] == want).all(), (i, int(np.argmax(enc[i] != want)))
        assert int(np.atleast_1d(cfg.History)[i]) == p.history
    # decoder: CriAdxFormat.ToPcm16 starts from History 0 (CriAdxFormat.cs:39-48)
    dkw = {k: v for k, v in kw.items() if k != "Filter"}
    dec = CriAdxCodec.Decode(enc, n, CriAdxParameters(**dkw))
    for i in range(len(chans)):
        want = po.adx_decode(enc[i], n, _op(dkw))
        assert (dec[i] == want).all(), i


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65])
def test_ragged_lengths(n):
    rng = np.random.default_rng(n)
    pcm = rng.integers(-20000, 20000, (3, n)).astype(np.int16)
    for kw in (dict(Type=3), dict(Type=4, Version=3), dict(Type=2, Filter=3, Padding=5)):
        enc = CriAdxCodec.Encode(list(pcm), CriAdxParameters(**kw))
        for c in range(3):
            assert (enc[c] == po.adx_encode(pcm[c], _op(kw))).all()


def test_decode_random_bitstreams_and_history():
    rng = np.random.default_rng(9)
    n = 32 * 200 + 5
    frames = -(-n // 32)
    data = rng.integers(0, 256, (40, frames * 18)).astype(np.uint8)
    data[:, 0::18] &= 0x1F                       # filter 0 (the only one Linear/Exponential streams may name)
    for kw in (dict(Type=3, History=1234), dict(Type=3, Version=3, History=-77), dict(Type=4)):
        if kw["Type"] == 4:
            data[:, 0::18] = 0
            data[:, 1::18] %= 13
        dec = CriAdxCodec.Decode(list(data), n, CriAdxParameters(**kw))
        for c in range(40):
            assert (dec[c] == po.adx_decode(data[c], n, _op(kw))).all(), (kw, c)
    fixed = data.copy()
    fixed[:, 0::18] = rng.integers(0, 4, fixed[:, 0::18].shape).astype(np.uint8) << 5
    dec = CriAdxCodec.Decode(list(fixed), n, CriAdxParameters(Type=2))
    for c in range(40):
        assert (dec[c] == po.adx_decode(fixed[c], n, _op(dict(Type=2)))).all()


def test_errors():
    import vgaudio_amd
    with pytest.raises(vgaudio_amd.ArgumentError):
        CriAdxCodec.Encode(np.zeros(0, np.int16), CriAdxParameters())            # reads pcm[0]
    with pytest.raises(vgaudio_amd.ArgumentError):
        CriAdxCodec.Encode(np.zeros(32, np.int16), CriAdxParameters(Type=2, Filter=4))
    with pytest.raises(vgaudio_amd.ArgumentError):
        CriAdxCodec.Decode(np.zeros(17, np.uint8), 32)                           # stream too short
    bad = np.zeros(18, np.uint8)
    bad[0] = 0x20                                                                # filter 1 in a Linear stream
    with pytest.raises(vgaudio_amd.ArgumentError):
        CriAdxCodec.Decode(bad, 32)


def test_format_roundtrip_config3_shape_reduced():
    """BASELINE configs[2]: ADX encode+decode round trip, bit-exact check (128 channels x 10 s here)."""
    pcm = synth.generate(128, 480000)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    want, hist = po.adx_encode_batch(pcm, po.adx_params(), threads=8)
    for c in range(128):
        assert (fmt.Channels[c].Audio == want[c]).all() and fmt.Channels[c].History == hist[c]
    back = fmt.ToPcm16()
    wdec = po.adx_decode_batch(want, 480000, po.adx_params(), threads=8)
    for c in range(128):
        assert (back.Channels[c] == wdec[c]).all()
    err = np.stack(back.Channels).astype(float) - pcm
    assert np.sqrt((err ** 2).mean()) < 0.12 * np.sqrt((pcm.astype(float) ** 2).mean())   # 4-bit ADPCM, fixed high-pass predictor


def test_encoder_pieces_and_seams_on_random_shapes():
    """The 18-byte-frame encoder cuts channels into time pieces and hands the seams between them out a lane at a time
    (adx_encode_fs18_fixup_kernel, LABNOTES.md 8.7): random channel counts (the queue's refills start and end anywhere),
    lengths with and without a partial last frame, piece counts from 2 to the shortest the hook allows (64 frames), the
    encodings the kernel is instantiated for, seams forced open in none / all / a pattern of the places.  Bytes and the
    returned history are the oracle's."""
    import ctypes as C

    import torch

    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20260927)
    cases = []
    for _ in range(22):
        cases.append((int(rng.integers(1, 200)), 32 * int(rng.integers(130, 900)) + int(rng.choice([0, 0, 1, 13, 31])),
                      int(rng.choice([2, 3, 5, 8, 13, 1000])), int(rng.choice([0, 0, 1, 2, 3])),
                      [dict(Type=3), dict(Type=4), dict(Type=3, Version=3), dict(Type=2, Filter=2), dict(Type=4, Version=3),
                       dict(Type=3, Padding=24), dict(Type=4, Padding=57), dict(Type=3, Version=3, Padding=33)][int(rng.integers(0, 8))]))
    for nch, n, pieces, mode, kw in cases:
        host = np.stack([synth.generate(1, n, first_channel=int(rng.integers(0, 4096)))[0] for _ in range(min(nch, 6))])
        host = np.concatenate([host, rng.integers(-32768, 32768, (nch - len(host), n)).astype(np.int16)]) if nch > len(host) else host
        pcm = vdev.alloc_pcm(nch, n, d)
        pcm[:, :n] = torch.from_numpy(host).to(d)
        cfg = CriAdxParameters(**kw)
        p = _lib.AdxParams()
        L.vga_adx_default_params(C.byref(p))
        p.type, p.version, p.filter, p.padding = cfg.Type, cfg.Version, cfg.Filter, cfg.Padding
        nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
        pitch = (nb + 15) // 16 * 16
        adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
        hist = torch.zeros(nch, dtype=torch.int16, device=d)
        L.vga_testing_gc_encoder_segments_this_thread(pieces)
        old = L.vga_testing_force_open_seams_this_thread(mode)
        try:
            _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch,
                                               hist.data_ptr(), st))
            torch.cuda.synchronize()
        finally:
            L.vga_testing_force_open_seams_this_thread(old)
            L.vga_testing_gc_encoder_segments_this_thread(0)
        want, whist = po.adx_encode_batch(host, _op(kw), threads=8)
        got = adx[:, :nb].cpu().numpy()
        assert np.array_equal(got, want), (nch, n, pieces, mode, kw, int(np.argmax((got != want).any(axis=1))))
        assert np.array_equal(hist.cpu().numpy(), whist), (nch, n, pieces, mode, kw)


def test_padded_streams_decode_in_pieces():
    """Padded (looping) streams through the time-piece decoder (round 6): random bitstreams, paddings inside the first frame /
    of one frame / reaching into the second, lengths whose last samples the reference leaves zero (it reads
    ceil(sampleCount / 32) frames from the one the padding ends in, CriAdxCodec.cs:18-34), piece counts from the hook,
    seams forced open in none / all / a pattern of the places."""
    import ctypes as C

    import torch

    from vgaudio_amd import _lib
    L = _lib.lib()
    d = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(20261001)
    for case in range(16):
        nch = int(rng.integers(1, 150))
        n = 32 * int(rng.integers(40, 400)) + int(rng.choice([0, 0, 1, 13, 31]))
        padding = int(rng.choice([1, 7, 8, 24, 31, 32, 33, 57, 63, 64]))
        pieces = int(rng.choice([0, 2, 3, 5, 8]))
        mode = int(rng.choice([0, 0, 1, 2]))
        kw = [dict(Type=3), dict(Type=4), dict(Type=3, Version=3), dict(Type=2)][int(rng.integers(0, 4))]
        p = _lib.AdxParams()
        L.vga_adx_default_params(C.byref(p))
        cfg = CriAdxParameters(Padding=padding, **kw)
        p.type, p.version, p.padding = cfg.Type, cfg.Version, padding
        nb = (padding // 32 + -(-n // 32)) * 18
        data = rng.integers(0, 256, (nch, nb)).astype(np.uint8)
        if cfg.Type == 4:
            data[:, 0::18] = 0
            data[:, 1::18] %= 13
        elif cfg.Type == 2:
            data[:, 0::18] = rng.integers(0, 4, data[:, 0::18].shape).astype(np.uint8) << 5
        else:
            data[:, 0::18] &= 0x1F
        pitch = (nb + 15) // 16 * 16
        adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
        adx[:, :nb] = torch.from_numpy(data).to(d)
        opitch = (n + 7) // 8 * 8
        out = torch.full((nch, opitch), 0x5A5A, dtype=torch.int16, device=d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        L.vga_testing_gc_encoder_segments_this_thread(pieces)
        old = L.vga_testing_force_open_seams_this_thread(mode)
        try:
            _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), out.data_ptr(), opitch, status.data_ptr(), st))
            torch.cuda.synchronize()
        finally:
            L.vga_testing_force_open_seams_this_thread(old)
            L.vga_testing_gc_encoder_segments_this_thread(0)
        got = out[:, :n].cpu().numpy()
        okw = dict(kw)
        okw["Padding"] = padding
        for c in range(nch):
            want = po.adx_decode(data[c], n, _op(okw))
            assert np.array_equal(got[c], want), (case, nch, n, padding, pieces, mode, kw, c, int(np.argmax(got[c] != want)))
