"""The host-pointer entry points run a pipeline (feeders -> pinned rings -> per-chunk kernels -> drainers,
vgaudio_amd/csrc/host_pipeline.hpp).  Its shape normally follows the volume of the call; here the test hook forces
many workers, tiny ring slots and small chunks on small inputs, and the results must equal the default shape's and
the oracle's, byte for byte."""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.criadx import CriAdxCodec, CriAdxParameters
from vgaudio_amd.crihca import CriHcaDecoder, CriHcaFormat, CriHcaParameters
from vgaudio_amd.gcadpcm import GcAdpcmDecoder, GcAdpcmFormat, GcAdpcmParameters, Pcm16Format

pytestmark = pytest.mark.gpu

SHAPES = [(0, 0, 0, 0), (5, 3, 7, 4096), (8, 4, 1, 1), (1, 1, 1000, 1 << 20), (3, 8, 13, 30000), (3, 2, 5, -4096), (1, 1, 0, -1)]


@pytest.fixture(params=SHAPES, ids=[str(s) for s in SHAPES])
def shape(request):
    L = _lib.lib()
    L.vga_testing_host_pipeline_this_thread(*request.param)
    yield request.param
    L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)


def test_gc_encode_and_decode_through_every_pipeline_shape(shape):
    nch, n = 37, 14 * 900 + 5
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    wc, wa = po.gc_encode_batch(pcm, threads=4)
    for c in range(nch):
        assert fmt.Channels[c].Coefs.tolist() == np.asarray(wc).reshape(nch, 16)[c].tolist(), c
        assert np.array_equal(fmt.Channels[c].GetAdpcmAudio(), np.asarray(wa)[c][:len(fmt.Channels[c].GetAdpcmAudio())]), c
    dec = GcAdpcmDecoder.Decode([ch.GetAdpcmAudio() for ch in fmt.Channels], np.stack([ch.Coefs for ch in fmt.Channels]),
                                GcAdpcmParameters(SampleCount=n))
    want = po.gc_decode_batch(np.stack([ch.GetAdpcmAudio() for ch in fmt.Channels]), np.asarray(wc).reshape(nch, 16), n, threads=4)
    assert np.array_equal(np.stack(dec), want)


def test_adx_through_every_pipeline_shape(shape):
    nch, n = 29, 32 * 300 + 13
    pcm = synth.generate(nch, n)
    cfg = CriAdxParameters()
    enc = CriAdxCodec.Encode(list(pcm), cfg)
    want, hist = po.adx_encode_batch(pcm, po.adx_params(), threads=4)
    assert np.array_equal(np.stack(enc), want) and np.array_equal(np.asarray(cfg.History), hist)
    dec = CriAdxCodec.Decode(enc, n, CriAdxParameters())
    assert np.array_equal(np.stack(dec), po.adx_decode_batch(want, n, po.adx_params(), threads=4))


def test_hca_through_every_pipeline_shape(shape):
    ns, n = 11, 6000
    streams = [synth.generate(2, n, first_channel=2 * s) for s in range(ns)]
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(s), 48000) for s in streams], CriHcaParameters())
    rc, info, want = po.hca_encode_batch(np.stack(streams), po.hca_params(2, n), threads=4)
    assert rc == 0
    for s in range(ns):
        assert np.array_equal(fmts[s].AudioData.reshape(-1), want[s]), s
    dec = CriHcaDecoder.Decode(fmts[0].Hca, [f.AudioData for f in fmts])
    rc, wdec = po.hca_decode_batch(info, want, threads=4)
    assert rc == 0
    for s in range(ns):
        assert np.array_equal(np.stack(dec[s]), wdec[s]), s


@pytest.fixture(params=[[0, 0], [0, 0, 0], "all"], ids=["twice", "thrice", "all-devices"])
def devices(request):
    """vga_set_devices: one process, several GPUs (a device may be listed more than once -- which is how a one-GPU box
    exercises the shares)"""
    import ctypes as C
    L = _lib.lib()
    lst = list(range(L.vga_device_count())) if request.param == "all" else request.param
    _lib.check(L.vga_set_devices((C.c_int * len(lst))(*lst), len(lst)))
    yield lst
    _lib.check(L.vga_set_devices(None, 0))


def test_calls_spread_over_listed_devices_give_the_same_bytes(devices):
    """every *_batch entry point with its units cut into shares (>= 128 channels / 32 streams each), against the oracle"""
    nch, n = 128 * len(devices) + 41, 14 * 700 + 9
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    wc, wa = po.gc_encode_batch(pcm, threads=8)
    wc = np.asarray(wc).reshape(nch, 16)
    got_coefs = np.stack([ch.Coefs for ch in fmt.Channels])
    got_adpcm = np.stack([ch.GetAdpcmAudio() for ch in fmt.Channels])
    assert np.array_equal(got_coefs, wc) and np.array_equal(got_adpcm, np.asarray(wa)[:, :got_adpcm.shape[1]])
    dec = GcAdpcmDecoder.Decode(list(got_adpcm), got_coefs, GcAdpcmParameters(SampleCount=n))
    assert np.array_equal(np.stack(dec), po.gc_decode_batch(got_adpcm, wc, n, threads=8))
    # ADX
    cfg = CriAdxParameters()
    enc = CriAdxCodec.Encode(list(pcm), cfg)
    want, hist = po.adx_encode_batch(pcm, po.adx_params(), threads=8)
    assert np.array_equal(np.stack(enc), want) and np.array_equal(np.asarray(cfg.History), hist)
    assert np.array_equal(np.stack(CriAdxCodec.Decode(enc, n, CriAdxParameters())), po.adx_decode_batch(want, n, po.adx_params(), threads=8))
    # HCA: streams are the units
    ns, hn = 32 * len(devices) + 5, 4000
    streams = [synth.generate(2, hn, first_channel=2 * s) for s in range(ns)]
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(s), 48000) for s in streams], CriHcaParameters())
    rc, info, hwant = po.hca_encode_batch(np.stack(streams), po.hca_params(2, hn), threads=8)
    assert rc == 0
    for s in range(ns):
        assert np.array_equal(fmts[s].AudioData.reshape(-1), hwant[s]), s
    hdec = CriHcaDecoder.Decode(fmts[0].Hca, [f.AudioData for f in fmts])
    rc, wdec = po.hca_decode_batch(info, hwant, threads=8)
    for s in range(ns):
        assert np.array_equal(np.stack(hdec[s]), wdec[s]), s


def test_a_failing_share_fails_the_call(devices):
    import vgaudio_amd
    nch, n = 128 * len(devices), 14 * 50
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    adpcm = [ch.GetAdpcmAudio().copy() for ch in fmt.Channels]
    adpcm[nch - 3][8] = 0xF0                                  # predictor 15 in the LAST share: IndexOutOfRange in the reference
    with pytest.raises(vgaudio_amd.ArgumentError):
        GcAdpcmDecoder.Decode(adpcm, np.stack([ch.Coefs for ch in fmt.Channels]), GcAdpcmParameters(SampleCount=n))


class _Progress:
    """IProgressReport (VGAudio/IProgressReport.cs:3-28) as a recorder"""

    def __init__(self):
        self.total, self.adds = None, []

    def SetTotal(self, total):
        self.total = total

    def ReportAdd(self, value):
        self.adds.append(value)


def test_progress_is_reported_chunk_by_chunk(shape):
    """vga_set_progress_callback through the host mirrors: SetTotal(frames x channels) as the reference (GcAdpcmFormat.cs:63,
    CriAdxFormat.cs:65, CriHcaFormat.cs:48), ReportAdd per finished chunk, adding up to the total -- and the bytes do not
    care whether anybody listens"""
    chunk = shape[2]
    nch, n = 37, 14 * 300 + 5
    pcm = synth.generate(nch, n)
    quiet = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    pg = _Progress()
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000), GcAdpcmParameters(Progress=pg))
    frames = -(-n // 14)
    assert pg.total == frames * nch and sum(pg.adds) == frames * nch and all(a > 0 and a % frames == 0 for a in pg.adds)
    if 0 < chunk < nch:
        assert len(pg.adds) >= -(-nch // chunk)                 # one report per chunk (the last chunk may be split)
    for a, b in zip(fmt.Channels, quiet.Channels):
        assert np.array_equal(a.GetAdpcmAudio(), b.GetAdpcmAudio())
    # ADX
    pa = _Progress()
    cfg = CriAdxParameters(Progress=pa)
    enc = CriAdxCodec.Encode(list(pcm), cfg)
    per = len(enc[0]) // cfg.FrameSize
    assert sum(pa.adds) == per * nch and all(a % per == 0 for a in pa.adds)
    assert np.array_equal(np.stack(enc), np.stack(CriAdxCodec.Encode(list(pcm), CriAdxParameters())))
    # HCA: streams are the units
    ns, hn = 9, 3000
    streams = [Pcm16Format(list(synth.generate(2, hn, first_channel=2 * s)), 48000) for s in range(ns)]
    ph = _Progress()
    fmts = CriHcaFormat.EncodeBatchFromPcm16(streams, CriHcaParameters(Progress=ph))
    fc = fmts[0].Hca.FrameCount
    assert ph.total == fc * ns and sum(ph.adds) == fc * ns and all(a % fc == 0 for a in ph.adds)
    # a callback that was removed stays removed
    L = _lib.lib()
    calls = []
    fn = _lib.PROGRESS_FN(lambda u, d, t: calls.append(d))
    import ctypes as C
    L.vga_set_progress_callback(C.cast(fn, C.c_void_p), None)
    L.vga_set_progress_callback(None, None)
    CriAdxCodec.Encode(list(pcm[:3]), CriAdxParameters())
    assert calls == []


def test_progress_of_a_call_spread_over_devices_is_one_count(devices):
    """the shares of all listed devices report into the same (done, total), one report at a time"""
    import ctypes as C
    L = _lib.lib()
    nch, n = 128 * len(devices) + 41, 14 * 200
    pcm = synth.generate(nch, n)
    seen = []
    fn = _lib.PROGRESS_FN(lambda user, done, total: seen.append((done, total)))
    L.vga_testing_host_pipeline_this_thread(2, 2, 40, 4096)
    L.vga_set_progress_callback(C.cast(fn, C.c_void_p), None)
    try:
        enc = CriAdxCodec.Encode(list(pcm), CriAdxParameters())
    finally:
        L.vga_set_progress_callback(None, None)
        L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)
    assert seen and seen[-1] == (nch, nch) and all(t == nch for _, t in seen)
    assert [d for d, _ in seen] == sorted(d for d, _ in seen) and len(set(d for d, _ in seen)) == len(seen)
    assert len(seen) >= len(devices) * 3
    want, _ = po.adx_encode_batch(pcm, po.adx_params(), threads=8)
    assert np.array_equal(np.stack(enc), want)
