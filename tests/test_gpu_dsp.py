"""GPU parity for the DSP container writer: the file image assembled on the device
(vga_dsp_write / vga_dsp_write_device) must equal the oracle's byte for byte, and parse back to the audio that went
in -- the reference's own DspBuildAndParseEqual (VGAudio.Tests/Containers/DspTests.cs:9-18)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.dsp import DspConfiguration, DspWriter
from vgaudio_amd.gcadpcm import GcAdpcmFormat, Pcm16Format

pytestmark = pytest.mark.gpu

FREQS = [261.63, 329.63, 392, 523.25, 659.25, 783.99, 1046.50, 130.81]      # GenerateAudio.cs:14


def sine(n, f, rate):
    i = np.arange(n, dtype=np.float64)
    return np.trunc(32767 * np.sin(2 * np.pi * f / rate * i)).astype(np.int16)


def sine_format(nch, n, rate=48000):                                         # GenerateAudio.GenerateAdpcmSineWave
    return GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([sine(n, FREQS[i % 8], rate) for i in range(nch)], rate))


def oracle_file(fmt, cfg):
    p = po.dsp_params(fmt.SampleRate, fmt.SampleCount, fmt.Looping, fmt.LoopStart, fmt.LoopEnd, cfg.SamplesPerInterleave,
                      cfg.LoopPointAlignment, cfg.TrimFile)
    ch = fmt.Channels
    rc, f = po.dsp_write([c.GetAdpcmAudio() for c in ch], np.stack([c.Coefs for c in ch]), p,
                         gain=[c.Gain for c in ch],
                         start_context=[[c.StartContext.PredScale, c.StartContext.Hist1, c.StartContext.Hist2] for c in ch],
                         loop_context=[[c.LoopContext.PredScale, c.LoopContext.Hist1, c.LoopContext.Hist2] for c in ch])
    assert rc == 0
    return bytes(f)


@pytest.mark.parametrize("nch", [1, 2, 8])
def test_dsp_build_and_parse_equal(nch):
    fmt = sine_format(nch, 48000)
    f = DspWriter().GetFile(fmt)
    assert f == oracle_file(fmt, DspConfiguration())
    rc, h, coefs, gain, sc, lc, chans = po.dsp_read(f)
    assert rc == 0
    assert (h.sample_count, h.sample_rate, h.channel_count, h.looping) == (48000, 48000, nch, 0)
    for i, c in enumerate(fmt.Channels):
        assert np.array_equal(chans[i], c.GetAdpcmAudio())
        assert coefs[i].tolist() == np.asarray(c.Coefs).tolist()
        assert sc[i].tolist() == [int(c.GetAdpcmAudio()[0]), 0, 0]


@pytest.mark.parametrize("nch,n,spi", [(1, 1, 14), (1, 13, 0x3800), (2, 14, 14), (2, 100, 14), (3, 1000, 14 * 8), (5, 14 * 64 * 3, 14 * 64),
                                       (2, 14 * 64 * 3 + 1, 14 * 64), (7, 50001, 14 * 1000), (16, 100000, 0x3800),
                                       (64, 30000, 14 * 512)])
def test_file_matches_oracle_geometries(nch, n, spi):
    rng = np.random.default_rng(nch * 1000 + n)
    pcm = [rng.integers(-32768, 32768, n).astype(np.int16) for _ in range(nch)]
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(pcm, 32000))
    cfg = DspConfiguration(SamplesPerInterleave=spi)
    f = DspWriter(cfg).GetFile(fmt)
    assert f == oracle_file(fmt, cfg)
    rc, h, coefs, gain, sc, lc, chans = po.dsp_read(f)
    assert rc == 0 and h.channel_count == nch
    for i, c in enumerate(fmt.Channels):
        assert np.array_equal(chans[i], c.GetAdpcmAudio())


@pytest.mark.parametrize("nch", [1, 2, 4])
@pytest.mark.parametrize("loop,align,trim", [((1400, 9000), 0, True), ((1399, 9001), 0, False), ((1000, 9000), 1024, True),
                                             ((1000, 9000), 1024, False), ((0, 12000), 0, True)])
def test_looping_files_match_oracle(nch, loop, align, trim):
    n = 12000
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 44100)).WithLoop(True, *loop)
    if align:
        fmt = fmt.WithAlignment(align)                                        # aligned audio + recomputed loop context
    cfg = DspConfiguration(SamplesPerInterleave=14 * 128, TrimFile=trim)
    f = DspWriter(cfg).GetFile(fmt)
    assert f == oracle_file(fmt, cfg)
    rc, h, coefs, gain, sc, lc, chans = po.dsp_read(f)
    assert rc == 0 and h.looping == 1
    assert h.start_addr == po.gc_sample_to_nibble(fmt.LoopStart) and h.end_addr == po.gc_sample_to_nibble(fmt.LoopEnd)
    for i, c in enumerate(fmt.Channels):
        assert lc[i].tolist() == [c.LoopContext.PredScale, c.LoopContext.Hist1, c.LoopContext.Hist2]


def test_loop_point_alignment_moves_header_only():
    # DspWriter.cs:29-31: LoopPointAlignment shifts the header's loop points; the audio is written as it is.
    # Multi-channel: the shorter input leaves zero gaps; mono: Stream.Write past the array throws.
    pcm = synth.generate(2, 9000)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000)).WithLoop(True, 100, 9000)
    cfg = DspConfiguration(LoopPointAlignment=1024, SamplesPerInterleave=14 * 32)
    assert DspWriter(cfg).GetFile(fmt) == oracle_file(fmt, cfg)
    mono = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([pcm[0]], 48000)).WithLoop(True, 100, 9000)
    with pytest.raises(_lib.ArgumentError):
        DspWriter(cfg).GetFile(mono)


def test_configuration_errors():
    for bad in (0, -14, 15):
        with pytest.raises(_lib.ArgumentOutOfRangeError):
            DspConfiguration(SamplesPerInterleave=bad)
    p = _lib.DspParamsC(48000, 100, 0, 0, 0, 15, 1, 1)
    L = _lib.DspLayoutC()
    assert _lib.lib().vga_dsp_layout_for(C.byref(p), 2, C.byref(L)) == _lib.ArgumentOutOfRangeError.code
    p.samples_per_interleave = 14
    assert _lib.lib().vga_dsp_layout_for(C.byref(p), 0, C.byref(L)) == _lib.ArgumentError.code


def test_device_resident_image_with_default_start_context():
    """vga_dsp_write_device straight from the encoder's HBM output; NULL start context = (Adpcm[0], 0, 0)."""
    nch, n = 12, 14 * 5000 + 3
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    nb = po.gc_sample_count_to_byte_count(n)
    pitch = (nb + 15) // 16 * 16
    host = np.zeros((nch, pitch), np.uint8)
    for i, c in enumerate(fmt.Channels):
        host[i, :nb] = c.GetAdpcmAudio()
    d_adpcm = torch.from_numpy(host).cuda()
    d_coefs = torch.from_numpy(np.stack([c.Coefs for c in fmt.Channels]).astype(np.int16)).cuda()
    cfg = DspConfiguration(SamplesPerInterleave=14 * 256)
    p = DspWriter(cfg)._params(fmt)
    L = DspWriter(cfg).Layout(fmt)
    d_file = torch.full((L.file_size,), 0xEE, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(_lib.lib().vga_dsp_write_device(d_adpcm.data_ptr(), pitch, nb, d_coefs.data_ptr(), None, None, None, nch,
                                               C.byref(p), d_file.data_ptr(), stream))
    torch.cuda.synchronize()
    assert d_file.cpu().numpy().tobytes() == oracle_file(fmt, cfg)
