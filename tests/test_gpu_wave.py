"""GPU parity for the WAVE reader/writer: the device transposes against the oracle, the
reference's own build -> parse identities (VGAudio.Tests/Containers/WaveTests.cs:9-43, 16-bit cases), and the whole
chain WAVE -> GC-ADPCM -> DSP file through device kernels only."""
import ctypes as C
import struct

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.dsp import DspWriter
from vgaudio_amd.gcadpcm import GcAdpcmFormat, Pcm16Format
from vgaudio_amd.wave import WaveReader, WaveWriter

pytestmark = pytest.mark.gpu

FREQS = [261.63, 329.63, 392, 523.25, 659.25, 783.99, 1046.50, 130.81]


def sine(n, f, rate):
    i = np.arange(n, dtype=np.float64)
    return np.trunc(32767 * np.sin(2 * np.pi * f / rate * i)).astype(np.int16)


@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("looped", [False, True])
def test_wave_pcm16_build_and_parse_equal(nch, looped):
    audio = Pcm16Format([sine(48000, FREQS[i], 48000) for i in range(nch)], 48000).WithLoop(looped)
    f = WaveWriter.GetFile(audio)
    rc, want = po.wave_write(audio.Channels, 48000, looped, audio.LoopStart, audio.LoopEnd)
    assert rc == 0 and f == want.tobytes()
    parsed = WaveReader.ReadFormat(f)
    assert (parsed.ChannelCount, parsed.SampleCount, parsed.SampleRate) == (nch, 48000, 48000)
    assert (parsed.Looping, parsed.LoopStart, parsed.LoopEnd) == (audio.Looping, audio.LoopStart, audio.LoopEnd)
    for a, b in zip(audio.Channels, parsed.Channels):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nch,n", [(1, 1), (1, 255), (2, 256), (2, 257), (3, 1000), (31, 513), (32, 512), (33, 700), (100, 3000),
                                   (5, 200001)])
def test_transposes_match_oracle(nch, n):
    rng = np.random.default_rng(nch * 7 + n)
    pcm = [rng.integers(-32768, 32768, n).astype(np.int16) for _ in range(nch)]
    audio = Pcm16Format(pcm, 22050)
    f = WaveWriter.GetFile(audio)
    rc, want = po.wave_write(pcm, 22050)
    assert rc == 0 and f == want.tobytes()
    rc, w, chans = po.wave_read(f)
    got = WaveReader.ReadFormat(f)
    for a, b, c in zip(pcm, got.Channels, chans):
        assert np.array_equal(a, b) and np.array_equal(a, c)


def test_unaligned_data_chunk_on_device():
    """A data chunk at an odd offset inside a device-resident file (byte-wise loads), straight into planar rows."""
    nch, n = 6, 10007
    rng = np.random.default_rng(2)
    inter = rng.integers(-32768, 32768, (n, nch)).astype("<i2")
    blob = np.concatenate([np.zeros(3, np.uint8), np.frombuffer(inter.tobytes(), np.uint8)])
    d_blob = torch.from_numpy(blob).cuda()
    pitch = n + 9
    d_pcm = torch.full((nch, pitch), 0x5A5A, dtype=torch.int16, device="cuda")
    _lib.check(_lib.lib().vga_wave_deinterleave_pcm16_device(d_blob.data_ptr() + 3, n, nch, d_pcm.data_ptr(), pitch,
                                                              torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = d_pcm.cpu().numpy()
    assert np.array_equal(got[:, :n], inter.T)
    assert (got[:, n:] == 0x5A5A).all()
    # and back, into an odd destination
    p = _lib.WaveParamsC(48000, n, 0, 0, 0)
    size = _lib.lib().vga_wave_file_size(C.byref(p), nch)
    d_file = torch.zeros(size + 1, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().vga_wave_write_pcm16_device(d_pcm.data_ptr(), pitch, nch, C.byref(p), d_file.data_ptr() + 1,
                                                       torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    rc, want = po.wave_write(list(inter.T), 48000)
    assert d_file.cpu().numpy()[1:].tobytes() == want.tobytes()


def test_wave_to_dsp_chain():
    """The converter's main road (WAVE in, DSP out): every step a device kernel, the result equal to the oracle's."""
    nch, n = 2, 30000
    pcm = synth.generate(nch, n)
    rc, wav = po.wave_write(list(pcm), 32000, True, 1400, 28000)
    audio = WaveReader.ReadFormat(wav.tobytes())
    fmt = GcAdpcmFormat().EncodeFromPcm16(audio)
    f = DspWriter().GetFile(fmt)
    coefs = [po.gc_calculate_coefficients(pcm[c]) for c in range(nch)]
    adpcm = [po.gc_encode(pcm[c], coefs[c]) for c in range(nch)]
    chans = [po.gc_build_channel(adpcm[c], coefs[c], po.gc_channel_params(n, True, 1400, 28000)) for c in range(nch)]
    rc, h, rcoefs, gain, sc, lc, raud = po.dsp_read(f)
    assert rc == 0 and (h.sample_count, h.looping, h.channel_count) == (28000, 1, nch)
    for c in range(nch):
        assert rcoefs[c].tolist() == coefs[c].tolist()
        assert np.array_equal(raud[c], adpcm[c][:len(raud[c])])
        assert chans[c][0] == 0 and lc[c].tolist() == chans[c][5].tolist()      # loop context at sample 1400


def test_errors():
    with pytest.raises(_lib.InvalidDataError):
        WaveReader.ReadFormat(b"RIFX" + bytes(60))
    eight = (b"RIFF" + struct.pack("<i", 36 + 8) + b"WAVEfmt " + struct.pack("<iHhiihH", 16, 1, 1, 8000, 8000, 1, 8)
             + b"data" + struct.pack("<i", 8) + bytes(8))
    assert WaveReader.ReadMetadata(eight).bits_per_sample == 8
    with pytest.raises(_lib.ArgumentError):
        WaveReader.ReadFormat(eight)
    p = _lib.WaveParamsC(48000, 0x7FFFFFFF // 2, 0, 0, 0)
    assert _lib.lib().vga_wave_file_size(C.byref(p), 2) == _lib.ArgumentOutOfRangeError.code
