"""GPU parity for the ADX and HCA container writers: images assembled in HBM
(vga_adx_write / _device, vga_hca_write_device) equal the oracle's byte for byte."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.adx import AdxConfiguration, AdxWriter
from vgaudio_amd.criadx import CriAdxFormat, CriAdxParameters, CriAdxType
from vgaudio_amd.crihca import CriHcaFormat, CriHcaParameters
from vgaudio_amd.gcadpcm import Pcm16Format
from vgaudio_amd.hca import HcaConfiguration, HcaWriter

pytestmark = pytest.mark.gpu


def oracle_adx_file(fmt, cfg):
    p = po.adxfile_params(fmt.SampleRate, fmt.SampleCount, fmt.Looping, fmt.LoopStart, fmt.LoopEnd, fmt.AlignmentSamples,
                          fmt.FrameSize, fmt.Version, fmt.Type, fmt.HighpassFrequency, cfg.EncryptionType, cfg.TrimFile)
    rc, f = po.adxfile_write([c.Audio for c in fmt.Channels], [c.History for c in fmt.Channels], p)
    assert rc == 0
    return bytes(f)


@pytest.mark.parametrize("nch,n", [(1, 1), (1, 33), (2, 32), (2, 48000), (3, 1000), (6, 20001), (32, 100000)])
@pytest.mark.parametrize("version", [3, 4])
def test_adx_file_matches_oracle(nch, n, version):
    pcm = synth.generate(nch, n)
    cfg = AdxConfiguration(Version=version)
    f = AdxWriter(cfg).GetFile(Pcm16Format(list(pcm), 44100))                 # encodes with the writer's settings (:39-51)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 44100), CriAdxParameters(Version=version, Filter=2))
    assert f == oracle_adx_file(fmt, cfg)
    rc, h, hist, chans = po.adxfile_read(f)
    assert rc == 0 and (h.channel_count, h.sample_count, h.version) == (nch, n, version)
    for a, c in zip(chans, fmt.Channels):
        assert np.array_equal(a, c.Audio)


@pytest.mark.parametrize("nch", [1, 2, 4])
@pytest.mark.parametrize("version", [3, 4])
@pytest.mark.parametrize("loop,trim", [((1000, 9000), True), ((1000, 9000), False), ((0, 12000), True), ((37, 11999), True),
                                       ((4096, 4097), False)])
def test_adx_looping_file_matches_oracle(nch, version, loop, trim):
    pcm = synth.generate(nch, 12000)
    src = Pcm16Format(list(pcm), 48000).WithLoop(True, *loop)
    fmt = CriAdxFormat().EncodeFromPcm16(src, CriAdxParameters(Version=version))
    assert fmt.Looping and fmt.LoopStart % (64 if nch == 1 else 32) == 0
    cfg = AdxConfiguration(Version=version, TrimFile=trim)
    f = AdxWriter(cfg).GetFile(fmt)
    assert f == oracle_adx_file(fmt, cfg)
    rc, h, hist, chans = po.adxfile_read(f)
    assert rc == 0 and h.looping == 1 and (h.loop_start_sample, h.loop_end_sample) == (fmt.LoopStart, fmt.LoopEnd)


@pytest.mark.parametrize("type_,filt", [(CriAdxType.Fixed, 1), (CriAdxType.Exponential, 0)])
def test_adx_file_other_encodings(type_, filt):
    pcm = synth.generate(2, 5000)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 32000), CriAdxParameters(Type=type_, Filter=filt))
    cfg = AdxConfiguration(Type=type_, Filter=filt, EncryptionType=8)
    assert AdxWriter(cfg).GetFile(fmt) == oracle_adx_file(fmt, cfg)


def test_adx_device_resident_image():
    nch, n = 24, 32 * 4000 + 5
    pcm = synth.generate(nch, n)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    nb = len(fmt.Channels[0].Audio)
    pitch = nb + 2                                                             # an even pitch that is not a multiple of 4
    host = np.zeros((nch, pitch), np.uint8)
    for i, c in enumerate(fmt.Channels):
        host[i, :nb] = c.Audio
    d_audio = torch.from_numpy(host).cuda()
    d_hist = torch.tensor([c.History for c in fmt.Channels], dtype=torch.int16).cuda()
    w = AdxWriter()
    L, p = w.Layout(fmt), w._params(fmt)
    d_file = torch.full((L.file_size,), 0xEE, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().vga_adx_write_device(d_audio.data_ptr(), pitch, nb, d_hist.data_ptr(), nch, C.byref(p), d_file.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert d_file.cpu().numpy().tobytes() == oracle_adx_file(fmt, w.Configuration)


def test_adx_errors():
    p = _lib.AdxFileParamsC(48000, 100, 0, 0, 0, 0, 18, 4, 3, 500, 0, 1)
    L = _lib.AdxFileLayoutC()
    assert _lib.lib().vga_adx_file_layout_for(C.byref(p), 0, C.byref(L)) == _lib.ArgumentError.code
    assert _lib.lib().vga_adx_file_layout_for(C.byref(p), 256, C.byref(L)) == _lib.ArgumentError.code


# ------------------------------------------------------------------ HCA
@pytest.mark.parametrize("nch,n,loop", [(1, 3000, None), (2, 48000, None), (2, 20000, (3000, 17000)), (6, 9000, None)])
def test_hca_file_matches_oracle(nch, n, loop):
    pcm = synth.generate(nch, n)
    src = Pcm16Format(list(pcm), 48000)
    if loop:
        src = src.WithLoop(True, *loop)
    f = HcaWriter().GetFile(src)
    prm = po.hca_params(nch, n, looping=bool(loop), loop_start=loop[0] if loop else 0, loop_end=loop[1] if loop else 0)
    rc, info, frames = po.hca_encode(pcm, prm)
    assert rc == 0
    rc, want = po.hcafile_write(info, frames)
    assert rc == 0 and f == want.tobytes()
    rc, r, vol, enc, comment, ver = po.hcafile_read(f)
    assert rc == 0 and (r.channel_count, r.frame_count, r.frame_size, r.looping) == (nch, info.frame_count, info.frame_size, int(bool(loop)))


def test_hca_device_batch_of_files():
    """vga_hca_write_device: a batch of equally shaped streams -> one image each, straight from the encoder's output."""
    ns, nch, n = 5, 2, 12000
    pcm = synth.generate(ns * nch, n).reshape(ns, nch, n)
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(pcm[s]), 48000) for s in range(ns)], CriHcaParameters())
    hca = fmts[0].Hca
    audio = hca.FrameCount * hca.FrameSize
    fpitch = audio + 6
    host = np.zeros((ns, fpitch), np.uint8)
    for s in range(ns):
        host[s, :audio] = np.asarray(fmts[s].AudioData).reshape(-1)
    d_frames = torch.from_numpy(host).cuda()
    size = _lib.lib().vga_hca_file_size(C.byref(hca.c))
    pitch = size + 10
    d_files = torch.full((ns, pitch), 0xEE, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().vga_hca_write_device(C.byref(hca.c), d_frames.data_ptr(), fpitch, ns, None, 1.0, 0, 0, d_files.data_ptr(), pitch,
                                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    got = d_files.cpu().numpy()
    rc, info, oframes = po.hca_encode_batch(pcm, po.hca_params(nch, n))
    assert rc == 0
    for s in range(ns):
        rc, want = po.hcafile_write(info, oframes[s])
        assert rc == 0 and got[s, :size].tobytes() == want.tobytes()
        assert (got[s, size:] == 0xEE).all()
