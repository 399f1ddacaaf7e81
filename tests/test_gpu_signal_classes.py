"""The codecs on signals other than the synthetic generator's (vgaudio_amd/signals.py: the reference's benchmark tone,
VGAudio.Benchmark/AdpcmBenchmarks/EncodeBenchmarks.cs:8-24; full-scale white noise; +-3 LSB noise; silence; a clipped square;
the slowest-closing synthetic channel on every row) at BASELINE configs[1]'s size -- 4096 channels x 60 s through the kernels
the full batch takes (persistent workgroups, sixteen time pieces per channel, seams closed inside) -- with 64 channels spread
over the batch held to the oracle bit for bit: coefficients, GC-ADPCM bytes, decoded samples, ADX bytes.  The encoders' work
is data-dependent (third trips of the retry loop, GcAdpcmEncoder.cs:127-170; how soon the seams between pieces close), so
parity on one signal family says little about another.  Silence: 16 zero coefficients and zero bytes
(VGAudio.Tests/GenerateAudio.cs:73-88)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from vgaudio_amd import _lib, device as vdev, signals

pytestmark = pytest.mark.gpu

N = 2_880_000
THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


def test_generators_agree_between_host_and_device():
    d = torch.device("cuda:0")
    for cls in signals.CLASSES:
        got = signals.device(cls, 70, 14 * 700 + 5, d, first_channel=4000)
        torch.cuda.synchronize()
        assert np.array_equal(got[:, :14 * 700 + 5].cpu().numpy(), signals.host(cls, 70, 14 * 700 + 5, first_channel=4000)), cls


@pytest.mark.parametrize("cls", signals.CLASSES)
def test_gcadpcm_full_batch_of_a_signal_class_matches_the_oracle(cls):
    d = torch.device("cuda:0")
    nch = 4096
    pcm = signals.device(cls, nch, N, d)
    coefs = vdev.gc_coefs(pcm, N)
    adpcm = vdev.gc_encode(pcm, N, coefs)
    torch.cuda.synchronize()
    nb = vdev.gc_byte_count(N)
    idx = torch.arange(0, nch, 64, device=d)
    idx[-1] = nch - 1
    host = pcm[idx, :N].cpu().numpy()
    if cls != "slow_channel_93":                       # (that one is the device generator's channel 93 on every row: test_gpu_golden)
        assert np.array_equal(host[::8], np.stack([signals.host(cls, 1, N, first_channel=int(c))[0] for c in idx.cpu().numpy()[::8]])), cls
    wc, wa = po.gc_encode_batch(host, threads=THREADS)
    wc = np.asarray(wc).reshape(-1, 16)
    got_c = coefs[idx].cpu().numpy().reshape(-1, 16)
    assert np.array_equal(got_c, wc), (cls, np.nonzero((got_c != wc).any(axis=1))[0][:8].tolist())
    got_a = adpcm[idx, :nb].cpu().numpy()
    bad = np.nonzero((got_a != np.asarray(wa)[:, :nb]).any(axis=1))[0]
    assert bad.size == 0, (cls, "channels", idx.cpu().numpy()[bad][:8].tolist(), "first byte", int(np.nonzero(got_a[bad[0]] != np.asarray(wa)[bad[0], :nb])[0][0]))
    if cls == "silence":
        assert not coefs.any() and not adpcm.any()
    if cls in ("sine440", "clipped_square", "slow_channel_93", "silence"):
        # rows that hold the same samples must hold the same bytes, wherever in the batch they sit (sine: the phase repeats
        # every 1200 channels; the square's period every 97; the slow channel and silence on every row)
        step = {"sine440": 1200, "clipped_square": 97 * 16, "slow_channel_93": 1, "silence": 1}[cls]
        if step == 1:
            assert bool((adpcm[:, :nb] == adpcm[:1, :nb]).all()) and bool((coefs == coefs[:1]).all())
        else:
            a, b = slice(0, nch - step), slice(step, nch)
            same = (pcm[a, :N] == pcm[b, :N]).all(dim=1)
            assert bool(same.all()) or cls == "clipped_square"
            assert bool((adpcm[a, :nb][same] == adpcm[b, :nb][same]).all()) and bool((coefs[a][same] == coefs[b][same]).all())
    dec, status = vdev.gc_decode(adpcm[idx].contiguous(), coefs[idx].contiguous(), N)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert np.array_equal(dec[:, :N].cpu().numpy(), po.gc_decode_batch(np.asarray(wa)[:, :nb], wc, N, threads=THREADS)), cls


@pytest.mark.parametrize("cls", signals.CLASSES)
def test_adx_full_batch_of_a_signal_class_matches_the_oracle(cls):
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nch = 4096
    pcm = signals.device(cls, nch, N, d)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(N, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    hist = torch.zeros(nch, dtype=torch.int16, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    dec = vdev.alloc_pcm(nch, N, d)
    _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, N, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
    _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, N, C.byref(p), dec.data_ptr(), dec.stride(0), status.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    idx = torch.arange(0, nch, 64, device=d)
    idx[-1] = nch - 1
    host = pcm[idx, :N].cpu().numpy()
    want, whist = po.adx_encode_batch(host, po.adx_params(), threads=THREADS)
    got = adx[idx, :nb].cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (cls, "channels", idx.cpu().numpy()[bad][:8].tolist())
    assert np.array_equal(hist[idx].cpu().numpy(), whist), cls
    assert np.array_equal(dec[idx, :N].cpu().numpy(), po.adx_decode_batch(want, N, po.adx_params(), threads=THREADS)), cls


@pytest.mark.parametrize("nch,n,pieces", [(70, 14 * 9000 + 5, 12), (130, 14 * 20000, 0), (64, 14 * 4096 * 3 + 13, 5)])
@pytest.mark.parametrize("mode", [0, 3])
def test_gc_decoder_repair_launch_on_seams_that_never_close(nch, n, pieces, mode):
    """Pure tones (mode 0: the 440 Hz sine's predictor has its poles on the unit circle, a run from a wrong history never falls
    into step) and any signal with the hook's mode 3 (every seam refused and counted): the decoder's REPAIR launch -- the
    affected waves decoded again as one piece -- produces the samples.  The oracle's, sample for sample."""
    d = torch.device("cuda:0")
    L = _lib.lib()
    host = signals.host("sine440", nch, n) if mode == 0 else np.ascontiguousarray(po.synth_generate(nch, n, first_channel=60))
    pcm = vdev.alloc_pcm(nch, n, d)
    pcm[:, :n] = torch.from_numpy(host).to(d)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    torch.cuda.synchronize()
    nb = vdev.gc_byte_count(n)
    wc, wa = po.gc_encode_batch(host, threads=THREADS)
    assert np.array_equal(adpcm[:, :nb].cpu().numpy(), np.asarray(wa)[:, :nb])
    want = po.gc_decode_batch(np.asarray(wa)[:, :nb], np.asarray(wc).reshape(nch, 16), n, threads=THREADS)
    L.vga_testing_gc_encoder_segments_this_thread(pieces)
    old = L.vga_testing_force_open_seams_this_thread(mode)
    try:
        dec, status = vdev.gc_decode(adpcm, coefs, n)
        torch.cuda.synchronize()
    finally:
        L.vga_testing_force_open_seams_this_thread(old)
        L.vga_testing_gc_encoder_segments_this_thread(0)
    assert int(status.item()) == 0
    got = dec[:, :n].cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, (bad[:8].tolist(), int(np.nonzero(got[bad[0]] != want[bad[0]])[0][0]))


@pytest.mark.parametrize("nch,tone_rows,n", [(200, (77,), 14 * 24000 + 3), (320, (5, 250), 14 * 20000)])
def test_decoders_hand_a_few_never_closing_channels_from_the_tail_kernel_to_the_repair_launch(nch, tone_rows, n):
    """A batch of ordinary channels with ONE or two rows whose seams never close (GC-ADPCM: the 440 Hz sine; ADX: the clipped
    square): far fewer open seams than the fix-up's `many`, so the tail kernel chains them -- and a lane that has walked its
    budget (4096 frames, ADX 2048) without meeting hands its channel to the REPAIR launch in mid-launch (first_open[ch] = k + 1,
    slow_seams raised to `many`; gc_decode_kernel.hip / adx_kernels.hip tail kernels).  Every row against the oracle."""
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    host = np.ascontiguousarray(po.synth_generate(nch, n, first_channel=300))
    tone = signals.host("sine440", len(tone_rows), n)
    square = signals.host("clipped_square", len(tone_rows), n)
    # ---- GC-ADPCM
    gc_host = host.copy()
    for i, r in enumerate(tone_rows):
        gc_host[r] = tone[i]
    pcm = vdev.alloc_pcm(nch, n, d)
    pcm[:, :n] = torch.from_numpy(gc_host).to(d)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    nb = vdev.gc_byte_count(n)
    wc, wa = po.gc_encode_batch(gc_host, threads=THREADS)
    torch.cuda.synchronize()
    assert np.array_equal(adpcm[:, :nb].cpu().numpy(), np.asarray(wa)[:, :nb])
    dec, status = vdev.gc_decode(adpcm, coefs, n)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    want = po.gc_decode_batch(np.asarray(wa)[:, :nb], np.asarray(wc).reshape(nch, 16), n, threads=THREADS)
    got = dec[:, :n].cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, ("gc", bad[:8].tolist())
    # ---- ADX (encode: the same few rows go through the encoder's tail kernel; decode as above)
    adx_host = host.copy()
    for i, r in enumerate(tone_rows):
        adx_host[r] = square[i]
    pcm[:, :n] = torch.from_numpy(adx_host).to(d)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nba = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nba + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    hist = torch.zeros(nch, dtype=torch.int16, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    back = vdev.alloc_pcm(nch, n, d)
    _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
    _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nba, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    wadx, whist = po.adx_encode_batch(adx_host, po.adx_params(), threads=THREADS)
    got = adx[:, :nba].cpu().numpy()
    bad = np.nonzero((got != wadx).any(axis=1))[0]
    assert bad.size == 0, ("adx encode", bad[:8].tolist())
    assert np.array_equal(hist.cpu().numpy(), whist)
    wback = po.adx_decode_batch(wadx, n, po.adx_params(), threads=THREADS)
    got = back[:, :n].cpu().numpy()
    bad = np.nonzero((got != wback).any(axis=1))[0]
    assert bad.size == 0, ("adx decode", bad[:8].tolist())
