"""Seeded random sweeps over the parameter space of the three codecs, every output against the oracle bit for bit.  The fixed
cases of test_gpu_gcadpcm / test_gpu_adx / test_gpu_hca name the shapes somebody thought of; these draw the ones nobody did:
lengths around the kernels' block sizes (14, 32, 64 x 14, 256 x 14, 1024), paddings, histories, sample rates, bitrates, loop
points, channel counts, signal kinds, forced time pieces.  VGA_SWEEP_CASES=n (environment) draws n cases per codec instead of
the default few dozen (round 6 ran 1200 per codec on the final kernels: 3600 passed); VGA_SWEEP_BIG=1 draws 60-300 channels x
3-30 s instead -- the sizes at which the time pieces and the persistent workgroups engage by themselves (30 per codec: 90 passed)."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.criadx import CriAdxCodec, CriAdxParameters
from vgaudio_amd.crihca import CriHcaDecoder, CriHcaFormat, CriHcaParameters
from vgaudio_amd.gcadpcm import Pcm16Format

pytestmark = pytest.mark.gpu
CASES = int(os.environ.get("VGA_SWEEP_CASES", "0"))
BIG = int(os.environ.get("VGA_SWEEP_BIG", "0"))             # 1: 60-300 channels x 3-30 s (the sizes at which the time pieces engage by themselves)
THREADS = max(1, min(16, len(os.sched_getaffinity(0))))


def _signal(rng, nch, n, first_channel):
    kind = rng.integers(0, 6)
    t = np.arange(n)
    if kind == 0 or n == 0:
        return synth.generate(nch, n, first_channel=first_channel) if n else np.zeros((nch, 0), np.int16)
    if kind == 1:
        return rng.integers(-32768, 32768, (nch, n)).astype(np.int16)
    if kind == 2:
        return rng.integers(-5, 6, (nch, n)).astype(np.int16)
    if kind == 3:
        period = int(rng.integers(2, 70))
        return np.tile(np.where((t // period) % 2 == 0, 32767, -32768).astype(np.int16), (nch, 1))
    if kind == 4:
        f = rng.uniform(20.0, 20000.0)
        return np.tile(np.round(32767 * rng.uniform(0.01, 1.0) * np.sin(2 * np.pi * f * t / 48000.0)).astype(np.int16), (nch, 1))
    x = synth.generate(nch, n, first_channel=first_channel).astype(np.int32) * 3       # clipped
    return np.clip(x, -32768, 32767).astype(np.int16)


def _length(rng, top):
    """lengths that sit on, just below and just above the block sizes the kernels use, and log-uniform ones"""
    blocks = (14, 32, 14 * 64, 14 * 256, 1024, 32 * 64, 14 * 3072, 14 * 4096)
    if rng.random() < 0.5:
        b = int(blocks[rng.integers(0, len(blocks))]) * int(rng.integers(1, 4))
        return max(1, min(top, b + int(rng.integers(-2, 3))))
    return int(np.exp(rng.uniform(0.0, np.log(top))))


@pytest.mark.parametrize("seed", range(CASES or 30))
def test_gcadpcm_random_shapes(seed):
    from vgaudio_amd import device as vdev
    import torch
    rng = np.random.default_rng(10_000 + seed)
    L = _lib.lib()
    d = torch.device("cuda:0")
    nch = int(rng.integers(1, 40)) if rng.random() < 0.8 else int(rng.integers(60, 140))
    n = _length(rng, 400_000 if nch < 40 else 150_000)
    if BIG:
        nch, n = int(rng.integers(60, 300)), int(rng.integers(150_000, 1_500_000))
    host = _signal(rng, nch, n, first_channel=seed * 200)
    pcm = vdev.alloc_pcm(nch, n, d)
    pcm[:, :n] = torch.from_numpy(np.ascontiguousarray(host)).to(d)
    L.vga_testing_gc_coefs_variant_this_thread(int(rng.integers(0, 4)))
    L.vga_testing_gc_encoder_segments_this_thread(int(rng.choice([0, 0, 3, 12, 40])))
    L.vga_testing_gc_encoder_layout_this_thread(int(rng.choice([0, 4, 8])))
    try:
        coefs = vdev.gc_coefs(pcm, n)
        adpcm = vdev.gc_encode(pcm, n, coefs)
        dec, status = vdev.gc_decode(adpcm, coefs, n)
        torch.cuda.synchronize()
    finally:
        L.vga_testing_gc_coefs_variant_this_thread(0)
        L.vga_testing_gc_encoder_segments_this_thread(0)
        L.vga_testing_gc_encoder_layout_this_thread(0)
    assert int(status.item()) == 0
    nb = vdev.gc_byte_count(n)
    wc, wa = po.gc_encode_batch(host, threads=THREADS)
    assert np.array_equal(coefs.cpu().numpy().reshape(nch, 16), np.asarray(wc).reshape(nch, 16)), (nch, n)
    assert np.array_equal(adpcm[:, :nb].cpu().numpy(), np.asarray(wa)[:, :nb]), (nch, n)
    want = po.gc_decode_batch(np.asarray(wa)[:, :nb], np.asarray(wc).reshape(nch, 16), n, threads=THREADS)
    assert np.array_equal(dec[:, :n].cpu().numpy(), want), (nch, n)


def _adx_oracle_params(kw):
    m = dict(Type="type", Filter="filter", Version="version", FrameSize="frame_size", Padding="padding",
             SampleRate="sample_rate", History="history", HighpassFrequency="highpass_frequency")
    return po.adx_params(**{m[k]: v for k, v in kw.items()})


@pytest.mark.parametrize("seed", range(CASES or 40))
def test_adx_random_shapes(seed):
    rng = np.random.default_rng(20_000 + seed)
    L = _lib.lib()
    nch = int(rng.integers(1, 12)) if rng.random() < 0.7 else int(rng.integers(60, 140))
    n = _length(rng, 300_000 if nch < 12 else 100_000)
    if BIG:
        nch, n = int(rng.integers(60, 300)), int(rng.integers(150_000, 1_500_000))
    kw = dict(Type=int(rng.choice([2, 3, 4])), Version=int(rng.choice([3, 4])))
    if kw["Type"] == 2:
        kw["Filter"] = int(rng.integers(0, 4))
    r = rng.random()
    if r < 0.6:
        kw["Padding"] = int(rng.integers(0, 65))             # the time-piece kernels
    elif r < 0.75:
        kw["Padding"] = int(rng.integers(65, 200))           # the general kernel
    if rng.random() < 0.15:
        kw["FrameSize"] = int(rng.choice([10, 20, 34]))
    if rng.random() < 0.4:
        kw["SampleRate"] = int(rng.choice([8000, 22050, 32000, 44100, 96000]))
    if rng.random() < 0.3:
        kw["HighpassFrequency"] = int(rng.choice([0, 100, 500, 2000]))
    host = _signal(rng, nch, n, first_channel=seed * 150)
    L.vga_testing_gc_encoder_segments_this_thread(int(rng.choice([0, 0, 5, 12, 40])))      # (the ADX kernels read the same hook)
    L.vga_testing_force_open_seams_this_thread(int(rng.choice([0, 0, 0, 1, 2, 3])))
    try:
        cfg = CriAdxParameters(**kw)
        enc = CriAdxCodec.Encode(list(host), cfg)
        dkw = {k: v for k, v in kw.items() if k != "Filter"}
        dec = CriAdxCodec.Decode(enc, n, CriAdxParameters(**dkw))
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(0)
        L.vga_testing_force_open_seams_this_thread(0)
    want, whist = po.adx_encode_batch(host, _adx_oracle_params(kw), threads=THREADS)
    wdec = po.adx_decode_batch(want, n, _adx_oracle_params(dkw), threads=THREADS)
    for c in range(nch):
        assert len(enc[c]) == want.shape[1] and (enc[c] == want[c]).all(), (kw, nch, n, c)
        assert int(np.atleast_1d(cfg.History)[c]) == int(whist[c]), (kw, nch, n, c)
        assert (dec[c] == wdec[c]).all(), (kw, nch, n, c)


@pytest.mark.parametrize("seed", range(CASES or 24))
def test_hca_random_shapes(seed):
    rng = np.random.default_rng(30_000 + seed)
    nch = int(rng.choice([1, 1, 2, 2, 2, 2, 3, 4, 5, 6, 8]))
    n = int(rng.integers(100_000, 600_000)) if BIG else _length(rng, 60_000)
    quality = str(rng.choice(["Highest", "High", "Middle", "Low", "Lowest"]))
    q = {"Highest": 1, "High": 2, "Middle": 3, "Low": 4, "Lowest": 5}[quality]
    rate = int(rng.choice([48000, 48000, 44100, 32000, 22050, 16000]))
    bitrate = int(rng.choice([0, 0, 0, 64000, 128000, 320000]))
    limit = bool(rng.random() < 0.2)
    looping = bool(rng.random() < 0.35) and n > 10
    ns = int(rng.integers(1, 4))
    streams = [_signal(rng, nch, n, first_channel=seed * 40 + 8 * s) for s in range(ns)]
    pcms = [Pcm16Format(list(s), rate) for s in streams]
    loop_start = loop_end = 0
    if looping:
        loop_start = int(rng.integers(0, n - 1))
        loop_end = int(rng.integers(loop_start + 1, n + (50 if rng.random() < 0.2 else 1)))
        for pcm in pcms:
            pcm.Looping, pcm.LoopStart, pcm.LoopEnd = True, loop_start, loop_end
    p = po.hca_params(nch, n, sample_rate=rate, quality=quality, bitrate=bitrate, limit_bitrate=limit, looping=looping,
                      loop_start=loop_start, loop_end=loop_end)
    rc, info = po.hca_init(p)
    what = (nch, n, quality, rate, bitrate, limit, looping, loop_start, loop_end)
    if rc != 0:                                            # parameters the reference rejects: so must the library
        with pytest.raises(_lib.VgaError):
            CriHcaFormat.EncodeBatchFromPcm16(pcms, CriHcaParameters(Quality=q, Bitrate=bitrate, LimitBitrate=limit))
        return
    fmts = CriHcaFormat.EncodeBatchFromPcm16(pcms, CriHcaParameters(Quality=q, Bitrate=bitrate, LimitBitrate=limit))
    wants = []
    for s, fmt in zip(streams, fmts):
        rc, info, want = po.hca_encode(s, p)
        assert rc == 0, what
        for k, v in info.as_dict().items():
            assert getattr(fmt.Hca.c, k) == v, (k, what)
        assert fmt.AudioData.shape == want.shape, what
        bad = np.argwhere(fmt.AudioData != want)
        assert bad.size == 0, (what, bad[0].tolist(), len(bad))
        wants.append(want)
    dec = CriHcaDecoder.Decode(fmts[0].Hca, [fmt.AudioData for fmt in fmts])
    for want, d in zip(wants, dec):
        rc, pcm = po.hca_decode(info, want)
        assert rc == 0, what
        for c in range(nch):
            assert (d[c] == pcm[c]).all(), (what, c, int(np.argmax(d[c] != pcm[c])))
