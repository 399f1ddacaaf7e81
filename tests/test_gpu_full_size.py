"""Full BASELINE.json sizes on one GPU: config 2 (4096 channels x 60 s GC-ADPCM), config 3 (the same through CRI ADX),
config 4 (1024 stereo HCA streams).  Every unit is checked through decode(encode(x)) ~ x; against the oracle, bit for bit:
EVERY channel of config 3 and EVERY stream of config 4, encoded bytes and decoded PCM (the C restatements of ADX and HCA
run a whole configuration in ~10 s on the box's 16 host threads, a slice of the batch at a time), and for config 2
256 channels spread over the batch encoded by the oracle in the test itself plus ALL 4096 channels against the digests
the oracle wrote in the build container (tests/golden/gc_shard_oracle_digests.json; the GC-ADPCM restatement needs
150 s for 4096 channels on the box's host threads)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
from vgaudio_amd import _lib, device as vdev

pytestmark = pytest.mark.gpu

N = 2_880_000                                   # 48 kHz x 60 s
THREADS = max(1, min(16, len(os.sched_getaffinity(0))))
# relative rms error of decode(encode(x)) over a whole channel; calibrated on the synthetic channels (4-bit codes:
# the quietest, noisiest channels set the maximum) with a factor of two of slack
GC_BOUND, ADX_BOUND, HCA_BOUND = 0.17, 0.8, 0.12           # measured maxima 0.086, 0.534 (13 kHz tones), 0.060


def _rel_rms(dec, pcm, n, chunk=256):
    out = []
    for c0 in range(0, pcm.shape[0], chunk):
        a = dec[c0:c0 + chunk, :n].to(torch.float32)
        b = pcm[c0:c0 + chunk, :n].to(torch.float32)
        out.append(((a - b).pow(2).mean(dim=1).sqrt() / b.pow(2).mean(dim=1).sqrt().clamp_min(1.0)).cpu())
    return torch.cat(out)


def test_config2_gcadpcm_4096_channels():
    d = torch.device("cuda:0")
    nch = 4096
    pcm = vdev.synth_pcm(nch, N, d)
    coefs = vdev.gc_coefs(pcm, N)
    adpcm = vdev.gc_encode(pcm, N, coefs)
    dec, status = vdev.gc_decode(adpcm, coefs, N)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    rel = _rel_rms(dec, pcm, N)
    print("config 2 relative rms: max %.4f mean %.4f" % (float(rel.max()), float(rel.mean())))
    assert (rel < GC_BOUND).all() and float(rel.mean()) < 0.04, float(rel.max())
    del dec
    nb = vdev.gc_byte_count(N)
    # 256 channels spread over the batch (first and last included: row offsets beyond 4 GiB), bit for bit against
    # the oracle with the reference's scheduling (one task per channel)
    idx = torch.arange(0, nch, 16, device=d)
    idx[-1] = nch - 1
    assert idx.numel() == 256
    host = pcm[idx, :N].cpu().numpy()
    wc, wa = po.gc_encode_batch(host, threads=THREADS)
    assert np.array_equal(coefs[idx].cpu().numpy().reshape(256, 16), np.asarray(wc).reshape(256, 16))
    assert np.array_equal(adpcm[idx, :nb].cpu().numpy(), np.asarray(wa)[:, :nb])
    wd = po.gc_decode_batch(np.asarray(wa)[:, :nb], np.asarray(wc).reshape(256, 16), N, threads=THREADS)
    dec, status = vdev.gc_decode(adpcm[idx].contiguous(), coefs[idx].contiguous(), N)
    torch.cuda.synchronize()
    assert int(status.item()) == 0 and np.array_equal(dec[:, :N].cpu().numpy(), wd)
    # every frame header names a predictor 0..7 and a scale 0..12 (GcAdpcmEncoder.cs:83, :118-170)
    heads = adpcm[:, :nb - nb % 8].reshape(nch, -1, 8)[:, :, 0]
    assert int((heads >> 4).max()) <= 7 and int((heads & 15).max()) <= 12
    # EVERY channel against the oracle: the digests of what oracle/gcadpcm_oracle.c produces for these 4096 channels,
    # written in the build container without a GPU (tests/golden/make_gc_shard_oracle_digests.py, 11 core-minutes per
    # 1024 channels) -- one 64-bit digest per channel (names the channel that differs), the shard's positional digest
    # and SHA-256 over coefficients + rows
    import hashlib
    import json
    from vgaudio_amd import distributed as vdist
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gold = json.load(open(os.path.join(gold_dir, "gc_shard_oracle_digests.json")))
    assert gold["provenance"] == "oracle" and gold["samples_per_channel"] == N and gold["bytes_per_row"] == nb
    per_channel = np.load(os.path.join(gold_dir, "gc_channel_oracle_digests.npy"))
    inner = vdist.row_digests(adpcm, nb, coefs).cpu().numpy().view(np.uint64)
    differ = np.nonzero(inner != per_channel[:nch])[0]
    assert differ.size == 0, "channels that differ from the oracle: %s" % differ[:16].tolist()
    sha = hashlib.sha256()
    sha.update(coefs.cpu().numpy().tobytes())
    for c0 in range(0, nch, 512):
        sha.update(adpcm[c0:c0 + 512, :nb].cpu().numpy().tobytes())
    shard0 = [s for s in gold["shards"] if s["rank"] == 0][0]
    assert "0x%016x" % vdist.rows_digest(adpcm, nb, coefs, 0) == shard0["rows_digest"] and sha.hexdigest() == shard0["sha256"]


def test_config3_adx_4096_channels():
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nch = 4096
    pcm = vdev.synth_pcm(nch, N, d)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(N, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    hist = torch.zeros(nch, dtype=torch.int16, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    dec = vdev.alloc_pcm(nch, N, d)
    _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, N, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
    _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, N, C.byref(p), dec.data_ptr(), dec.stride(0),
                                       status.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    rel = _rel_rms(dec, pcm, N)
    print("config 3 relative rms: max %.4f mean %.4f" % (float(rel.max()), float(rel.mean())))
    assert (rel < ADX_BOUND).all() and float(rel.mean()) < 0.08, float(rel.max())
    # every channel against the oracle, bit for bit: bytes, final history, decoded PCM (512 channels at a time)
    for c0 in range(0, nch, 512):
        host = pcm[c0:c0 + 512, :N].cpu().numpy()
        want, whist = po.adx_encode_batch(host, po.adx_params(), threads=THREADS)
        assert np.array_equal(adx[c0:c0 + 512, :nb].cpu().numpy(), want), c0
        assert np.array_equal(hist[c0:c0 + 512].cpu().numpy(), whist), c0
        assert np.array_equal(dec[c0:c0 + 512, :N].cpu().numpy(), po.adx_decode_batch(want, N, po.adx_params(), threads=THREADS)), c0
        del host, want
    # frame scales are 13-bit (CriAdxCodec.cs:140-141)
    assert int((adx[:, :nb].reshape(nch, -1, 18)[:, :, 0] >> 5).max()) == 0


def test_config4_hca_1024_stereo_streams():
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    ns = 1024
    hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, N, 0, 0, 0)
    info = _lib.HcaInfoC()
    _lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
    assert (info.frame_size, info.frame_count) == (682, 2813)          # config 4's derived parameters
    spcm = vdev.synth_pcm(ns * 2, N, d)
    ch_pitch = spcm.stride(0)
    fbytes = info.frame_count * info.frame_size
    fpitch = (fbytes + 8 + 15) // 16 * 16
    frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    _lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * ch_pitch, ch_pitch, ns, N, C.byref(info), frames.data_ptr(), fpitch,
                                       status.data_ptr(), st))
    wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns)
    ws = torch.empty(wsb, dtype=torch.uint8, device=d)
    dec = torch.zeros_like(spcm)
    _lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, dec.data_ptr(), 2 * ch_pitch, ch_pitch,
                                       ws.data_ptr(), wsb, status.data_ptr(), st))
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    rel = _rel_rms(dec, spcm, N)
    print("config 4 relative rms: max %.4f mean %.4f" % (float(rel.max()), float(rel.mean())))
    assert (rel < HCA_BOUND).all() and float(rel.mean()) < 0.04, float(rel.max())
    # every frame starts with the sync word and carries a valid CRC (a CRC over data + CRC is 0); sampled
    fr = frames[:, :fbytes].reshape(ns, info.frame_count, info.frame_size)
    assert bool((fr[:, :, 0] == 0xFF).all()) and bool((fr[:, :, 1] == 0xFF).all())
    sample = fr[::97, ::211].reshape(-1, info.frame_size).cpu().numpy()
    for f in sample:
        assert po.lib().vgo_crc16(po._u8(np.ascontiguousarray(f)), info.frame_size) == 0
    # every stream against the oracle, bit for bit: frames and decoded PCM (128 streams at a time)
    for s0 in range(0, ns, 128):
        host = spcm[2 * s0:2 * (s0 + 128), :N].cpu().numpy().reshape(128, 2, N)
        rc, oinfo, want = po.hca_encode_batch(host, po.hca_params(2, N), threads=THREADS)
        assert rc == 0
        assert np.array_equal(fr[s0:s0 + 128].cpu().numpy().reshape(128, -1), want), s0
        rc, wdec = po.hca_decode_batch(oinfo, want, threads=THREADS)
        assert rc == 0
        assert np.array_equal(dec[2 * s0:2 * (s0 + 128), :N].cpu().numpy().reshape(128, 2, N), np.asarray(wdec)), s0
        del host, want, wdec


def test_time_segment_fallbacks_are_exact():
    """GC decode and ADX encode/decode cut long channels into time pieces decoded side by side from a guessed history
    and close the seams afterwards (LABNOTES.md 4.3).  With the test hook no seam is ever accepted as closed, so the
    fall-back paths (serial redo of the rest of the channel) produce the output: it must not change."""
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nch, n = 130, 32 * 14 * 700 + 11           # several pieces, ragged tails, a channel count that fills no workgroup
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    status = torch.zeros(1, dtype=torch.int32, device=d)

    def run():
        enc = vdev.gc_encode(pcm, n, coefs)
        dec, _ = vdev.gc_decode(adpcm, coefs, n)
        adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
        hist = torch.zeros(nch, dtype=torch.int16, device=d)
        back = vdev.alloc_pcm(nch, n, d)
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch,
                                           hist.data_ptr(), st))
        _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0),
                                           status.data_ptr(), st))
        torch.cuda.synchronize()
        return dec[:, :n].clone(), adx[:, :nb].clone(), back[:, :n].clone(), enc[:, :vdev.gc_byte_count(n)].clone()

    # 12 pieces (ADX encode included)
    L.vga_testing_gc_encoder_segments_this_thread(12)
    try:
        normal = run()
        for mode in (1, 2):                      # every seam open / open and closed seams mixed inside a workgroup
            old = L.vga_testing_force_open_seams_this_thread(mode)
            try:
                forced = run()
            finally:
                L.vga_testing_force_open_seams_this_thread(old)
            for a, b in zip(normal, forced):
                assert torch.equal(a, b), mode
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(0)
    # and both equal the oracle on a few channels
    for c in (0, 64, 129):
        host = pcm[c, :n].cpu().numpy()
        assert (normal[3][c].cpu().numpy() == po.gc_encode(host, coefs[c].cpu().numpy())).all()
        assert (normal[0][c].cpu().numpy() == po.gc_decode(adpcm[c, :vdev.gc_byte_count(n)].cpu().numpy(),
                                                            coefs[c].cpu().numpy(), n)).all()
        op = po.adx_params()
        want = po.adx_encode(host, op)
        assert (normal[1][c].cpu().numpy() == want).all()
        assert (normal[2][c].cpu().numpy() == po.adx_decode(want, n, po.adx_params())).all()


@pytest.mark.parametrize("pieces", [40, 700, 3000])
def test_decoders_with_short_pieces_chain_their_open_seams(pieces):
    """Pieces of a few frames: many seams are still open when their piece ends, several in a row, so the chained tail
    kernels produce much of the output (with mode 2 some seams are never accepted as closed on top of that).  The PCM is
    the oracle's."""
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nch, n = 70, 32 * 14 * 55 + 5
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    hist = torch.zeros(nch, dtype=torch.int16, device=d)
    status = torch.zeros(1, dtype=torch.int32, device=d)
    _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
    torch.cuda.synchronize()
    want_gc = [po.gc_decode(adpcm[c, :vdev.gc_byte_count(n)].cpu().numpy(), coefs[c].cpu().numpy(), n) for c in (0, 33, 69)]
    want_adx = po.adx_decode_batch(adx[:, :nb].cpu().numpy()[[0, 33, 69]], n, po.adx_params(), threads=3)
    L.vga_testing_gc_encoder_segments_this_thread(pieces)
    try:
        for mode in (0, 2):
            old = L.vga_testing_force_open_seams_this_thread(mode)
            try:
                dec, _ = vdev.gc_decode(adpcm, coefs, n)
                back = vdev.alloc_pcm(nch, n, d)
                _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0),
                                                   status.data_ptr(), st))
                torch.cuda.synchronize()
            finally:
                L.vga_testing_force_open_seams_this_thread(old)
            for i, c in enumerate((0, 33, 69)):
                assert np.array_equal(dec[c, :n].cpu().numpy(), want_gc[i]), (pieces, mode, c)
                assert np.array_equal(back[c, :n].cpu().numpy(), want_adx[i]), (pieces, mode, c)
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(0)


@pytest.mark.parametrize("n", [32 * 64 * 40, 32 * 64 * 40 + 7])
def test_adx_encoder_with_short_pieces_chains_its_open_seams(n):
    """64-frame pieces: the ADX encoder's seams need ~100 frames to close, so most are still open at the end of their piece
    and the chained tail kernel produces most of the stream.  Bytes and history are the oracle's."""
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nch = 70
    pcm = vdev.synth_pcm(nch, n, d)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    host = pcm[:, :n].cpu().numpy()
    want, whist = po.adx_encode_batch(host, po.adx_params(), threads=4)
    L.vga_testing_gc_encoder_segments_this_thread(1000)
    try:
        for mode in (0, 2):
            old = L.vga_testing_force_open_seams_this_thread(mode)
            try:
                adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
                hist = torch.zeros(nch, dtype=torch.int16, device=d)
                _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch,
                                                   hist.data_ptr(), st))
                torch.cuda.synchronize()
            finally:
                L.vga_testing_force_open_seams_this_thread(old)
            assert np.array_equal(adx[:, :nb].cpu().numpy(), want), mode
            assert np.array_equal(hist.cpu().numpy(), whist), mode
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(0)



@pytest.mark.parametrize("nch", [1, 63, 65])
def test_decoders_block_boundaries(nch):
    """The helper-free decoders work in blocks (GC: 8 frames of 14 samples, ADX: 2 frames of 32) with a per-frame path for
    what is left of a piece and a partial last frame: lengths on, just before and just after every such boundary, one piece
    and several, against the oracle"""
    d = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    lengths = sorted({base + k for base in (14 * 8 * 3, 32 * 2 * 5, 14 * 8 * 32 * 3) for k in (-33, -15, -14, -1, 0, 1, 13, 14, 31, 32, 45)} | {1, 13, 14, 15, 31, 32, 33})
    for n in lengths:
        pcm = vdev.synth_pcm(nch, n, d)
        coefs = vdev.gc_coefs(pcm, n)
        adpcm = vdev.gc_encode(pcm, n, coefs)
        nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
        pitch = (nb + 15) // 16 * 16
        adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
        hist = torch.zeros(nch, dtype=torch.int16, device=d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
        torch.cuda.synchronize()
        want_gc = po.gc_decode_batch(adpcm[:, :vdev.gc_byte_count(n)].cpu().numpy(), coefs.cpu().numpy().reshape(nch, 16), n, threads=4)
        want_adx = po.adx_decode_batch(adx[:, :nb].cpu().numpy(), n, po.adx_params(), threads=4)
        for pieces in (0, 5):
            L.vga_testing_gc_encoder_segments_this_thread(pieces)
            try:
                dec, s1 = vdev.gc_decode(adpcm, coefs, n)
                back = vdev.alloc_pcm(nch, n, d)
                _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0),
                                                   status.data_ptr(), st))
                torch.cuda.synchronize()
            finally:
                L.vga_testing_gc_encoder_segments_this_thread(0)
            assert int(s1.item()) == 0 and int(status.item()) == 0
            assert np.array_equal(dec[:, :n].cpu().numpy(), want_gc), (n, pieces)
            assert np.array_equal(back[:, :n].cpu().numpy(), want_adx), (n, pieces)
            assert int(dec[:, n:].abs().sum()) == 0 and int(back[:, n:].abs().sum()) == 0, (n, pieces)
