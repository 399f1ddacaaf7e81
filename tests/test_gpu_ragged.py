"""Ragged batches (the `*_v` entry points): channels of different lengths in ONE call -- the shape of the reference's
file-level batch (VGAudio.Cli/Batch.cs:24-25: Parallel.ForEach over files, each through GcAdpcmFormat.EncodeFromPcm16,
Formats/GcAdpcm/GcAdpcmFormat.cs:58-74).  Every channel of every call must be what one call of the oracle on that
channel alone produces: coefficients, bitstream, decoded samples."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.gcadpcm import GcAdpcmFormat, GcAdpcmParameters, Pcm16Format, decode_files, encode_files

pytestmark = pytest.mark.gpu

i16p, u8p = _lib.i16p, _lib.u8p


def _ptrs(t, arrays):
    return (t * len(arrays))(*[a.ctypes.data_as(t) for a in arrays])


def _lengths(n, lo, hi, seed, extra=()):
    """log-uniform lengths in [lo, hi] samples plus the reference's awkward ones"""
    rng = np.random.default_rng(seed)
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), n)).astype(np.int64)
    lens[:len(extra)] = extra
    rng.shuffle(lens)
    return [int(x) for x in lens]


def _channels(lens, first_channel=0):
    return [po.synth_generate(1, n, first_channel=first_channel + c)[0] if n else np.zeros(0, np.int16) for c, n in enumerate(lens)]


def _oracle(pcm, h1=0, h2=0):
    coefs = po.gc_calculate_coefficients(pcm)
    return coefs, po.gc_encode(pcm, coefs, hist1=h1, hist2=h2)


def _encode_v(chans, h1=None, h2=None):
    L = _lib.lib()
    nch = len(chans)
    counts = np.array([len(c) for c in chans], dtype=np.int32)
    coefs = np.full((nch, 16), 0x5A5A, dtype=np.int16)
    outs = [np.full(L.vga_gcadpcm_sample_count_to_byte_count(int(n)) + 1, 0xEE, dtype=np.uint8) for n in counts]
    _lib.check(L.vga_gcadpcm_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                            h1.ctypes.data_as(i16p) if h1 is not None else None,
                                            h2.ctypes.data_as(i16p) if h2 is not None else None,
                                            coefs.ctypes.data_as(i16p), _ptrs(u8p, outs)))
    for o in outs:
        assert o[-1] == 0xEE, "wrote past the end of a row"
    return coefs, [o[:-1] for o in outs]


def _decode_v(adpcm, coefs, counts, h1=None, h2=None):
    L = _lib.lib()
    nch = len(adpcm)
    counts = np.array(counts, dtype=np.int32)
    outs = [np.full(int(n) + 1, 0x7777, dtype=np.int16) for n in counts]
    co = np.ascontiguousarray(coefs, dtype=np.int16)
    _lib.check(L.vga_gcadpcm_decode_batch_v(_ptrs(u8p, adpcm), co.ctypes.data_as(i16p), counts.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                            h1.ctypes.data_as(i16p) if h1 is not None else None,
                                            h2.ctypes.data_as(i16p) if h2 is not None else None, _ptrs(i16p, outs)))
    for o in outs:
        assert o[-1] == 0x7777, "wrote past the end of a row"
    return [o[:-1] for o in outs]


AWKWARD = (0, 1, 13, 14, 15, 27, 28, 29, 8 * 14, 8 * 14 + 1, 3072 * 14, 3072 * 14 + 5)


def test_ragged_encode_and_decode_match_the_oracle_per_channel():
    lens = _lengths(96, 200, 400_000, 1, AWKWARD)
    chans = _channels(lens)
    coefs, adpcm = _encode_v(chans)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist(), (c, lens[c])
        assert np.array_equal(adpcm[c], wa), (c, lens[c])
    pcm_back = _decode_v(adpcm, coefs, lens)
    for c in range(len(chans)):
        assert np.array_equal(pcm_back[c], po.gc_decode(adpcm[c], coefs[c], lens[c])), (c, lens[c])


def test_ragged_with_per_channel_history():
    lens = _lengths(40, 100, 60_000, 2, (0, 1, 14, 15))
    chans = _channels(lens, first_channel=500)
    rng = np.random.default_rng(5)
    h1 = rng.integers(-32768, 32768, len(lens)).astype(np.int16)
    h2 = rng.integers(-32768, 32768, len(lens)).astype(np.int16)
    coefs, adpcm = _encode_v(chans, h1, h2)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm, int(h1[c]), int(h2[c]))
        assert coefs[c].tolist() == wc.tolist() and np.array_equal(adpcm[c], wa), (c, lens[c])
    back = _decode_v(adpcm, coefs, lens, h1, h2)
    for c in range(len(chans)):
        assert np.array_equal(back[c], po.gc_decode(adpcm[c], coefs[c], lens[c], hist1=int(h1[c]), hist2=int(h2[c]))), c


@pytest.mark.parametrize("force_open", [0, 1, 2, 3])
def test_ragged_many_pieces_and_open_seams(force_open):
    """few long channels next to many short ones: the long ones are cut into many time pieces; with the seam hook every
    seam (or every other one) is refused, so the chain / tail paths produce the output"""
    L = _lib.lib()
    lens = [14 * 40_000 + 3, 14 * 31_000, 14 * 25_000 + 13] + _lengths(61, 50, 30_000, 3, (0, 5, 14 * 3072))
    chans = _channels(lens, first_channel=64)          # (channels 64.. hold the slow-closing seams of the synthetic set)
    L.vga_testing_force_open_seams_this_thread(force_open)
    L.vga_testing_gc_encoder_segments_this_thread(24)
    try:
        coefs, adpcm = _encode_v(chans)
        back = _decode_v(adpcm, coefs, lens)
    finally:
        L.vga_testing_force_open_seams_this_thread(0)
        L.vga_testing_gc_encoder_segments_this_thread(0)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist() and np.array_equal(adpcm[c], wa), (c, lens[c])
        assert np.array_equal(back[c], po.gc_decode(wa, wc, lens[c])), (c, lens[c])


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("force_open", [0, 1, 2])
@pytest.mark.parametrize("ragged", [False, True])
def test_persistent_workgroups_and_plain_grid_give_the_oracles_bytes(mode, force_open, ragged):
    """the encoder hands its (channel group, piece) items out either as a plain grid + a seam launch (1) or to persistent
    workgroups that take them from a queue and close the seams themselves (2); neither may change a byte"""
    L = _lib.lib()
    if ragged:
        lens = [14 * 30_000 + 5, 14 * 30_000, 14 * 22_000 + 1] + _lengths(45, 100, 14 * 25_000, 21, (0, 3, 14 * 2500))
    else:
        lens = [14 * 20_000 + 9] * 40
    chans = _channels(lens, first_channel=64)
    L.vga_testing_gc_encoder_persistent_this_thread(mode)
    L.vga_testing_force_open_seams_this_thread(force_open)
    L.vga_testing_gc_encoder_segments_this_thread(12)
    try:
        coefs, adpcm = _encode_v(chans)
    finally:
        L.vga_testing_gc_encoder_persistent_this_thread(0)
        L.vga_testing_force_open_seams_this_thread(0)
        L.vga_testing_gc_encoder_segments_this_thread(0)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist() and np.array_equal(adpcm[c], wa), (c, lens[c])


def test_a_channel_longer_than_the_two_size_schedule_holds_is_encoded_to_its_end():
    """A 21-minute channel in a batch of one group: the persistent schedule's two piece sizes would need more than the 1024
    pieces the scratch holds (tests/test_host_gc_piece_plan.py); until round 5 the plan was cut off there and the tail of the
    channel kept whatever the output row held.  ~20 s of oracle time."""
    lens = [60_000_000 + 9, 1000, 14 * 4000 + 3]
    chans = _channels(lens, first_channel=93)          # (channel 93: the slowest-closing seams of the synthetic set)
    coefs, adpcm = _encode_v(chans)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist(), (c, lens[c])
        bad = np.flatnonzero(adpcm[c] != wa)
        assert bad.size == 0, (c, lens[c], "first differing byte", int(bad[0]), "of", wa.size)
    back = _decode_v(adpcm, coefs, lens)
    for c in range(len(chans)):
        assert np.array_equal(back[c], po.gc_decode(adpcm[c], coefs[c], lens[c])), (c, lens[c])


def test_equal_lengths_through_the_ragged_entry_point_equal_the_batch_entry_point():
    L = _lib.lib()
    n = 14 * 5000 + 9
    chans = list(synth.generate(20, n))
    coefs, adpcm = _encode_v(chans)
    want = np.zeros((20, 16), dtype=np.int16)
    outs = [np.zeros(L.vga_gcadpcm_sample_count_to_byte_count(n), dtype=np.uint8) for _ in chans]
    _lib.check(L.vga_gcadpcm_encode_batch(_ptrs(i16p, chans), 20, n, 0, 0, want.ctypes.data_as(i16p), _ptrs(u8p, outs)))
    assert np.array_equal(coefs, want)
    for a, b in zip(adpcm, outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("chunk,slot", [(7, 0), (5, 4096), (64, -1)])
def test_ragged_through_every_pipeline_shape(chunk, slot):
    """small chunks, staged / direct copies: every hand-off of the host pipeline with rows of different sizes"""
    L = _lib.lib()
    lens = _lengths(50, 10, 20_000, 4, (0, 0, 1, 14))
    chans = _channels(lens, first_channel=200)
    L.vga_testing_host_pipeline_this_thread(2, 2, chunk, slot)
    try:
        coefs, adpcm = _encode_v(chans)
        back = _decode_v(adpcm, coefs, lens)
    finally:
        L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist() and np.array_equal(adpcm[c], wa), (c, lens[c])
        assert np.array_equal(back[c], po.gc_decode(wa, wc, lens[c])), c


@pytest.mark.parametrize("variant", [1, 3])
def test_ragged_coefficients_from_either_kernel_form(variant):
    """The ragged coefficient search has two forms (gcadpcm_kernels.hip launch_coefs): one wave per channel (hook 1) and five waves
    per channel for every channel that holds a chunk of records (hook 3; the shorter ones stay on the one-wave kernel in the same
    call) -- the launcher picks by the batch's shape; both against the oracle, awkward lengths included."""
    L = _lib.lib()
    lens = _lengths(60, 20, 90_000, 11, AWKWARD + (14 * 64 - 1, 14 * 64, 14 * 64 + 1, 14 * 256, 14 * 257 + 3))
    chans = _channels(lens, first_channel=1300)
    counts = np.array(lens, dtype=np.int32)
    coefs = np.full((len(lens), 16), 0x5A5A, dtype=np.int16)
    L.vga_testing_gc_coefs_variant_this_thread(variant)
    try:
        _lib.check(L.vga_gcadpcm_calculate_coefficients_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), len(lens),
                                                                coefs.ctypes.data_as(i16p)))
    finally:
        L.vga_testing_gc_coefs_variant_this_thread(0)
    for c, pcm in enumerate(chans):
        assert coefs[c].tolist() == po.gc_calculate_coefficients(pcm).tolist(), (c, lens[c])


def test_coefficients_only_and_encode_with_given_coefficients():
    L = _lib.lib()
    lens = _lengths(30, 20, 50_000, 6, (0, 13))
    chans = _channels(lens, first_channel=900)
    counts = np.array(lens, dtype=np.int32)
    coefs = np.zeros((len(lens), 16), dtype=np.int16)
    _lib.check(L.vga_gcadpcm_calculate_coefficients_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), len(lens),
                                                            coefs.ctypes.data_as(i16p)))
    for c, pcm in enumerate(chans):
        assert coefs[c].tolist() == po.gc_calculate_coefficients(pcm).tolist(), c
    rng = np.random.default_rng(9)
    given = rng.integers(-3000, 3000, (len(lens), 16)).astype(np.int16)
    outs = [np.zeros(L.vga_gcadpcm_sample_count_to_byte_count(n), dtype=np.uint8) for n in lens]
    _lib.check(L.vga_gcadpcm_encode_with_coefs_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), len(lens),
                                                       given.ctypes.data_as(i16p), None, None, _ptrs(u8p, outs)))
    for c, pcm in enumerate(chans):
        assert np.array_equal(outs[c], po.gc_encode(pcm, given[c])), c


def test_files_of_different_shapes_in_one_call():
    """the host mirror of Batch.cs: one Pcm16Format per file, any length and channel count"""
    rng = np.random.default_rng(12)
    files = []
    for f in range(9):
        n = int(rng.integers(1, 40_000))
        k = int(rng.integers(1, 4))
        files.append(Pcm16Format(list(po.synth_generate(k, n, first_channel=10 * f)), sampleRate=32000 + f))
    files.append(Pcm16Format([], 48000))
    got = encode_files(files, [GcAdpcmParameters(History1=3 * i, History2=-i) if i % 2 else None for i in range(len(files))])
    for i, (f, g) in enumerate(zip(files, got)):
        want = GcAdpcmFormat().EncodeFromPcm16(f, GcAdpcmParameters(History1=3 * i, History2=-i) if i % 2 else None)
        assert g.ChannelCount == f.ChannelCount and g.SampleRate == f.SampleRate
        for a, b in zip(g.Channels, want.Channels):
            assert a.Coefs.tolist() == b.Coefs.tolist() and np.array_equal(a.Adpcm, b.Adpcm), i
    back = decode_files(got)
    for i, (g, p) in enumerate(zip(got, back)):
        want = g.ToPcm16()
        for a, b in zip(p.Channels, want.Channels):
            assert np.array_equal(a, b), i


def test_ragged_arguments():
    L = _lib.lib()
    a = np.zeros(20, dtype=np.int16)
    out = np.zeros(16, dtype=np.uint8)
    coefs = np.zeros(16, dtype=np.int16)
    bad = np.array([-1], dtype=np.int32)
    assert L.vga_gcadpcm_encode_batch_v(_ptrs(i16p, [a]), bad.ctypes.data_as(C.POINTER(C.c_int)), 1, None, None, coefs.ctypes.data_as(i16p),
                                        _ptrs(u8p, [out])) == _lib.VGA_ERR_ARGUMENT
    ok = np.array([20], dtype=np.int32)
    assert L.vga_gcadpcm_encode_batch_v(_ptrs(i16p, [a]), ok.ctypes.data_as(C.POINTER(C.c_int)), 1, None, None, coefs.ctypes.data_as(i16p),
                                        None) == _lib.VGA_ERR_ARGUMENT
    assert L.vga_gcadpcm_encode_batch_v(None, None, 0, None, None, None, None) == _lib.VGA_OK


def test_device_resident_ragged_batch():
    import torch
    from vgaudio_amd import device as vdev
    dev = torch.device("cuda", 0)
    lens = _lengths(300, 500, 300_000, 8, AWKWARD)
    rb = vdev.GcRaggedBatch(lens, dev)
    pcm = rb.synth(first_channel=1000)
    host = rb.rows(pcm, rb.pcm_offsets, rb.counts)
    for c in (0, 17, 150, 299):
        assert np.array_equal(host[c], po.synth_generate(1, lens[c], first_channel=1000 + c)[0] if lens[c] else host[c]), c
    coefs = rb.coefs(pcm)
    adpcm = rb.encode(pcm, coefs)
    dec, status = rb.decode(adpcm, coefs)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    co = coefs.cpu().numpy()
    rows = rb.rows(adpcm, rb.adpcm_offsets, rb.byte_counts)
    back = rb.rows(dec, rb.pcm_offsets, rb.counts)
    for c in range(0, 300, 7):
        wc, wa = _oracle(host[c])
        assert co[c].tolist() == wc.tolist() and np.array_equal(rows[c], wa), (c, lens[c])
        assert np.array_equal(back[c], po.gc_decode(wa, wc, lens[c])), (c, lens[c])
    rb.close()


# ---------------------------------------------------------------------------------------------- CRI ADX, CRI HCA
def _adx_param_sets():
    """(reference-side parameters for the oracle, the same as vga_adx_params) -- different files, different parameters"""
    sets = [dict(), dict(sample_rate=44100), dict(sample_rate=22050, version=3), dict(type=4), dict(type=2, filter=2),
            dict(frame_size=34), dict(sample_rate=32000, highpass_frequency=500)]
    return sets


def test_adx_ragged_encode_and_decode_match_the_oracle_per_channel():
    L = _lib.lib()
    sets = _adx_param_sets()
    lens = _lengths(70, 40, 120_000, 31, (1, 31, 32, 33, 63, 64, 65))
    chans = _channels(lens, first_channel=300)
    nch = len(lens)
    params = (_lib.AdxParams * nch)()
    oracle_params = []
    for c in range(nch):
        L.vga_adx_default_params(C.byref(params[c]))
        kw = sets[c % len(sets)]
        for k, v in kw.items():
            setattr(params[c], k, v)
        oracle_params.append(po.adx_params(**kw))
    counts = np.array(lens, dtype=np.int32)
    outs = [np.full(L.vga_adx_encoded_byte_count(n, C.byref(params[c])) + 1, 0xEE, dtype=np.uint8) for c, n in enumerate(lens)]
    hist = np.full(nch, 0x1234, dtype=np.int16)
    _lib.check(L.vga_adx_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch, params, _ptrs(u8p, outs),
                                        hist.ctypes.data_as(i16p)))
    for c, pcm in enumerate(chans):
        p = oracle_params[c]
        want = po.adx_encode(pcm, p)
        assert outs[c][-1] == 0xEE and np.array_equal(outs[c][:-1], want), (c, lens[c], sets[c % len(sets)])
        assert int(hist[c]) == int(p.history), c                    # CriAdxCodec.cs:73: the encoder leaves History in the config
    # decode what was encoded, every channel with its own length and parameters
    enc = [o[:-1].copy() for o in outs]
    alens = np.array([len(e) for e in enc], dtype=np.int32)
    pcm_out = [np.full(n + 1, 0x7777, dtype=np.int16) for n in lens]
    _lib.check(L.vga_adx_decode_batch_v(_ptrs(u8p, enc), alens.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                        counts.ctypes.data_as(C.POINTER(C.c_int)), params, _ptrs(i16p, pcm_out)))
    for c in range(nch):
        p = po.adx_params(**sets[c % len(sets)])
        want = po.adx_decode(enc[c], lens[c], p)
        assert pcm_out[c][-1] == 0x7777 and np.array_equal(pcm_out[c][:-1], want), (c, lens[c])


@pytest.mark.parametrize("kw", [dict(type=4, padding=10), dict(type=2, filter=2, padding=10), dict(padding=33), dict(padding=32)])
def test_adx_ragged_empty_channels_with_padding_keep_the_skipped_frame_zero(kw):
    """An empty channel whose padding ends inside a frame: the reference skips that frame (`samplesToCopy == 0`,
    CriAdxCodec.cs:84) and leaves zeros; a zero-padded run of a longer channel would put silence's header there.  Empty
    channels therefore never share a bucket with longer ones (round 4's advisor)."""
    L = _lib.lib()
    lens = [0, 500, 0, 900, 64, 0, 1000]
    chans = _channels(lens, first_channel=700)
    nch = len(lens)
    params = (_lib.AdxParams * nch)()
    for c in range(nch):
        L.vga_adx_default_params(C.byref(params[c]))
        for k, v in kw.items():
            setattr(params[c], k, v)
    counts = np.array(lens, dtype=np.int32)
    outs = [np.full(L.vga_adx_encoded_byte_count(n, C.byref(params[c])) + 1, 0xEE, dtype=np.uint8) for c, n in enumerate(lens)]
    _lib.check(L.vga_adx_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch, params, _ptrs(u8p, outs), None))
    for c, pcm in enumerate(chans):
        want = po.adx_encode(pcm, po.adx_params(**kw))
        assert outs[c][-1] == 0xEE and np.array_equal(outs[c][:-1], want), (c, lens[c], kw, outs[c][:40].tolist(), want[:40].tolist())


def test_adx_ragged_rejects_what_the_reference_rejects():
    L = _lib.lib()
    params = (_lib.AdxParams * 2)()
    for c in range(2):
        L.vga_adx_default_params(C.byref(params[c]))
    chans = [np.zeros(64, np.int16), np.zeros(0, np.int16)]
    counts = np.array([64, 0], dtype=np.int32)                       # an empty version-4 channel: the reference reads pcm[0]
    outs = [np.zeros(64, np.uint8), np.zeros(16, np.uint8)]
    assert L.vga_adx_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), 2, params, _ptrs(u8p, outs),
                                    None) == _lib.VGA_ERR_ARGUMENT


def test_hca_ragged_streams_match_the_oracle_per_stream():
    """mono and stereo files, two sample rates, three qualities, one looping stream, lengths from half a frame to 60 frames"""
    L = _lib.lib()
    rng = np.random.default_rng(41)
    shapes = []
    for s in range(22):
        nch = 1 if s % 3 == 0 else 2
        n = int(rng.integers(300, 60 * 1024))
        shapes.append(dict(channel_count=nch, sample_count=n, sample_rate=48000 if s % 4 else 44100,
                           quality=("High", "Middle", "Highest")[s % 3]))
    shapes.append(dict(channel_count=2, sample_count=20_000, looping=True, loop_start=3000, loop_end=17_000))
    shapes.append(dict(channel_count=2, sample_count=1024 * 7))
    shapes.append(dict(channel_count=2, sample_count=1024 * 7 - 128))
    ns = len(shapes)
    pcm, rows = [], []
    for s, sh in enumerate(shapes):
        a = po.synth_generate(sh["channel_count"], sh["sample_count"], first_channel=40 * s)
        pcm.append(a)
        rows += [a[c] for c in range(sh["channel_count"])]
    configs = (_lib.HcaParamsC * ns)()
    oparams = []
    for s, sh in enumerate(shapes):
        op = po.hca_params(**sh)
        oparams.append(op)
        for f, _ in _lib.HcaParamsC._fields_:
            setattr(configs[s], f, getattr(op, f))
    infos = (_lib.HcaInfoC * ns)()
    want = []
    for s in range(ns):
        rc, info, frames = po.hca_encode(pcm[s], oparams[s])
        assert rc == 0, s
        want.append((info, frames))
    outs = [np.full(w[0].frame_count * w[0].frame_size + 1, 0xEE, dtype=np.uint8) for w in want]
    _lib.check(L.vga_hca_encode_batch_v(_ptrs(i16p, rows), ns, configs, infos, _ptrs(u8p, outs)))
    for s in range(ns):
        info, frames = want[s]
        for f, _ in _lib.HcaInfoC._fields_:
            assert getattr(infos[s], f) == getattr(info, f), (s, f)
        assert outs[s][-1] == 0xEE and np.array_equal(outs[s][:-1], frames.reshape(-1)), (s, shapes[s])
    # decode: every stream from its own HcaInfo
    enc = [o[:-1].copy() for o in outs]
    pcm_out = []
    for s in range(ns):
        pcm_out += [np.full(max(infos[s].sample_count, 0) + 1, 0x7777, dtype=np.int16) for _ in range(infos[s].channel_count)]
    _lib.check(L.vga_hca_decode_batch_v(infos, _ptrs(u8p, enc), ns, _ptrs(i16p, pcm_out)))
    at = 0
    for s in range(ns):
        rc, dec = po.hca_decode(want[s][0], want[s][1])
        assert rc == 0
        for c in range(infos[s].channel_count):
            assert pcm_out[at][-1] == 0x7777 and np.array_equal(pcm_out[at][:-1], dec[c]), (s, c)
            at += 1


def test_adx_and_hca_files_of_different_shapes_in_one_call():
    """the host mirrors of Batch.cs for the other two codecs: the per-file classes must return the same formats"""
    from vgaudio_amd import criadx, crihca
    rng = np.random.default_rng(77)
    files = []
    for f in range(8):
        n = int(rng.integers(2000, 50_000))
        k = int(rng.integers(1, 3))
        p = Pcm16Format(list(po.synth_generate(k, n, first_channel=30 * f)), sampleRate=(48000, 44100, 32000)[f % 3])
        if f == 5:
            p.WithLoop(True, 1234, n - 500)
        files.append(p)
    got = criadx.encode_files(files, [criadx.CriAdxParameters(Version=3) if f % 4 == 3 else None for f in range(len(files))])
    for f, (pcm16, g) in enumerate(zip(files, got)):
        want = criadx.CriAdxFormat().EncodeFromPcm16(pcm16, criadx.CriAdxParameters(Version=3) if f % 4 == 3 else None)
        assert (g.SampleRate, g.AlignmentSamples, g.Version, g.Looping) == (want.SampleRate, want.AlignmentSamples, want.Version, want.Looping)
        for a, b in zip(g.Channels, want.Channels):
            assert np.array_equal(a.Audio, b.Audio) and a.History == b.History, f
    got = crihca.encode_files(files)
    for f, (pcm16, g) in enumerate(zip(files, got)):
        want = crihca.CriHcaFormat().EncodeFromPcm16(pcm16)
        assert g.Hca.FrameCount == want.Hca.FrameCount and g.Hca.Looping == want.Hca.Looping
        assert np.array_equal(g.AudioData, want.AudioData), f


@pytest.mark.parametrize("pieces,force_open", [(0, 0), (12, 0), (12, 1), (40, 0), (40, 1)])
def test_adx_ragged_bucket_padding_is_nobodys_output(pieces, force_open):
    """Channels of one length bucket are zero-padded on the device to the bucket's longest and run through the equal-length
    kernels cut into time pieces; a seam that lies in a channel's padding -- or runs into it -- is left alone (round 5: in
    digital silence two runs need never meet, and the ragged call of bench.py's 10 008 files spent 100 ms per bucket chaining
    them).  Every channel must still be its own encoding / decoding, with short pieces and with every seam held open too
    (CriAdxCodec.cs:56-104, :9-54)."""
    L = _lib.lib()
    # one bucket (within a quarter of each other), lengths that end inside different pieces, one a multiple of 32
    lens = [100_000 + 997 * i for i in range(23)] + [3200 * 32]
    chans = _channels(lens, first_channel=90)          # channel 93 (the slowest to fall into step) among them
    chans[5] = (12000 * np.sign(np.sin(np.arange(lens[5]) * (2 * np.pi / 64)))).astype(np.int16)      # a frame-periodic square: never meets
    nch = len(lens)
    params = (_lib.AdxParams * nch)()
    for c in range(nch):
        L.vga_adx_default_params(C.byref(params[c]))
    counts = np.array(lens, dtype=np.int32)
    outs = [np.full(L.vga_adx_encoded_byte_count(n, C.byref(params[c])) + 1, 0xEE, dtype=np.uint8) for c, n in enumerate(lens)]
    old_p = L.vga_testing_gc_encoder_segments_this_thread(pieces)
    old_f = L.vga_testing_force_open_seams_this_thread(force_open)
    try:
        _lib.check(L.vga_adx_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch, params, _ptrs(u8p, outs), None))
        enc = []
        for c, pcm in enumerate(chans):
            want = po.adx_encode(pcm, po.adx_params())
            assert outs[c][-1] == 0xEE and np.array_equal(outs[c][:-1], want), (c, lens[c])
            enc.append(want.copy())
        alens = np.array([len(e) for e in enc], dtype=np.int32)
        pcm_out = [np.full(n + 1, 0x7777, dtype=np.int16) for n in lens]
        _lib.check(L.vga_adx_decode_batch_v(_ptrs(u8p, enc), alens.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                            counts.ctypes.data_as(C.POINTER(C.c_int)), params, _ptrs(i16p, pcm_out)))
        for c in range(nch):
            assert pcm_out[c][-1] == 0x7777 and np.array_equal(pcm_out[c][:-1], po.adx_decode(enc[c], lens[c], po.adx_params())), (c, lens[c])
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(old_p)
        L.vga_testing_force_open_seams_this_thread(old_f)


# ---------------------------------------------------------------------------------------------- seeded random files
@pytest.mark.parametrize("seed", range(int(os.environ.get("VGA_SWEEP_CASES", "0")) or 10))
def test_random_files_through_the_three_ragged_entry_points(seed):
    """Files nobody thought of (tests/test_gpu_random_sweep.py for the equal-length entry points): random counts, lengths on and
    around the block sizes, per-file ADX parameters (type, version, padding, rate), per-file HCA shapes (channels, quality,
    rate, loop) -- every file against the oracle."""
    L = _lib.lib()
    rng = np.random.default_rng(50_000 + seed)

    def length(top):
        blocks = (14, 32, 14 * 64, 14 * 256, 1024, 32 * 64, 14 * 3072)
        if rng.random() < 0.5:
            return max(1, min(top, int(blocks[rng.integers(0, len(blocks))]) * int(rng.integers(1, 4)) + int(rng.integers(-2, 3))))
        return int(np.exp(rng.uniform(0.0, np.log(top))))

    # ---- GC-ADPCM
    nfiles = int(rng.integers(1, 50))
    lens = [length(200_000) for _ in range(nfiles)]
    if rng.random() < 0.3:
        lens[int(rng.integers(0, nfiles))] = 0
    chans = _channels(lens, first_channel=700 + 60 * seed)
    coefs, adpcm = _encode_v(chans)
    for c, pcm in enumerate(chans):
        wc, wa = _oracle(pcm)
        assert coefs[c].tolist() == wc.tolist() and np.array_equal(adpcm[c], wa), ("gc", seed, c, lens[c])
    back = _decode_v(adpcm, coefs, lens)
    for c in range(nfiles):
        assert np.array_equal(back[c], po.gc_decode(adpcm[c], coefs[c], lens[c])), ("gc decode", seed, c, lens[c])

    # ---- CRI ADX
    nfiles = int(rng.integers(1, 40))
    lens = [length(150_000) for _ in range(nfiles)]
    chans = _channels(lens, first_channel=900 + 60 * seed)
    params = (_lib.AdxParams * nfiles)()
    kws = []
    for c in range(nfiles):
        L.vga_adx_default_params(C.byref(params[c]))
        kw = dict(type=int(rng.choice([2, 3, 4])), version=int(rng.choice([3, 4])))
        if kw["type"] == 2:
            kw["filter"] = int(rng.integers(0, 4))
        if rng.random() < 0.5:
            kw["padding"] = int(rng.integers(0, 80))
        if rng.random() < 0.4:
            kw["sample_rate"] = int(rng.choice([22050, 32000, 44100]))
        if rng.random() < 0.1:
            kw["frame_size"] = 34
        for k, v in kw.items():
            setattr(params[c], k, v)
        kws.append(kw)
    counts = np.array(lens, dtype=np.int32)
    outs = [np.full(L.vga_adx_encoded_byte_count(n, C.byref(params[c])) + 1, 0xEE, dtype=np.uint8) for c, n in enumerate(lens)]
    hist = np.full(nfiles, 0x1234, dtype=np.int16)
    _lib.check(L.vga_adx_encode_batch_v(_ptrs(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nfiles, params, _ptrs(u8p, outs),
                                        hist.ctypes.data_as(i16p)))
    for c, pcm in enumerate(chans):
        p = po.adx_params(**kws[c])
        want = po.adx_encode(pcm, p)
        assert outs[c][-1] == 0xEE and np.array_equal(outs[c][:-1], want), ("adx", seed, c, lens[c], kws[c])
        assert int(hist[c]) == int(p.history), ("adx history", seed, c, kws[c])
    enc = [o[:-1].copy() for o in outs]
    alens = np.array([len(e) for e in enc], dtype=np.int32)
    pcm_out = [np.full(n + 1, 0x7777, dtype=np.int16) for n in lens]
    _lib.check(L.vga_adx_decode_batch_v(_ptrs(u8p, enc), alens.ctypes.data_as(C.POINTER(C.c_int)), nfiles,
                                        counts.ctypes.data_as(C.POINTER(C.c_int)), params, _ptrs(i16p, pcm_out)))
    for c in range(nfiles):
        dk = {k: v for k, v in kws[c].items() if k != "filter"}
        want = po.adx_decode(enc[c], lens[c], po.adx_params(**kws[c]))
        assert pcm_out[c][-1] == 0x7777 and np.array_equal(pcm_out[c][:-1], want), ("adx decode", seed, c, lens[c], dk)

    # ---- CRI HCA
    ns = int(rng.integers(1, 14))
    shapes = []
    for s in range(ns):
        n = max(1, length(40_000))
        sh = dict(channel_count=int(rng.choice([1, 2, 2, 2, 3, 4, 6])), sample_count=n,
                  sample_rate=int(rng.choice([48000, 44100, 32000])), quality=str(rng.choice(["Highest", "High", "Middle", "Low", "Lowest"])))
        if rng.random() < 0.3 and n > 10:
            sh.update(looping=True, loop_start=int(rng.integers(0, n - 1)))
            sh["loop_end"] = int(rng.integers(sh["loop_start"] + 1, n + 1))
        shapes.append(sh)
    pcm, rows = [], []
    for s, sh in enumerate(shapes):
        a = po.synth_generate(sh["channel_count"], sh["sample_count"], first_channel=40 * s + 1000 * seed)
        pcm.append(a)
        rows += [a[c] for c in range(sh["channel_count"])]
    configs = (_lib.HcaParamsC * ns)()
    want = []
    for s, sh in enumerate(shapes):
        op = po.hca_params(**sh)
        for f, _ in _lib.HcaParamsC._fields_:
            setattr(configs[s], f, getattr(op, f))
        rc, info, frames = po.hca_encode(pcm[s], op)
        assert rc == 0, ("hca oracle", seed, s, sh)
        want.append((info, frames))
    infos = (_lib.HcaInfoC * ns)()
    outs = [np.full(w[0].frame_count * w[0].frame_size + 1, 0xEE, dtype=np.uint8) for w in want]
    _lib.check(L.vga_hca_encode_batch_v(_ptrs(i16p, rows), ns, configs, infos, _ptrs(u8p, outs)))
    for s in range(ns):
        info, frames = want[s]
        for f, _ in _lib.HcaInfoC._fields_:
            assert getattr(infos[s], f) == getattr(info, f), ("hca info", seed, s, f)
        assert outs[s][-1] == 0xEE and np.array_equal(outs[s][:-1], frames.reshape(-1)), ("hca", seed, s, shapes[s])
    enc = [o[:-1].copy() for o in outs]
    pcm_out = []
    for s in range(ns):
        pcm_out += [np.full(max(infos[s].sample_count, 0) + 1, 0x7777, dtype=np.int16) for _ in range(infos[s].channel_count)]
    _lib.check(L.vga_hca_decode_batch_v(infos, _ptrs(u8p, enc), ns, _ptrs(i16p, pcm_out)))
    at = 0
    for s in range(ns):
        rc, dec = po.hca_decode(want[s][0], want[s][1])
        assert rc == 0
        for c in range(infos[s].channel_count):
            assert pcm_out[at][-1] == 0x7777 and np.array_equal(pcm_out[at][:-1], dec[c]), ("hca decode", seed, s, c, shapes[s])
            at += 1
