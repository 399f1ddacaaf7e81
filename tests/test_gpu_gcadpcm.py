"""GPU parity tests for GC-ADPCM: the HIP path (through the C ABI) must be
bit-exact with the CPU oracle, and must reproduce the reference's own KATs.
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth
from vgaudio_amd.gcadpcm import (GcAdpcmCoefficients, GcAdpcmDecoder, GcAdpcmEncoder, GcAdpcmFormat,
                                 GcAdpcmMath, GcAdpcmParameters, Pcm16Format)

pytestmark = pytest.mark.gpu


def _edge_channels(n, rng):
    """Inputs chosen to hit the rare branches: silence, DC, full-scale square/noise,
    impulses, ramps, tiny signals, alternating extremes."""
    t = np.arange(n)
    chans = {
        "silence": np.zeros(n),
        "dc_max": np.full(n, 32767),
        "dc_min": np.full(n, -32768),
        "square_fs": np.where((t // 7) % 2 == 0, 32767, -32768),
        "alt_fs": np.where(t % 2 == 0, 32767, -32768),
        "noise_fs": rng.integers(-32768, 32768, n),
        "noise_small": rng.integers(-3, 4, n),
        "impulse": np.where(t % 97 == 0, 30000, 0),
        "ramp": ((t * 37) % 65536) - 32768,
        "sine440": synth.sine(n).astype(np.int64),
        "sine56": synth.sine(n, 1, 56).astype(np.int64),
        "burst": np.where((t // 500) % 2 == 0, rng.integers(-20000, 20000, n), 0),
        "tiny_then_loud": np.concatenate([rng.integers(-2, 3, n // 2), rng.integers(-32768, 32768, n - n // 2)]),
    }
    return {k: v.astype(np.int16) for k, v in chans.items()}


def test_coefs_match_oracle_synthetic():
    pcm = synth.generate(16, 14 * 3000 + 5)
    got = GcAdpcmCoefficients.CalculateCoefficients(pcm)
    for c in range(pcm.shape[0]):
        assert got[c].tolist() == po.gc_calculate_coefficients(pcm[c]).tolist(), c


@pytest.mark.parametrize("n", [14 * 700, 14 * 700 + 9])
def test_coefs_match_oracle_edge_inputs(n):
    rng = np.random.default_rng(7)
    chans = _edge_channels(n, rng)
    got = GcAdpcmCoefficients.CalculateCoefficients(list(chans.values()))
    for i, (name, pcm) in enumerate(chans.items()):
        assert got[i].tolist() == po.gc_calculate_coefficients(pcm).tolist(), name


@pytest.mark.parametrize("n", [0, 1, 2, 13, 14, 15, 27, 28, 29, 255, 256, 257, 14 * 256, 14 * 256 + 1, 14 * 257])
def test_coefs_ragged_lengths(n):
    rng = np.random.default_rng(n)
    pcm = rng.integers(-20000, 20000, (3, n)).astype(np.int16)
    got = GcAdpcmCoefficients.CalculateCoefficients(list(pcm))
    for c in range(3):
        assert got[c].tolist() == po.gc_calculate_coefficients(pcm[c]).tolist()


def test_encode_matches_oracle_synthetic():
    pcm = synth.generate(24, 14 * 2000 + 11)
    coefs = np.stack([po.gc_calculate_coefficients(p) for p in pcm])
    got = GcAdpcmEncoder.Encode(list(pcm), coefs)
    for c in range(pcm.shape[0]):
        assert (got[c] == po.gc_encode(pcm[c], coefs[c])).all(), c


@pytest.mark.parametrize("n", [14 * 300, 14 * 300 + 3])
def test_encode_matches_oracle_edge_inputs_and_coefs(n):
    rng = np.random.default_rng(11)
    chans = _edge_channels(n, rng)
    names = list(chans)
    pcm = [chans[k] for k in names]
    # real coefs, then hostile ones.  |c| <= 16383 keeps the int32 predictor from wrapping, so the
    # reference's retry loop provably terminates; the full-range set can reach the state where
    # the reference loops forever (GcAdpcmEncoder.cs:166-170) -- oracle and kernel carry the same
    # termination guard, and still have to agree byte for byte.
    real = np.stack([po.gc_calculate_coefficients(p) for p in pcm])
    bounded = rng.integers(-16383, 16384, real.shape).astype(np.int16)
    wrapping = rng.integers(-32768, 32768, real.shape).astype(np.int16)
    wrapping[0] = 32767
    wrapping[1] = -32768
    zero = np.zeros_like(real)
    for label, coefs in (("real", real), ("bounded", bounded), ("zero", zero), ("wrapping", wrapping)):
        got = GcAdpcmEncoder.Encode(pcm, coefs)
        for i, name in enumerate(names):
            want = po.gc_encode(pcm[i], coefs[i])
            if label != "wrapping":
                assert not po.gc_last_encode_hit_nontermination(), (label, name)
            assert (got[i] == want).all(), (label, name, int(np.argmax(got[i] != want)))


@pytest.mark.parametrize("n", [1, 2, 13, 14, 15, 27, 28, 29, 141])
def test_encode_ragged_lengths_and_history(n):
    rng = np.random.default_rng(100 + n)
    pcm = rng.integers(-32768, 32768, (5, n)).astype(np.int16)
    coefs = rng.integers(-4096, 4096, (5, 16)).astype(np.int16)
    h1 = rng.integers(-32768, 32768, 5).astype(np.int16)
    h2 = rng.integers(-32768, 32768, 5).astype(np.int16)
    got = GcAdpcmEncoder.Encode(list(pcm), coefs, GcAdpcmParameters(History1=h1, History2=h2))
    for c in range(5):
        want = po.gc_encode(pcm[c], coefs[c], hist1=int(h1[c]), hist2=int(h2[c]))
        assert len(got[c]) == GcAdpcmMath.SampleCountToByteCount(n)
        assert (got[c] == want).all()
    dec = GcAdpcmDecoder.Decode(got, coefs, GcAdpcmParameters(SampleCount=n, History1=h1, History2=h2))
    for c in range(5):
        assert (dec[c] == po.gc_decode(got[c], coefs[c], n, int(h1[c]), int(h2[c]))).all()


def test_sample_count_override_and_errors():
    import vgaudio_amd
    pcm = synth.generate(1, 1000)[0]
    coefs = po.gc_calculate_coefficients(pcm)
    part = GcAdpcmEncoder.Encode(pcm, coefs, GcAdpcmParameters(SampleCount=140))
    assert (part == po.gc_encode(pcm, coefs, sample_count=140)).all()
    with pytest.raises(vgaudio_amd.ArgumentError):
        GcAdpcmEncoder.Encode(pcm, coefs, GcAdpcmParameters(SampleCount=1001))
    bad = po.gc_encode(pcm, coefs).copy()
    bad[8] = 0x90                      # predictor 9: coefficients[18] is out of range in the reference
    with pytest.raises(vgaudio_amd.ArgumentError):
        GcAdpcmDecoder.Decode(bad, coefs, GcAdpcmParameters(SampleCount=1000))


def test_decode_matches_oracle_random_bitstreams():
    rng = np.random.default_rng(5)
    n = 14 * 500 + 6
    nb = GcAdpcmMath.SampleCountToByteCount(n)
    adpcm = rng.integers(0, 256, (70, nb)).astype(np.uint8)
    adpcm[:, 0::8] &= 0x7F             # predictors 0..7
    coefs = rng.integers(-32768, 32768, (70, 16)).astype(np.int16)
    got = GcAdpcmDecoder.Decode(list(adpcm), coefs, GcAdpcmParameters(SampleCount=n))
    for c in range(70):
        assert (got[c] == po.gc_decode(adpcm[c], coefs[c], n)).all(), c


# ---- the reference's own KATs, through the mirrored interface ----
@pytest.mark.parametrize("starts,entries,expected", [
    ([0], 3, [0, 0, 50, 49, 100, 99]),
    ([0, 50], 3, [0, 0, 0, 0, 50, 49, 100, 99, 100, 99, 150, 149]),
    ([0, 50, 200, 100], 3, [0, 0, 0, 0, 0, 0, 0, 0, 50, 49, 100, 99, 250, 249, 150, 149,
                            100, 99, 150, 149, 300, 299, 200, 199]),
    ([0, 50, 200, 100], 2, [0, 0, 0, 0, 0, 0, 0, 0, 50, 49, 100, 99, 250, 249, 150, 149]),
])
@pytest.mark.parametrize("big_endian", [True, False])
def test_build_seek_table_kats(starts, entries, expected, big_endian):
    # Tests/Formats/GcAdpcmFormatTests.cs:92-158
    pcm = [(np.arange(112) + 1 + s).astype(np.int16) for s in starts]
    adpcm = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(pcm, 48000)).WithSamplesPerSeekTableEntry(50)
    table = adpcm.BuildSeekTable(entries, big_endian)
    want = np.array(expected, dtype=np.int16).astype(">i2" if big_endian else "<i2").tobytes()
    assert table == want


def test_format_roundtrip_matches_oracle_batch():
    pcm = synth.generate(9, 14 * 1500 + 4)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    coefs, adpcm = po.gc_encode_batch(pcm, threads=4)
    for c in range(9):
        assert (fmt.Channels[c].Coefs == coefs[c]).all()
        assert (fmt.Channels[c].Adpcm == adpcm[c]).all()
    back = fmt.ToPcm16()
    want = po.gc_decode_batch(adpcm, coefs, pcm.shape[1], threads=4)
    for c in range(9):
        assert (back.Channels[c] == want[c]).all()


def test_silence_and_empty():
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([np.zeros(1000, np.int16)] * 2))
    assert not fmt.Channels[0].Coefs.any() and not fmt.Channels[1].Adpcm.any()
    empty = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([np.zeros(0, np.int16)]))
    assert len(empty.Channels[0].Adpcm) == 0 and not empty.Channels[0].Coefs.any()
    assert GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([])).ChannelCount == 0


def test_dsptool_compatible_exports():
    import ctypes as C
    from vgaudio_amd import _lib
    L = _lib.lib()
    pcm = synth.generate(1, 14 * 400 + 3)[0]
    n = len(pcm)
    coefs = np.zeros(16, np.int16)
    L.correlateCoefs(pcm.ctypes.data_as(_lib.i16p), n, coefs.ctypes.data_as(_lib.i16p))
    want_coefs = po.gc_calculate_coefficients(pcm)
    assert coefs.tolist() == want_coefs.tolist()
    info = _lib.ADPCMINFO()
    out = np.zeros(GcAdpcmMath.SampleCountToByteCount(n), np.uint8)
    L.encode(pcm.ctypes.data_as(_lib.i16p), out.ctypes.data_as(_lib.u8p), C.byref(info), n)
    want = po.gc_encode(pcm, want_coefs)
    assert (out == want).all() and list(info.coef) == want_coefs.tolist() and info.pred_scale == want[0]
    dec = np.zeros(n, np.int16)
    L.decode(out.ctypes.data_as(_lib.u8p), dec.ctypes.data_as(_lib.i16p), C.byref(info), n)
    assert (dec == po.gc_decode(want, want_coefs, n)).all()
    # encodeFrame: in-place reconstruction like DspEncodeFrame
    buf = np.zeros(16, np.int16)
    buf[0], buf[1] = 123, -456
    buf[2:] = pcm[:14]
    frame = np.zeros(8, np.uint8)
    wf, wbuf = po.gc_encode_frame(buf, want_coefs)
    GcAdpcmEncoder.DspEncodeFrame(buf, 14, frame, want_coefs)
    assert (frame == wf).all() and (buf == wbuf).all()


def test_device_resident_path_and_synth_bits():
    import torch
    from vgaudio_amd import device as dev
    d = torch.device("cuda:0")
    nch, n = 40, 14 * 2500 + 7
    pcm = dev.synth_pcm(nch, n, d)
    host = synth.generate(nch, n)
    assert (pcm[:, :n].cpu().numpy() == host).all()
    coefs = dev.gc_coefs(pcm, n)
    adpcm = dev.gc_encode(pcm, n, coefs)
    dec, status = dev.gc_decode(adpcm, coefs, n)
    torch.cuda.synchronize()
    nb = dev.gc_byte_count(n)
    wc, wa = po.gc_encode_batch(host, threads=4)
    assert (coefs.cpu().numpy() == wc).all()
    assert (adpcm[:, :nb].cpu().numpy() == wa).all()
    assert (dec[:, :n].cpu().numpy() == po.gc_decode_batch(wa, wc, n, threads=4)).all()
    assert int(status.item()) == 0


def test_full_size_channels_sampled_against_oracle():
    """BASELINE config 2 shape per channel (48 kHz x 60 s = 2 880 000 samples) on a reduced
    channel count (the full 4096 run is bench.py); 3 channels checked bit-exact against the
    oracle, all checked through the size-independent property decode(encode(x)) ~ x."""
    import torch
    from vgaudio_amd import device as dev
    d = torch.device("cuda:0")
    nch, n = 64, 2_880_000
    pcm = dev.synth_pcm(nch, n, d)
    coefs = dev.gc_coefs(pcm, n)
    adpcm = dev.gc_encode(pcm, n, coefs)
    dec, status = dev.gc_decode(adpcm, coefs, n)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    err = (dec[:, :n].to(torch.int32) - pcm[:, :n].to(torch.int32)).to(torch.float64)
    rms = err.pow(2).mean(dim=1).sqrt()
    sig = pcm[:, :n].to(torch.float64).pow(2).mean(dim=1).sqrt()
    assert (rms < 0.05 * sig).all(), (rms / sig).max().item()
    nb = dev.gc_byte_count(n)
    for c in (0, 31, 63):
        host = pcm[c, :n].cpu().numpy()
        wc = po.gc_calculate_coefficients(host)
        assert coefs[c].cpu().numpy().tolist() == wc.tolist()
        assert (adpcm[c, :nb].cpu().numpy() == po.gc_encode(host, wc)).all()


# ---- channel metadata: alignment re-encode, loop context, seek table ----
@pytest.mark.parametrize("n,loop,alignment,spe", [
    (3000, None, 0, 0),                      # nothing to derive
    (3000, None, 0, 1000),                   # seek table only (container writers)
    (3000, (0, 3000), 0, 0x200),             # loop from 0: default loop context (reference quirk)
    (3000, (100, 2900), 0, 0),               # loop context from the decoded PCM
    (3000, (100, 2900), 1000, 0x200),        # alignment: loop 100..2900 -> 1000..3800
    (20000, (2800, 2990), 0x3800, 0x3800),   # short loop wrapped ~60 times, BRSTM-sized alignment
    (3000, (15, 29), 14, 7),                 # frame-sized pieces
    (20000, (1, 19999), 2, 0),               # one sample of shift: almost everything is kept
    (6000, (4999, 5000), 64, 100),           # one-sample loop
    (14 * 300, (14 * 100, 14 * 300), 14, 14)])  # aligned already
def test_build_channels_matches_oracle(n, loop, alignment, spe):
    from vgaudio_amd.gcadpcm import build_channels
    nch = 5
    pcm = synth.generate(nch, n)
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    looping = loop is not None
    ls, le = loop if looping else (0, 0)
    built = build_channels(fmt.Channels, looping, ls, le, alignment, spe, keepPcm=True)
    p = po.gc_channel_params(n, looping, ls, le, alignment, spe)
    for c in range(nch):
        rc, L, want_adpcm, want_pcm, want_seek, want_ctx = po.gc_build_channel(fmt.Channels[c].Adpcm, fmt.Channels[c].Coefs, p)
        assert rc == 0
        ch = built[c]
        assert ch.SampleCount == L.sample_count_aligned and ch.AlignmentNeeded == bool(L.alignment_needed)
        assert (ch.GetAdpcmAudio() == want_adpcm).all(), c
        assert (ch.GetPcmAudio() == want_pcm).all(), (c, int(np.argmax(ch.GetPcmAudio() != want_pcm)))
        assert (ch.GetSeekTable() == want_seek).all(), c
        assert [ch.LoopContext.PredScale, ch.LoopContext.Hist1, ch.LoopContext.Hist2] == want_ctx.tolist(), c


def test_format_loop_alignment_properties_and_errors():
    from vgaudio_amd import _lib
    pcm = synth.generate(2, 3000)
    p16 = Pcm16Format(list(pcm), 48000).WithLoop(True, 100, 2900)
    fmt = GcAdpcmFormat().EncodeFromPcm16(p16)
    assert (fmt.Looping, fmt.LoopStart, fmt.LoopEnd, fmt.SampleCount) == (True, 100, 2900, 3000)
    al = fmt.WithAlignment(1000)                                       # GcAdpcmFormat.cs:19-22, :124-126
    assert (al.LoopStart, al.LoopEnd, al.SampleCount) == (1000, 3800, 3800)
    assert all(c.SampleCount == 3800 and len(c.GetAdpcmAudio()) == GcAdpcmMath.SampleCountToByteCount(3800) for c in al.Channels)
    back = al.ToPcm16()
    assert (back.Looping, back.LoopStart, back.LoopEnd, back.SampleCount) == (True, 1000, 3800, 3800)
    # the aligned stream decodes to the aligned PCM (GcAdpcmAlignmentTests.AlignedPcmIsCorrect)
    want = GcAdpcmDecoder.Decode([c.GetAdpcmAudio() for c in al.Channels], np.stack([c.Coefs for c in al.Channels]),
                                 GcAdpcmParameters(SampleCount=3800))
    assert all((a == b).all() for a, b in zip(back.Channels, want))
    with pytest.raises(_lib.InvalidOperationError):                    # the reference's fill loop never ends
        fmt.WithLoop(True, 100, 100).WithAlignment(1000)
    with pytest.raises(_lib.ArgumentOutOfRangeError):                  # pred/scale byte read past the original data
        fmt.WithLoop(True, 2999, 3000).WithAlignment(0x3800)
    with pytest.raises(_lib.ArgumentOutOfRangeError):
        fmt.WithLoop(True, 10, 4000)


def test_aligned_sine_kats_on_device():
    """GcAdpcmAlignmentTests.cs:64-108 through the device path (period-56 sine, +-2)."""
    from vgaudio_amd.gcadpcm import build_channels
    for multiple, loop_start, cycles, tol in [(1000, 4524, 100, 2), (1000, 2012, 1, 2), (1000, 60, 1, 2), (1000, 60, 20, 2)]:
        loop_end = cycles * 56 + loop_start
        n = -(-loop_end // 14) * 14
        pcm = synth.sine(n, 1, 56)
        fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([pcm], 48000))
        ch = build_channels(fmt.Channels, True, loop_start, loop_end, multiple, loopContext=False)[0]
        dec = GcAdpcmDecoder.Decode(ch.GetAdpcmAudio(), ch.Coefs, GcAdpcmParameters(SampleCount=ch.SampleCount))
        assert (np.asarray(dec).reshape(-1) == ch.GetPcmAudio()).all()
        want = synth.sine(ch.SampleCount, 1, 56)
        end = -(-ch.SampleCount // 14) * 14 - 14
        assert np.abs(want[56:end].astype(int) - ch.GetPcmAudio()[56:end].astype(int)).max() <= tol


def test_encode_decode_fuzz_mixed_signals():
    """Random concatenations of signal classes at random levels: level jumps between frames exercise the third and
    later quantise passes, the cap at scale 12 and the bump loop far more often than stationary signals do."""
    rng = np.random.default_rng(2024)
    n = 14 * 1500 + 5
    chans = []
    for _ in range(48):
        parts, total = [], 0
        while total < n:
            m = int(rng.integers(7, 400))
            amp = int(rng.choice([1, 3, 40, 700, 9000, 32767]))
            kind = rng.integers(0, 6)
            t = np.arange(m)
            if kind == 0:
                seg = np.zeros(m)
            elif kind == 1:
                seg = rng.integers(-amp, amp + 1, m)
            elif kind == 2:
                seg = amp * np.sin(t * rng.uniform(0.01, 3.1))
            elif kind == 3:
                seg = np.where((t // int(rng.integers(1, 9))) % 2 == 0, amp, -amp - 1)
            elif kind == 4:
                seg = np.cumsum(rng.integers(-amp // 8 - 1, amp // 8 + 2, m))
            else:
                seg = np.full(m, rng.integers(-amp - 1, amp + 1))
            parts.append(np.clip(seg, -32768, 32767))
            total += m
        chans.append(np.concatenate(parts)[:n].astype(np.int16))
    coefs = GcAdpcmCoefficients.CalculateCoefficients(chans)
    got = GcAdpcmEncoder.Encode(chans, coefs)
    dec = GcAdpcmDecoder.Decode(got, coefs, GcAdpcmParameters(SampleCount=n))
    for c in range(len(chans)):
        want_coefs = po.gc_calculate_coefficients(chans[c])
        assert (np.asarray(coefs[c]) == want_coefs).all(), c
        want = po.gc_encode(chans[c], want_coefs)
        assert (got[c] == want).all(), (c, int(np.argmax(got[c] != want)))
        assert (dec[c] == po.gc_decode(want, want_coefs, n)).all(), c


def test_batch_entry_points_are_reentrant_from_many_threads():
    """The reference calls the codec from arbitrary TPL threads (GcAdpcmFormat.cs:65, Cli/Batch.cs:24): the
    host-buffer exports own their stream and device buffers, so concurrent callers must not disturb each other."""
    import threading
    n = 14 * 900 + 3
    jobs = [synth.generate(3, n, first_channel=7 * k) for k in range(6)]
    want = [po.gc_encode_batch(j, threads=2) for j in jobs]
    got = [None] * len(jobs)
    errors = []

    def work(k):
        try:
            for _ in range(3):                                   # a few rounds to make overlap likely
                fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format(list(jobs[k]), 48000))
                back = fmt.ToPcm16()
                got[k] = (fmt, back)
        except Exception as e:                                   # noqa: BLE001 - reported below
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, (fmt, back) in enumerate(got):
        coefs, adpcm = want[k]
        for c in range(3):
            assert (fmt.Channels[c].Coefs == coefs[c]).all(), (k, c)
            assert (fmt.Channels[c].Adpcm == adpcm[c]).all(), (k, c)
            assert (back.Channels[c] == po.gc_decode(adpcm[c], coefs[c], n)).all(), (k, c)


def test_encoder_layouts_and_piece_counts_give_the_same_bytes():
    """The encoder's lane layout (8 = (channel, predictor), the product's; 4 = (channel, predictor, candidate)) and the
    number of time pieces a channel is cut into are performance choices: every combination must produce the oracle's bytes."""
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    d = torch.device("cuda:0")
    nch, n = 37, 14 * 9000 + 5
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    host = pcm[:, :n].cpu().numpy()
    wc, wa = po.gc_encode_batch(host, threads=4)
    nb = vdev.gc_byte_count(n)
    assert np.array_equal(coefs.cpu().numpy().reshape(nch, 16), np.asarray(wc).reshape(nch, 16))
    try:
        for layout in (4, 8):
            for segments in (0, 1, 2, 5, 16):
                L.vga_testing_gc_encoder_layout_this_thread(layout)
                L.vga_testing_gc_encoder_segments_this_thread(segments)
                out = vdev.gc_encode(pcm, n, coefs)
                torch.cuda.synchronize()
                assert np.array_equal(out[:, :nb].cpu().numpy(), np.asarray(wa)[:, :nb]), (layout, segments)
    finally:
        L.vga_testing_gc_encoder_layout_this_thread(0)
        L.vga_testing_gc_encoder_segments_this_thread(0)


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
def test_all_coefficient_kernels_match_the_oracle(variant):
    """0 = the launcher's choice, 1 = one wave per channel, 2 = four channels + a summing wave per workgroup: the ordered
    f64 sums must come out the same in all, on channel counts that fill no workgroup and on the edge inputs."""
    from vgaudio_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(11)
    old = L.vga_testing_gc_coefs_variant_this_thread(variant)
    try:
        for nch, n in ((1, 14 * 700 + 3), (5, 14 * 3000 + 5), (7, 64 * 14), (9, 64 * 14 + 1), (3, 13), (2, 0)):
            pcm = synth.generate(nch, n) if n else np.zeros((nch, 0), dtype=np.int16)
            got = GcAdpcmCoefficients.CalculateCoefficients(list(pcm))
            for c in range(nch):
                assert got[c].tolist() == po.gc_calculate_coefficients(pcm[c]).tolist(), (variant, nch, n, c)
        chans = _edge_channels(14 * 700 + 9, rng)
        got = GcAdpcmCoefficients.CalculateCoefficients(list(chans.values()))
        for i, (name, pcm) in enumerate(chans.items()):
            assert got[i].tolist() == po.gc_calculate_coefficients(pcm).tolist(), (variant, name)
    finally:
        L.vga_testing_gc_coefs_variant_this_thread(old)


@pytest.mark.parametrize("n", [14 * 40000, 14 * 40000 + 9])
def test_short_pieces_leave_seams_open_and_the_chain_closes_them(n):
    """Pieces of ~130 frames: a few percent of the seams are still open when their piece ends (often several in a row in
    one channel), so the chain launch -- and, with a partial last frame, the last-piece repair -- produce part of the
    output.  It is the oracle's, byte for byte."""
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    d = torch.device("cuda:0")
    nch = 45
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    host = pcm[:, :n].cpu().numpy()
    wc, wa = po.gc_encode_batch(host, threads=4)
    nb = vdev.gc_byte_count(n)
    try:
        for segments in (64, 300, 625):
            L.vga_testing_gc_encoder_segments_this_thread(segments)
            for mode in (0, 2):                              # 2: some seams are never accepted as closed
                old = L.vga_testing_force_open_seams_this_thread(mode)
                try:
                    out = vdev.gc_encode(pcm, n, coefs)
                    torch.cuda.synchronize()
                finally:
                    L.vga_testing_force_open_seams_this_thread(old)
                assert np.array_equal(out[:, :nb].cpu().numpy(), np.asarray(wa)[:, :nb]), (segments, mode)
    finally:
        L.vga_testing_gc_encoder_segments_this_thread(0)


def test_decode_into_rows_that_are_only_dword_aligned():
    """include/vgaudio_hip.h promises rows of samples on 4-byte boundaries and an even pitch, no more: the decoder's
    16-byte stores of whole runs are for rows that happen to sit on 16-byte boundaries, every other layout takes its
    dword path -- same samples (several pieces, ragged tail, 70 channels = a wave that is not full)"""
    import torch
    from vgaudio_amd import device as vdev
    d = torch.device("cuda:0")
    nch, n = 70, 14 * 9000 + 5
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    want, st0 = vdev.gc_decode(adpcm, coefs, n)
    host = po.gc_decode_batch(adpcm[:, :vdev.gc_byte_count(n)].cpu().numpy(), coefs.cpu().numpy().reshape(nch, 16), n, threads=4)
    assert np.array_equal(want[:, :n].cpu().numpy(), host) and int(want[:, n:].abs().sum()) == 0
    for pitch_extra, shift in ((2, 0), (0, 2), (6, 6)):                  # pitch % 8 != 0, base % 16 != 0, both
        pitch = (n + 7) // 8 * 8 + 8 + pitch_extra
        buf = torch.zeros(nch * pitch + 16, dtype=torch.int16, device=d)
        out = buf[shift:shift + nch * pitch].view(nch, pitch)
        assert (out.data_ptr() % 16 != 0) or (pitch % 8 != 0)
        got, st = vdev.gc_decode(adpcm, coefs, n, out=out)
        torch.cuda.synchronize()
        assert int(st.item()) == 0 and np.array_equal(got[:, :n].cpu().numpy(), host), (pitch_extra, shift)
        assert int(buf[:shift].abs().sum()) == 0 and int(out[:, n:].abs().sum()) == 0       # nothing outside the rows' samples
