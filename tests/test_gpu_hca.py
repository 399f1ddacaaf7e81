"""GPU parity tests for CRI HCA: frames from the HIP encoder and PCM from the HIP decoder must equal
the CPU oracle's byte for byte (the reference arithmetic is f64 in a fixed operation order, which the
kernels reproduce exactly; north_star's 1-ULP IMDCT tolerance is therefore not needed)."""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth
from vgaudio_amd.crihca import CriHcaDecoder, CriHcaEncoder, CriHcaFormat, CriHcaParameters, CriHcaQuality
from vgaudio_amd.gcadpcm import Pcm16Format

pytestmark = pytest.mark.gpu
Q = {"Highest": 1, "High": 2, "Middle": 3, "Low": 4, "Lowest": 5}


def _streams(ns, nch, n, kind="synth"):
    rng = np.random.default_rng(ns * 1000 + nch * 10 + n)
    out = []
    for s in range(ns):
        if kind == "synth":
            out.append(synth.generate(nch, n, first_channel=nch * s))
        elif kind == "noise":
            out.append(rng.integers(-32768, 32768, (nch, n)).astype(np.int16))
        elif kind == "quiet":
            out.append(rng.integers(-3, 4, (nch, n)).astype(np.int16))
        elif kind == "silence":
            out.append(np.zeros((nch, n), np.int16))
        elif kind == "square":
            t = np.arange(n)
            out.append(np.tile(np.where((t // 9) % 2 == 0, 32767, -32768).astype(np.int16), (nch, 1)))
    return out


def test_initialize_matches_oracle():
    for nch, q, n, br, lim in [(2, "High", 2_880_000, 0, False), (1, "Lowest", 1000, 0, True), (6, "Low", 99999, 0, False),
                               (2, "Middle", 5000, 96000, False), (8, "Highest", 4096, 0, False), (3, "Lowest", 777, 0, False)]:
        enc = CriHcaEncoder.InitializeNew(CriHcaParameters(Quality=Q[q], Bitrate=br, LimitBitrate=lim, ChannelCount=nch,
                                                           SampleRate=48000, SampleCount=n))
        rc, info = po.hca_init(po.hca_params(nch, n, quality=q, bitrate=br, limit_bitrate=lim))
        assert rc == 0
        for k, v in info.as_dict().items():
            assert getattr(enc.Hca.c, k) == v, (k, nch, q)
    enc = CriHcaEncoder.InitializeNew(CriHcaParameters(ChannelCount=2, SampleRate=48000, SampleCount=20000, Looping=True,
                                                       LoopStart=3000, LoopEnd=18000))
    rc, info = po.hca_init(po.hca_params(2, 20000, looping=True, loop_start=3000, loop_end=18000))
    assert all(getattr(enc.Hca.c, k) == v for k, v in info.as_dict().items())


@pytest.mark.parametrize("nch,quality,n,kind", [
    (2, "High", 48000, "synth"), (1, "High", 10000, "synth"), (2, "Highest", 20000, "synth"),
    (2, "Middle", 30000, "synth"), (2, "Low", 48000, "synth"), (2, "Lowest", 48000, "synth"),
    (1, "Lowest", 30000, "synth"), (4, "Middle", 12345, "synth"), (6, "Low", 9000, "synth"), (8, "High", 5000, "synth"),
    (3, "Lowest", 7000, "synth"), (5, "Low", 6000, "synth"),
    (2, "High", 9000, "noise"), (2, "Lowest", 9000, "noise"), (2, "High", 9000, "quiet"), (2, "High", 5000, "silence"),
    (2, "Low", 9000, "square"), (2, "High", 1, "synth"), (2, "High", 896, "synth"), (2, "High", 897, "synth"),
    (2, "High", 1024, "synth"), (1, "High", 2047, "synth")])
def test_encode_and_decode_match_oracle(nch, quality, n, kind):
    streams = _streams(3, nch, n, kind)
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(s), 48000) for s in streams],
                                             CriHcaParameters(Quality=Q[quality]))
    p = po.hca_params(nch, n, quality=quality)
    for s, fmt in zip(streams, fmts):
        rc, info, want = po.hca_encode(s, p)
        assert rc == 0
        assert fmt.AudioData.shape == want.shape
        bad = np.argwhere(fmt.AudioData != want)
        assert bad.size == 0, (bad[0].tolist(), len(bad))
    # decoder parity on the oracle's frames
    rc, info, frames = po.hca_encode(streams[0], p)
    dec = CriHcaDecoder.Decode(fmts[0].Hca, [fmt.AudioData for fmt in fmts])
    for s, fmt, d in zip(streams, fmts, dec):
        rc, want = po.hca_decode(info, fmt.AudioData)
        assert rc == 0
        for c in range(nch):
            assert (d[c] == want[c]).all(), (c, int(np.argmax(d[c] != want[c])))


def test_decode_rejects_bad_sync_and_encode_rejects_low_bitrate():
    import vgaudio_amd
    s = synth.generate(2, 8192)
    fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(s), 48000))
    bad = fmt.AudioData.copy()
    bad[2, 0] = 0
    with pytest.raises(vgaudio_amd.InvalidDataError):
        CriHcaDecoder.Decode(fmt.Hca, bad)
    with pytest.raises(vgaudio_amd.InvalidDataError):
        CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(s), 48000), CriHcaParameters(Bitrate=2000))
    with pytest.raises(vgaudio_amd.ArgumentOutOfRangeError):
        CriHcaEncoder.InitializeNew(CriHcaParameters(ChannelCount=9, SampleRate=48000, SampleCount=100))


def test_config4_shape_reduced_roundtrip():
    """BASELINE configs[3]: stereo streams, quality=high (16 streams x 20 s here; bench_hca.py runs 1024 x 60 s)."""
    streams = _streams(16, 2, 48000 * 20)
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(s), 48000) for s in streams], CriHcaParameters())
    assert fmts[0].Hca.FrameSize == 682
    p = po.hca_params(2, 48000 * 20)
    for i in (0, 7, 15):
        rc, info, want = po.hca_encode(streams[i], p)
        assert (fmts[i].AudioData == want).all()
    dec = CriHcaDecoder.Decode(fmts[0].Hca, [f.AudioData for f in fmts])
    for i in range(16):
        err = np.stack(dec[i]).astype(float) - streams[i]
        snr = 10 * np.log10((streams[i].astype(float) ** 2).mean() / (err ** 2).mean())
        assert snr > 30, snr


@pytest.mark.parametrize("nch,quality,n,loop_start,loop_end", [
    (2, "High", 20000, 3000, 18000),      # plain loop
    (2, "High", 20000, 0, 20000),         # whole stream
    (1, "Middle", 9000, 1, 8999),         # loop start just after a frame boundary: > 1024 samples of pre-audio
    (2, "Low", 5000, 4700, 4990),         # short loop at the end: replayed audio crosses the last chunk
    (2, "High", 5000, 1024, 6000),        # loop end past the data: HcaInfo.SampleCount = min(loop end, n)
    (2, "High", 3000, 2990, 3000),        # ten-sample loop, replay reads past the PCM (stale chunk tail)
    (1, "Lowest", 1500, 1030, 1100),      # loop inside the second chunk
    (4, "Middle", 12345, 5000, 12000)])
def test_looping_encode_and_decode_match_oracle(nch, quality, n, loop_start, loop_end):
    """CriHcaEncoder's input stream for looping files (pre-audio, replayed loop audio, CriHcaEncoder.cs:170-254)."""
    streams = _streams(2, nch, n, "synth")
    pcms = [Pcm16Format(list(s), 48000) for s in streams]
    for pcm in pcms:    # set directly: WithLoop rejects a loop end past the data, CriHcaEncoder clamps it (:84)
        pcm.Looping, pcm.LoopStart, pcm.LoopEnd = True, loop_start, loop_end
    fmts = CriHcaFormat.EncodeBatchFromPcm16(pcms, CriHcaParameters(Quality=Q[quality]))
    p = po.hca_params(nch, n, quality=quality, looping=True, loop_start=loop_start, loop_end=loop_end)
    for s, fmt in zip(streams, fmts):
        rc, info, want = po.hca_encode(s, p)
        assert rc == 0
        for k, v in info.as_dict().items():
            assert getattr(fmt.Hca.c, k) == v, k
        assert fmt.AudioData.shape == want.shape
        bad = np.argwhere(fmt.AudioData != want)
        assert bad.size == 0, (bad[0].tolist(), len(bad))
    dec = CriHcaDecoder.Decode(fmts[0].Hca, [fmt.AudioData for fmt in fmts])
    for fmt, d in zip(fmts, dec):
        rc, want = po.hca_decode(info, fmt.AudioData)
        assert rc == 0
        for c in range(nch):
            assert (d[c] == want[c]).all(), (c, int(np.argmax(d[c] != want[c])))


def test_randomised_configurations_match_oracle():
    """Channel counts, qualities, sample rates and explicit bitrates drawn at random: the derived band layout
    (base / stereo / HFR bands, frame size) differs in nearly every draw, so every packing path gets input."""
    rng = np.random.default_rng(2024)
    kinds = ["synth", "noise", "quiet", "square"]
    done = 0
    while done < 24:
        nch = int(rng.integers(1, 9))
        quality = ["Highest", "High", "Middle", "Low", "Lowest"][int(rng.integers(0, 5))]
        rate = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000, 96000]))
        n = int(rng.integers(1, 30000))
        bitrate = int(rng.choice([0, 0, 32000 * nch, 64000 * nch, 100000 * nch]))
        limit = bool(rng.integers(0, 2))
        p = po.hca_params(nch, n, sample_rate=rate, quality=quality, bitrate=bitrate, limit_bitrate=limit)
        rc, info = po.hca_init(p)
        if rc != 0:
            continue                                  # a combination the reference refuses as well
        s = _streams(1, nch, n, kinds[int(rng.integers(0, 4))])[0]
        rc, info, want = po.hca_encode(s, p)
        cfg = CriHcaParameters(Quality=Q[quality], Bitrate=bitrate, LimitBitrate=limit)
        if rc != 0:
            with pytest.raises(Exception):
                CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(s), rate), cfg)
            done += 1
            continue
        fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(s), rate), cfg)
        bad = np.argwhere(np.asarray(fmt.AudioData) != want)
        assert bad.size == 0, (nch, quality, rate, n, bitrate, limit, bad[0].tolist(), len(bad))
        dec = CriHcaDecoder.Decode(fmt.Hca, [fmt.AudioData])[0]
        rc, wdec = po.hca_decode(info, want)
        assert rc == 0
        for c in range(nch):
            assert (dec[c] == wdec[c]).all(), (nch, quality, rate, n, c)
        done += 1


def _device_info_words(info):
    """channel types [11:19] and coded band counts [19:27] as the library derives them (host code)"""
    import ctypes as C
    from vgaudio_amd import _lib
    out = (C.c_uint8 * 240)()
    pinfo = _lib.HcaInfoC()
    for name, _ in _lib.HcaInfoC._fields_:
        setattr(pinfo, name, getattr(info, name))
    _lib.check(_lib.lib().vga_testing_hca_device_info(C.byref(pinfo), out, 240))
    return np.frombuffer(bytes(out), np.int32, 27)


def test_decode_of_garbage_frames_reads_zeros_past_the_end_like_the_reference():
    """Random bits behind a valid header (raw 6-bit scale factors, so that delta decoding cannot fail) make the decoder
    walk far past the frame's end, where BitReader.PeekInt yields zeros (BitReader.cs:55-61); the scan's chunk offsets
    then point past the frame.  PCM must still equal the oracle's, for every frame alignment."""
    rng = np.random.default_rng(11)
    n = 1024 * 40 + 300
    for nch, quality, bitrate in ((2, "High", 0), (1, "Lowest", 0), (2, "Lowest", 0), (2, "High", 48047), (4, "Middle", 0)):
        pcm = synth.generate(nch, n)
        rc, info, frames = po.hca_encode(pcm, po.hca_params(nch, n, quality=quality, bitrate=bitrate))
        assert rc == 0
        words = _device_info_words(info)
        fr = np.array(frames, np.uint8).reshape(info.frame_count, info.frame_size)
        for f in range(1, info.frame_count, 2):
            bits = rng.integers(0, 2, info.frame_size * 8).astype(np.uint8)
            bits[:16] = 1
            pos = 32
            for c in range(nch):
                bits[pos:pos + 3] = (1, 1, int(rng.integers(0, 2)))
                pos += 3 + 6 * int(words[19 + c])
                pos += 32 if words[11 + c] == 2 else 6 * info.hfr_group_count
            if f % 3 == 0:
                bits[info.frame_size * 4:] = 1
            fr[f] = np.packbits(bits)
        rc, want = po.hca_decode(info, fr.reshape(-1))
        assert rc == 0
        fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000), CriHcaParameters(Quality=Q[quality], Bitrate=bitrate))
        got = CriHcaDecoder.Decode(fmt.Hca, [fr, fr[::1].copy()])
        for d in got:
            assert np.array_equal(np.stack(d), np.asarray(want).reshape(nch, n)), (nch, quality, bitrate)


def test_long_streams_carry_the_overlap_across_frame_runs():
    """Many streams x many frames so that the decoder's workgroups take runs of 16 frames (the overlap is carried inside a
    run and recomputed at its start); every stream against the oracle."""
    from vgaudio_amd import _lib
    ns, n = 12, 1024 * 61 + 77
    streams = _streams(ns, 2, n)
    fmts = CriHcaFormat.EncodeBatchFromPcm16([Pcm16Format(list(s), 48000) for s in streams], CriHcaParameters())
    rc, info = po.hca_init(po.hca_params(2, n))
    rc, want = po.hca_decode_batch(info, np.stack([fmt.AudioData.reshape(-1) for fmt in fmts]), threads=8)
    assert rc == 0
    for run in (16, 1, 2, 3, 7, 64):
        old = _lib.lib().vga_testing_hca_frames_per_group_this_thread(run)
        try:
            dec = CriHcaDecoder.Decode(fmts[0].Hca, [fmt.AudioData for fmt in fmts])
        finally:
            _lib.lib().vga_testing_hca_frames_per_group_this_thread(old)
        for k in range(ns):
            assert np.array_equal(np.stack(dec[k]), want[k]), (run, k)


def _feed(enc, stream, n, nch, frame_size, garbage_tail):
    """drives CriHcaEncoder.Encode the way CriHcaFormat.EncodeFromPcm16 does (CriHcaFormat.cs:50-68): 1024-sample blocks from
    ONE reused buffer (a block the PCM does not fill keeps the previous block's tail) -- or, garbage_tail, with a marker there"""
    frames, per_call = [], []
    buf = np.zeros((nch, 1024), dtype=np.int16)
    out = np.zeros(frame_size, dtype=np.uint8)
    pos = 0
    while enc.FramesProcessed < enc.Hca.FrameCount:
        take = max(0, min(1024, n - pos))
        if garbage_tail:
            buf[:] = 12345
        buf[:, :take] = stream[:, pos:pos + take]
        pos += 1024
        got = enc.Encode(list(buf), out)
        per_call.append(got)
        if got:
            frames.append(out.copy())
            assert enc.PendingFrameCount == got - 1
            while enc.PendingFrameCount:
                frames.append(enc.GetPendingFrame())
    return np.stack(frames), per_call


@pytest.mark.parametrize("nch,quality,n,loop", [
    (2, "High", 20000, None), (1, "Middle", 1024 * 5, None), (2, "High", 1024 * 7 - 128, None), (2, "Low", 300, None),
    (2, "High", 20000, (3000, 18000)), (1, "Middle", 9000, (1, 8999)), (2, "High", 3000, (2990, 3000)), (2, "Low", 5000, (4700, 4990))])
def test_streaming_encoder_object_matches_the_batch_encoder_and_the_reference_call_pattern(nch, quality, n, loop):
    """CriHcaEncoder.Encode / GetPendingFrame (CriHcaEncoder.cs:126-163): the frames a host receives 1024 samples at a time are
    the batch encoder's (the oracle's), and every call reports the number of frames the reference's counters give it -- checked
    against the independent Python restatement of the streaming shell (oracle/pyref/crihca.py, counters only)."""
    from vgaudio_amd import _lib
    stream = _streams(1, nch, n, "synth")[0]
    kw = dict(looping=True, loop_start=loop[0], loop_end=loop[1]) if loop else {}
    rc, info, want = po.hca_encode(stream, po.hca_params(nch, n, quality=quality, **kw))
    assert rc == 0
    cfg = CriHcaParameters(Quality=Q[quality], ChannelCount=nch, SampleRate=48000, SampleCount=n,
                           Looping=bool(loop), LoopStart=loop[0] if loop else 0, LoopEnd=loop[1] if loop else 0)
    enc = CriHcaEncoder.InitializeNew(cfg)
    assert enc.Hca.FrameCount == want.shape[0] and enc.FrameSize == want.shape[1]
    got, per_call = _feed(enc, stream, n, nch, enc.FrameSize, garbage_tail=False)
    assert got.shape == want.shape
    bad = np.argwhere(got != want)
    assert bad.size == 0, (bad[0].tolist(), len(bad))
    # the reference's counters, restated independently: frames per Encode call
    from oracle.pyref import crihca as ref
    r = ref.Encoder(ref.Params(nch, 48000, n, quality=quality, **kw))
    r.encode_frame = lambda pcm: b""                             # counters only: no frame is computed
    want_calls = []
    while r.frames_processed < r.hca.frame_count:
        want_calls.append(len(r.encode([[0] * 1024 for _ in range(nch)])))
    assert per_call == want_calls
    with pytest.raises(_lib.InvalidOperationError):
        enc.Encode([np.zeros(1024, np.int16)] * nch, np.zeros(enc.FrameSize, np.uint8))
    with pytest.raises(_lib.InvalidOperationError):
        enc.GetPendingFrame()
    enc.close()
    if not loop:
        # what lies behind the stream's last sample in the last block is never read when nothing loops
        enc2 = CriHcaEncoder.InitializeNew(cfg)
        got2, _ = _feed(enc2, stream, n, nch, enc2.FrameSize, garbage_tail=True)
        assert np.array_equal(got2, want)
        enc2.close()
