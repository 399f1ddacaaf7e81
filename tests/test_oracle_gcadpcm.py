"""Pins the CPU oracle (oracle/gcadpcm_oracle.c) against the reference's own
known-answer tests for GC-ADPCM, plus hand-derivable vectors.

Reference tests restated here:
  Tests/Formats/GcAdpcm/GcAdpcmHelpersTests.cs:8-99   (size math KATs)
  Tests/Formats/GcAdpcmFormatTests.cs:92-158          (ascending ramp -> seek table)
  Tests/Formats/GcAdpcm/GcAdpcmAlignmentTests.cs:69-91 (56-sample sine within +-2)
  Tests/GenerateAudio.cs:73-88                        (silence <-> zero coefs / bytes)
"""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth


# ---- GcAdpcmHelpersTests.cs ----
@pytest.mark.parametrize("nibble,expected", [(2, 0), (3, 1), (15, 13), (18, 14), (19, 15), (100010, 87508)])
def test_nibble_to_sample(nibble, expected):
    assert po.lib().vgo_gc_nibble_to_sample(nibble) == expected


@pytest.mark.parametrize("sample,expected", [(0, 2), (1, 3), (13, 15), (14, 18), (15, 19), (87508, 100010)])
def test_sample_to_nibble(sample, expected):
    assert po.lib().vgo_gc_sample_to_nibble(sample) == expected


@pytest.mark.parametrize("n,expected", [(0, 0), (1, 0), (2, 0), (3, 1), (15, 13), (16, 14), (17, 14), (18, 14),
                                        (19, 15), (100000, 87500)])
def test_nibble_count_to_sample_count(n, expected):
    assert po.lib().vgo_gc_nibble_count_to_sample_count(n) == expected


@pytest.mark.parametrize("n,expected", [(0, 0), (1, 3), (2, 4), (13, 15), (14, 16), (15, 19), (87500, 100000)])
def test_sample_count_to_nibble_count(n, expected):
    assert po.lib().vgo_gc_sample_count_to_nibble_count(n) == expected


@pytest.mark.parametrize("n,expected", [(0, 0), (1, 2), (2, 2), (3, 3), (13, 8), (14, 8), (15, 10), (87500, 50000)])
def test_sample_count_to_byte_count(n, expected):
    assert po.lib().vgo_gc_sample_count_to_byte_count(n) == expected


def test_sample_nibble_reversible():
    L = po.lib()
    for i in range(1, 10000):
        assert L.vgo_gc_nibble_to_sample(L.vgo_gc_sample_to_nibble(i)) == i
        assert L.vgo_gc_nibble_count_to_sample_count(L.vgo_gc_sample_count_to_nibble_count(i)) == i


# ---- GcAdpcmFormatTests.cs:92-158  BuildSeekTable* ----
def _ascending(start, count):
    # Tests/GenerateAudio.cs:109-117
    return (np.arange(count) + 1 + start).astype(np.int16)


@pytest.mark.parametrize("start,expected", [
    (0, [0, 0, 50, 49, 100, 99]),
    (50, [0, 0, 100, 99, 150, 149]),
    (200, [0, 0, 250, 249, 300, 299]),
    (100, [0, 0, 150, 149, 200, 199]),
])
def test_ramp_encode_decode_seek_table(start, expected):
    pcm = _ascending(start, 112)
    coefs = po.gc_calculate_coefficients(pcm)          # EncodeChannel, GcAdpcmFormat.cs:129-135
    adpcm = po.gc_encode(pcm, coefs)
    assert len(adpcm) == 64
    decoded = po.gc_decode(adpcm, coefs, 112)          # GcAdpcmChannelBuilder.cs:202
    table = po.gc_create_seek_table(decoded, 50)       # GcAdpcmSeekTable.cs:25-38
    assert table.tolist() == expected


# ---- silence: GenerateAudio.cs:73-88 assumes zero ADPCM + zero coefs ----
@pytest.mark.parametrize("n", [1, 13, 14, 15, 140, 1000])
def test_silence_gives_zero(n):
    pcm = np.zeros(n, dtype=np.int16)
    coefs = po.gc_calculate_coefficients(pcm)
    assert not coefs.any()
    adpcm = po.gc_encode(pcm, coefs)
    assert len(adpcm) == po.gc_sample_count_to_byte_count(n)
    assert not adpcm.any()
    assert not po.gc_decode(adpcm, coefs, n).any()


def test_empty_input():
    coefs = po.gc_calculate_coefficients(np.zeros(0, dtype=np.int16))
    assert not coefs.any()
    assert len(po.gc_encode(np.zeros(0, dtype=np.int16), coefs)) == 0


# ---- GcAdpcmAlignmentTests.cs:69-91 (period-56 sine, +-2 after the first cycle) ----
@pytest.mark.parametrize("cycles", [1, 20, 100])
def test_sine56_roundtrip_tolerance(cycles):
    n = cycles * 56 + 56
    pcm = synth.sine(n, 1, 56)
    coefs = po.gc_calculate_coefficients(pcm)
    adpcm = po.gc_encode(pcm, coefs)
    dec = po.gc_decode(adpcm, coefs, n)
    diff = np.abs(dec[56:n - 14].astype(int) - pcm[56:n - 14].astype(int))
    assert diff.max() <= 2


# ---- encoder reconstruction == decoder output (GcAdpcmEncoder.cs:156-160 vs GcAdpcmDecoder.cs:40-44) ----
def test_encoder_reconstruction_equals_decoder():
    pcm = synth.generate(3, 14 * 300)
    for c in range(3):
        coefs = po.gc_calculate_coefficients(pcm[c])
        adpcm = po.gc_encode(pcm[c], coefs)
        dec = po.gc_decode(adpcm, coefs, pcm.shape[1])
        buf = np.zeros(16, dtype=np.int16)
        for f in range(300):
            buf[2:] = pcm[c, f * 14:(f + 1) * 14]
            frame, buf = po.gc_encode_frame(buf, coefs)
            assert (frame == adpcm[f * 8:(f + 1) * 8]).all()
            assert (buf[2:] == dec[f * 14:(f + 1) * 14]).all()
            buf[0], buf[1] = buf[14], buf[15]


def test_partial_last_frame_and_history():
    pcm = synth.generate(1, 1000)[0]
    coefs = po.gc_calculate_coefficients(pcm)
    full = po.gc_encode(pcm, coefs)
    assert len(full) == po.gc_sample_count_to_byte_count(1000)  # 71 frames + 6 samples -> 568 + 4
    # SampleCount override (GcAdpcmEncoder.cs:17): a prefix encode equals the prefix of the bytes
    part = po.gc_encode(pcm, coefs, sample_count=140)
    assert (part == full[:80]).all()
    # restarting mid-stream with the decoder's history reproduces the tail (GcAdpcmAlignment.cs:54-58)
    dec = po.gc_decode(full, coefs, 1000)
    tail = po.gc_encode(pcm[140:], coefs, hist1=int(dec[139]), hist2=int(dec[138]))
    assert (tail == full[80:]).all()
    with pytest.raises(ValueError):
        po.gc_encode(pcm, coefs, sample_count=1001)


def test_coefs_are_deterministic_and_nontrivial():
    pcm = synth.generate(2, 48000)
    c0 = po.gc_calculate_coefficients(pcm[0])
    c1 = po.gc_calculate_coefficients(pcm[1])
    assert c0.any() and c1.any() and not (c0 == c1).all()
    assert (c0 == po.gc_calculate_coefficients(pcm[0])).all()
    # 8 distinct predictor pairs for rich material
    assert len({(int(c0[2 * i]), int(c0[2 * i + 1])) for i in range(8)}) >= 4


def test_batch_matches_single():
    pcm = synth.generate(5, 14 * 57 + 5)
    coefs, adpcm = po.gc_encode_batch(pcm, threads=3)
    for c in range(5):
        cc = po.gc_calculate_coefficients(pcm[c])
        assert (cc == coefs[c]).all()
        assert (po.gc_encode(pcm[c], cc) == adpcm[c]).all()
    dec = po.gc_decode_batch(adpcm, coefs, pcm.shape[1], threads=2)
    for c in range(5):
        assert (po.gc_decode(adpcm[c], coefs[c], pcm.shape[1]) == dec[c]).all()


def test_reference_nontermination_hazard_is_guarded_and_flagged():
    """With coefficients large enough to wrap the int32 predictor the reference's retry loop
    (GcAdpcmEncoder.cs:127-170) can repeat the scalePower-12 pass forever; the oracle stops and
    flags it.  Coefficients from CalculateCoefficients never get there."""
    n = 14 * 300 + 3
    pcm = np.full(n, 32767, dtype=np.int16)
    coefs = np.full(16, -32768, dtype=np.int16)
    out = po.gc_encode(pcm, coefs)                    # returns (the C# reference would spin)
    assert po.gc_last_encode_hit_nontermination()
    assert len(out) == po.gc_sample_count_to_byte_count(n)
    po.gc_encode(pcm, po.gc_calculate_coefficients(pcm))
    assert not po.gc_last_encode_hit_nontermination()
    rng = np.random.default_rng(3)
    noise = rng.integers(-32768, 32768, n).astype(np.int16)
    for _ in range(20):                               # |c| <= 16383: the predictor cannot wrap
        po.gc_encode(noise, rng.integers(-16383, 16384, 16).astype(np.int16))
        assert not po.gc_last_encode_hit_nontermination()


def test_c_generator_of_the_bench_pcm_equals_its_definition():
    """oracle/synth_oracle.c (fast enough to generate all of configs[4] for the whole-shard digests) against
    vgaudio_amd/synth.py, the generator's definition: lengths around the noise window, high channel numbers"""
    from vgaudio_amd import synth
    for first, nch, n in ((0, 3, 5000), (95, 2, 14 * 300 + 5), (4095, 2, 777), (32767, 1, 20000), (1 << 20, 2, 3), (7, 1, 0)):
        want = synth.generate(nch, n, first_channel=first)
        got = po.synth_generate(nch, n, first_channel=first, threads=2)
        assert np.array_equal(want, got), (first, nch, n)


def test_whole_shard_oracle_digests_are_the_oracles():
    """tests/golden/gc_shard_oracle_digests.json + gc_channel_oracle_digests.npy: written by the oracle for all 32 768
    channels of configs[4] (95 core-minutes); here two channels are encoded again and must reproduce their digests, and the
    file must agree with what a GPU wrote in round 3 (tests/golden/gc_shard_digests.json)"""
    import json
    import os
    import torch
    from vgaudio_amd import distributed as vdist
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    gold = json.load(open(os.path.join(gold_dir, "gc_shard_oracle_digests.json")))
    assert gold["provenance"] == "oracle" and gold["channels_per_shard"] == 4096 and len(gold["shards"]) == 8
    gpu = json.load(open(os.path.join(gold_dir, "gc_shard_digests.json")))
    assert gold["shards"] == gpu["shards"] and gold["bytes_per_row"] == gpu["bytes_per_row"]
    per_channel = np.load(os.path.join(gold_dir, "gc_channel_oracle_digests.npy"))
    assert per_channel.shape == (32768,) and per_channel.dtype == np.uint64
    n, nb = gold["samples_per_channel"], gold["bytes_per_row"]
    for ch in (0, 20000):
        pcm = po.synth_generate(1, n, first_channel=ch)
        coefs, adpcm = po.gc_encode_batch(pcm)
        rows = torch.zeros((1, (nb + 15) // 16 * 16), dtype=torch.uint8)
        rows[0, :nb] = torch.from_numpy(np.ascontiguousarray(adpcm[0]))
        inner = vdist.row_digests(rows, nb, torch.from_numpy(np.asarray(coefs).reshape(1, 16)))
        assert int(inner.numpy().view(np.uint64)[0]) == int(per_channel[ch]), ch
