"""Oracle for the GC-ADPCM channel metadata (loop-alignment re-encode, loop
context, seek table) pinned against the reference's own tests:
Tests/Formats/GcAdpcm/GcAdpcmAlignmentTests.cs, GcAdpcmLoopContextTests.cs, GcAdpcmSeekTableTests.cs
(vectors harvested by tests/golden/make_gc_fixtures.py)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import pyoracle as po

FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "gc_loop_context.json")))


def sine(n, freq, rate):
    """GenerateAudio.GenerateSineWave (Tests/GenerateAudio.cs:23-33): (short)(32767 * sin(c*i)), truncation"""
    c = 2 * math.pi * freq / rate
    return np.array([int(32767 * math.sin(c * i)) for i in range(n)], dtype=np.int16)


@pytest.mark.parametrize("loop_start,expected", FX["pred_scale"])
def test_loop_context_pred_scale(loop_start, expected):          # GcAdpcmLoopContextTests.cs:20-30
    ctx = po.gc_loop_context(FX["adpcm"], FX["pcm"], loop_start)
    assert ctx[0] == expected


@pytest.mark.parametrize("loop_start,h1,h2", FX["history"])
def test_loop_context_history(loop_start, h1, h2):              # :32-43
    ctx = po.gc_loop_context(FX["adpcm"], FX["pcm"], loop_start)
    assert (ctx[1], ctx[2]) == (h1, h2)


@pytest.mark.parametrize("multiple,loop_start,loop_end,alen", FX["alignment_not_needed"])
def test_alignment_not_needed(multiple, loop_start, loop_end, alen):        # GcAdpcmAlignmentTests.cs:13-24
    rc, L, _, _ = po.gc_alignment(multiple, loop_start, loop_end, np.zeros(alen, np.uint8), np.zeros(16, np.int16))
    assert rc == 0 and not L.alignment_needed


@pytest.mark.parametrize("multiple,loop_start,loop_end,alen,exp_start,exp_end", FX["aligned_loop_points"])
def test_aligned_loop_points(multiple, loop_start, loop_end, alen, exp_start, exp_end):    # :26-37, :51-62
    rc, L, adpcm, pcm = po.gc_alignment(multiple, loop_start, loop_end, np.zeros(alen, np.uint8), np.zeros(16, np.int16))
    assert rc == 0 and L.alignment_needed
    assert (L.loop_start_aligned, L.sample_count_aligned) == (exp_start, exp_end)
    assert len(pcm) == exp_end and not pcm.any() and not adpcm.any()        # silence in, silence out


@pytest.mark.parametrize("multiple,loop_start,cycles,tol", FX["aligned_sine"])
def test_aligned_adpcm_and_pcm_are_correct(multiple, loop_start, cycles, tol):      # :64-108
    loop_end = cycles * 4 * 14 + loop_start
    n = -(-loop_end // 14) * 14
    pcm = sine(n, 1, 14 * 4)
    coefs = po.gc_calculate_coefficients(pcm)
    adpcm = po.gc_encode(pcm, coefs)
    rc, L, aligned, pcm_aligned = po.gc_alignment(multiple, loop_start, loop_end, adpcm, coefs)
    assert rc == 0 and L.alignment_needed
    dec = po.gc_decode(aligned, coefs, L.sample_count_aligned)
    assert (dec == pcm_aligned).all()                                       # AlignedPcmIsCorrect
    want = sine(L.sample_count_aligned, 1, 14 * 4)
    end = -(-L.sample_count_aligned // 14) * 14 - 14
    d = np.abs(want[56:end].astype(int) - dec[56:end].astype(int))
    assert d.max() <= tol                                                   # AlignedAdpcmIsCorrect


def test_seek_table_kats():                                                 # GcAdpcmSeekTableTests.cs:20-49
    for n, spe, entries in [(1, 50, 1), (100, 50, 2), (101, 50, 3), (10000, 50, 200)]:
        assert len(po.gc_create_seek_table(np.zeros(n, np.int16), spe)) == entries * 2
    # GenerateAscendingShorts(0, 101): pcm[i] = i + 1 (Tests/GenerateAudio.cs:109-117)
    assert po.gc_create_seek_table(np.arange(1, 102, dtype=np.int16), 50).tolist() == [0, 0, 50, 49, 100, 99]


def test_build_channel_semantics():
    """GcAdpcmChannelBuilder paths for a fresh channel (GcAdpcmChannelBuilder.cs:148-202)."""
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal(3000) * 3000).astype(np.int16)
    coefs = po.gc_calculate_coefficients(pcm)
    adpcm = po.gc_encode(pcm, coefs)
    dec = po.gc_decode(adpcm, coefs, 3000)
    # not looping: nothing is derived, loop context is the default (0, 0, 0), no seek table
    rc, L, a, p, seek, ctx = po.gc_build_channel(adpcm, coefs, po.gc_channel_params(3000))
    assert rc == 0 and not L.alignment_needed and (a == adpcm).all() and (p == dec).all()
    assert ctx.tolist() == [0, 0, 0] and len(seek) == 0
    # looping from 0: LoopContextStart == 0 == loop start -> still the default context (reference quirk)
    rc, L, a, p, seek, ctx = po.gc_build_channel(adpcm, coefs, po.gc_channel_params(3000, True, 0, 3000, 0, 1000))
    assert ctx.tolist() == [0, 0, 0] and seek.tolist() == [0, 0, dec[999], dec[998], dec[1999], dec[1998]]
    # looping from 100, no alignment: context from the decoded PCM and the frame header of sample 100
    rc, L, a, p, seek, ctx = po.gc_build_channel(adpcm, coefs, po.gc_channel_params(3000, True, 100, 2900))
    assert ctx.tolist() == [adpcm[100 // 14 * 8], dec[99], dec[98]]
    # alignment to 1000: loop 100..2900 -> 1000..3800, samples past frame 207 re-encoded from the wrapped loop
    rc, L, a, p, seek, ctx = po.gc_build_channel(adpcm, coefs, po.gc_channel_params(3000, True, 100, 2900, 1000, 0x200))
    assert rc == 0 and L.alignment_needed and (L.loop_start_aligned, L.sample_count_aligned) == (1000, 3800)
    keep = 2900 // 14
    assert (a[:keep * 8] == adpcm[:keep * 8]).all() and len(p) == 3800
    assert (po.gc_decode(a, coefs, 3800) == p).all()
    assert ctx.tolist() == [adpcm[1000 // 14 * 8], p[999], p[998]]
    assert L.seek_table_entries == -(-3800 // 0x200) and seek[2] == p[0x200 - 1]
    # degenerate loop that needs alignment: the reference's fill loop would never end
    rc = po.gc_build_channel(adpcm, coefs, po.gc_channel_params(3000, True, 100, 100, 1000))[0]
    assert rc == -4
