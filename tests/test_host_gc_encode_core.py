"""CPU-side checks of the exact-arithmetic shortcuts the GC-ADPCM encode kernel uses
(vgaudio_amd/csrc/gc_encode_core.hpp): the header is compiled for the host together with a
lane emulator of the kernel's per-frame control flow and compared with the oracle.
This is host logic under test, not a CPU product path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "gc_encode_emulator.cpp")
HDR = os.path.join(HERE, "..", "vgaudio_amd", "csrc", "gc_encode_core.hpp")
SO = os.path.join(HERE, "host", "libgc_encode_emulator.so")


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fwrapv", "-ffp-contract=off",
                        "-fno-fast-math", SRC, "-o", SO], check=True)
    L = C.CDLL(SO)
    L.emu_encode.argtypes = [C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16), C.c_int16, C.c_int16,
                             C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)]
    L.emu_compare_pass.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int]
    L.emu_compare_pass_wide.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int]
    return L


def _emu_encode(L, pcm, coefs, h1=0, h2=0):
    pcm = np.ascontiguousarray(pcm, np.int16)
    coefs = np.ascontiguousarray(coefs, np.int16)
    out = np.zeros(po.gc_sample_count_to_byte_count(len(pcm)), np.uint8)
    stats = np.zeros(8, np.uint64)
    L.emu_encode(pcm.ctypes.data_as(C.POINTER(C.c_int16)), len(pcm), coefs.ctypes.data_as(C.POINTER(C.c_int16)),
                 h1, h2, out.ctypes.data_as(C.POINTER(C.c_uint8)), stats.ctypes.data_as(C.POINTER(C.c_uint64)))
    return out, stats


def test_halvings_closed_form_exhaustive(emu):
    assert emu.emu_check_halvings() == 0


@pytest.mark.parametrize("cls", ["synthetic", "sine440", "white_full_scale", "noise_3lsb", "silence", "clipped_square", "slow_channel_93"])
def test_frame_resolution_of_the_channel_predictor_layout_on_every_signal_class(emu, cls):
    """the kernel's round-5 control flow (per-lane bump / hostile / inexact-sum / third-trip handling, saturating argmin keys)
    restated on the host against the oracle, on the signal classes that exercise each branch"""
    from vgaudio_amd import signals
    emu.emu_encode8.argtypes = emu.emu_encode.argtypes
    n = 14 * 3000 + 5
    seen = np.zeros(8, np.uint64)
    for ch in (0, 17, 93):
        pcm = (synth.generate(1, n, first_channel=ch) if cls == "synthetic" else signals.host(cls, 1, n, first_channel=ch))[0]
        coefs = po.gc_calculate_coefficients(pcm)
        want = po.gc_encode(pcm, coefs)
        pcm = np.ascontiguousarray(pcm, np.int16)
        co = np.ascontiguousarray(coefs, np.int16)
        out = np.zeros(po.gc_sample_count_to_byte_count(len(pcm)), np.uint8)
        stats = np.zeros(8, np.uint64)
        emu.emu_encode8(pcm.ctypes.data_as(C.POINTER(C.c_int16)), len(pcm), co.ctypes.data_as(C.POINTER(C.c_int16)), 0, 0,
                        out.ctypes.data_as(C.POINTER(C.c_uint8)), stats.ctypes.data_as(C.POINTER(C.c_uint64)))
        bad = np.flatnonzero(out != want)
        assert bad.size == 0, (cls, ch, "first differing byte", int(bad[0]), stats.tolist())
        seen += stats
    if cls == "sine440":
        assert seen[1] > 0            # the bump loop is entered
    if cls in ("white_full_scale", "clipped_square"):
        assert seen[2] > 0            # final passes at the cap with an overflow above 3
    if cls == "synthetic":
        assert seen[3] > 0            # third trips
    if cls in ("synthetic", "noise_3lsb", "slow_channel_93"):
        assert seen[5] > 0            # frames encoded without the f32 detour
    if cls in ("white_full_scale", "clipped_square"):
        assert seen[5] < seen[0]      # ... and frames that are not


def test_hostile_coefficients_and_rail_to_rail_frames_through_the_same_resolution(emu):
    emu.emu_encode8.argtypes = emu.emu_encode.argtypes
    rng = np.random.default_rng(11)
    seen = np.zeros(8, np.uint64)
    for trial in range(12):
        n = 14 * 400 + int(rng.integers(0, 14))
        pcm = (rng.integers(-32768, 32768, n) if trial % 2 else np.where(rng.integers(0, 2, n) > 0, 32767, -32768)).astype(np.int16)
        coefs = rng.integers(-32768, 32768, 16).astype(np.int16) if trial % 3 == 0 else rng.integers(-6000, 6000, 16).astype(np.int16)
        want = po.gc_encode(pcm, coefs)
        out = np.zeros(po.gc_sample_count_to_byte_count(n), np.uint8)
        stats = np.zeros(8, np.uint64)
        emu.emu_encode8(pcm.ctypes.data_as(C.POINTER(C.c_int16)), n, coefs.ctypes.data_as(C.POINTER(C.c_int16)), 0, 0,
                        out.ctypes.data_as(C.POINTER(C.c_uint8)), stats.ctypes.data_as(C.POINTER(C.c_uint64)))
        bad = np.flatnonzero(out != want)
        assert bad.size == 0, (trial, int(bad[0]), stats.tolist())
        seen += stats
    assert seen[1] > 0 and seen[2] > 0 and seen[4] > 0, seen.tolist()


def test_pass_without_the_f32_detour_is_the_exact_pass_whenever_its_overflow_bound_holds(emu):
    """NO_ROUND (gc_encode_core.hpp): (int)(float)d is d below 2^24, a larger distance shows in the pass's overflow -- so a pass
    whose overflow stays under 2^(13 - scale) - 8 is the exact pass.  Random and adversarial frames (distances around 2^24 at
    every scale), including coefficients that can wrap: trusted passes must equal the fast pass field for field, and both
    outcomes must occur."""
    emu.emu_compare_pass_no_round.argtypes = [C.POINTER(C.c_int16), C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(23)
    seen = {0: 0, 2: 0}
    for trial in range(120000):
        kind = trial % 5
        if kind == 0:
            x = rng.integers(-32768, 32768, 16).astype(np.int16)
        elif kind == 1:
            x = rng.integers(-3000, 3000, 16).astype(np.int16)
        elif kind == 2:                                     # residuals around 8192 = 2^24 / 2048: the boundary itself
            base = rng.integers(-12000, 12000)
            x = (base + rng.integers(-9000, 9000, 16)).clip(-32768, 32767).astype(np.int16)
        elif kind == 3:
            x = (np.arange(16) * int(rng.integers(-900, 900)) + int(rng.integers(-9000, 9000))).clip(-32768, 32767).astype(np.int16)
        else:
            x = np.where(rng.integers(0, 2, 16) > 0, rng.integers(7000, 9500), -rng.integers(7000, 9500)).astype(np.int16)
        if trial % 7 == 0:
            c0, c1 = int(rng.integers(-32768, 32768)), int(rng.integers(-32768, 32768))
        else:
            c0, c1 = int(rng.integers(-4096, 4097)), int(rng.integers(-2048, 2049))
        if trial % 11 == 0:
            c0 = c1 = 0                                     # d = in * 2048 exactly: |in| >= 8192 is the boundary
        sp = int(rng.integers(0, 13))
        rc = emu.emu_compare_pass_no_round(x.ctypes.data_as(C.POINTER(C.c_int16)), c0, c1, sp)
        assert rc != 1, (x.tolist(), c0, c1, sp)
        seen[rc] += 1
    assert seen[0] > 10000 and seen[2] > 20000, seen


def test_wide_sum_pass_equals_literal_whatever_the_overflow(emu):
    """the cold block's pass for a final pass at the cap that overflowed by more than 3 (loud noise, clipped waves: the
    32-bit error sum of the fast pass may have wrapped): every scale, rail-to-rail frames, any coefficients that cannot wrap"""
    rng = np.random.default_rng(7)
    checked = big = 0
    for trial in range(40000):
        kind = trial % 4
        if kind == 0:
            x = rng.integers(-32768, 32768, 16).astype(np.int16)
        elif kind == 1:
            x = np.where(rng.integers(0, 2, 16) > 0, 32767, -32768).astype(np.int16)
        elif kind == 2:
            x = (32767 * np.sign(np.sin(np.arange(16) * rng.uniform(0.05, 3.0) + rng.uniform(0, 6.3)))).astype(np.int16)
        else:
            x = rng.integers(-3000, 3000, 16).astype(np.int16)
        c0 = int(rng.integers(-16384, 16384))
        c1 = int(rng.integers(-(32767 - abs(c0)), 32767 - abs(c0) + 1))
        sp = int(rng.integers(0, 13)) if trial % 3 else 12
        rc = emu.emu_compare_pass_wide(x.ctypes.data_as(C.POINTER(C.c_int16)), c0, c1, sp)
        assert rc == 0, (x.tolist(), c0, c1, sp, rc)
        checked += 1
    assert checked == 40000
    # and coefficients that can wrap are refused
    x = rng.integers(-32768, 32768, 16).astype(np.int16)
    assert emu.emu_compare_pass_wide(x.ctypes.data_as(C.POINTER(C.c_int16)), 30000, 30000, 12) == 2


def test_fast_pass_equals_literal_when_it_claims_exactness(emu):
    rng = np.random.default_rng(1)
    exact = 0
    for trial in range(60000):
        amp = int(rng.choice([8, 200, 3000, 20000, 32768]))
        x = rng.integers(-amp, amp, 16).clip(-32768, 32767).astype(np.int16)
        if trial % 5 == 0:      # predictable content so small scales are exact too
            x = (np.arange(16) * int(rng.integers(-300, 300)) + int(rng.integers(-2000, 2000))).clip(-32768, 32767).astype(np.int16)
        c0 = int(rng.integers(-4096, 4096)) if trial % 3 else int(rng.integers(-32768, 32768))
        c1 = int(rng.integers(-2048, 2048)) if trial % 3 else int(rng.integers(-32768, 32768))
        sp = int(rng.integers(0, 13))
        rc = emu.emu_compare_pass(x.ctypes.data_as(C.POINTER(C.c_int16)), c0, c1, sp)
        assert rc != 1, (x.tolist(), c0, c1, sp)
        exact += rc == 0
    assert exact > 10000


def test_emulated_kernel_flow_matches_oracle_synthetic(emu):
    pcm = synth.generate(6, 14 * 4000 + 9)
    tot = np.zeros(8, np.uint64)
    for c in range(6):
        coefs = po.gc_calculate_coefficients(pcm[c])
        got, stats = _emu_encode(emu, pcm[c], coefs)
        assert (got == po.gc_encode(pcm[c], coefs)).all()
        tot += stats
    # the speculation pays: nearly every predictor is resolved by candidate A or B
    assert tot[4] < 0.02 * (tot[2] + tot[3] + tot[4]), tot


@pytest.mark.parametrize("kind", ["noise_fs", "square_fs", "alt_fs", "dc_max", "impulse", "tiny", "sine", "burst"])
@pytest.mark.parametrize("coef_kind", ["real", "bounded", "wrapping", "zero"])
def test_emulated_kernel_flow_matches_oracle_edges(emu, kind, coef_kind):
    rng = np.random.default_rng(hash((kind, coef_kind)) % 2**32)
    n = 14 * 500 + 5
    t = np.arange(n)
    pcm = {
        "noise_fs": rng.integers(-32768, 32768, n),
        "square_fs": np.where((t // 7) % 2 == 0, 32767, -32768),
        "alt_fs": np.where(t % 2 == 0, 32767, -32768),
        "dc_max": np.full(n, 32767),
        "impulse": np.where(t % 97 == 0, 30000, 0),
        "tiny": rng.integers(-3, 4, n),
        "sine": synth.sine(n).astype(np.int64),
        "burst": np.where((t // 500) % 2 == 0, rng.integers(-20000, 20000, n), 0),
    }[kind].astype(np.int16)
    coefs = {
        "real": po.gc_calculate_coefficients(pcm),
        "bounded": rng.integers(-16383, 16384, 16).astype(np.int16),
        "wrapping": rng.integers(-32768, 32768, 16).astype(np.int16),
        "zero": np.zeros(16, np.int16),
    }[coef_kind]
    if coef_kind == "wrapping" and kind == "dc_max":
        coefs[:] = -32768              # the reference's non-terminating case (guarded identically)
    h1, h2 = int(rng.integers(-32768, 32768)), int(rng.integers(-32768, 32768))
    got, _ = _emu_encode(emu, pcm, coefs, h1, h2)
    want = po.gc_encode(pcm, coefs, hist1=h1, hist2=h2)
    assert (got == want).all(), int(np.argmax(got != want))
