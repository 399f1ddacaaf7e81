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
