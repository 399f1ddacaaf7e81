"""CPU-side check of the HCA decoder's per-lane logic (vgaudio_amd/csrc/hca_decode_core.hpp): the header the two
decode kernels are built from is compiled for the host into a lane emulator (tests/host/hca_decode_emulator.cpp) that
runs a workgroup's threads phase by phase, and its PCM is compared with the oracle's bit for bit.  This is host logic
under test (the length-only scan, the chunk offsets, the 8-lane DCT-IV decomposition, the row rotation and overlap
carry), not a CPU product path."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import _lib, synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "hca_decode_emulator.cpp")
CSRC = os.path.join(HERE, "..", "vgaudio_amd", "csrc")
DEPS = [SRC] + [os.path.join(CSRC, f) for f in ("hca_decode_core.hpp", "hca_info.hpp", "hca_tables_data.h")]
SO = os.path.join(HERE, "host", "libhca_decode_emulator.so")
DEVICE_INFO_BYTES = 11 * 4 + 8 * 4 + 8 * 4 + 128


@pytest.fixture(scope="module")
def emu():
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in DEPS):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fwrapv", "-ffp-contract=off", "-fno-fast-math",
                        "-Wno-unused-variable", SRC, "-o", SO], check=True)
    L = C.CDLL(SO)
    vp, u8p = C.c_void_p, C.POINTER(C.c_uint8)
    L.emu_record_bytes.argtypes = [vp]
    L.emu_hca_scan.argtypes = [vp, u8p, C.c_int64, u8p]
    L.emu_hca_frames.argtypes = [vp, u8p, C.c_int64, u8p, C.POINTER(C.c_int16), C.c_int64, C.c_int]
    return L


def device_info(info):
    """the product's own derivation (channel types, coded counts, scaled ATH curve); host code, no GPU"""
    out = (C.c_uint8 * DEVICE_INFO_BYTES)()
    pinfo = _lib.HcaInfoC()
    for name, _ in _lib.HcaInfoC._fields_:
        setattr(pinfo, name, getattr(info, name))
    _lib.check(_lib.lib().vga_testing_hca_device_info(C.byref(pinfo), out, DEVICE_INFO_BYTES))
    return out


def emu_decode(emu, info, frames, frames_per_group, shift=0):
    """frames: uint8 [frame_count * frame_size]; returns (flags, pcm [nch, sample_count])"""
    d = device_info(info)
    fbytes = info.frame_count * info.frame_size
    pitch = (fbytes + 8 + 15) // 16 * 16
    stream = np.zeros(pitch, np.uint8)
    stream[:fbytes] = np.asarray(frames, np.uint8).reshape(-1)[:fbytes]
    rb = emu.emu_record_bytes(d)
    records = np.zeros(info.frame_count * rb, np.uint8)
    u8p = C.POINTER(C.c_uint8)
    flags = emu.emu_hca_scan(d, stream.ctypes.data_as(u8p), pitch, records.ctypes.data_as(u8p))
    assert flags >= 0
    n = info.sample_count
    pcm = np.zeros((info.channel_count, max(n, 1)), np.int16)
    rc = emu.emu_hca_frames(d, stream.ctypes.data_as(u8p), pitch, records.ctypes.data_as(u8p),
                            pcm.ctypes.data_as(C.POINTER(C.c_int16)), pcm.shape[1], frames_per_group)
    assert rc == 0
    return flags, pcm[:, :n]


def _signal(nch, n, seed):
    x = synth.generate(nch, n, first_channel=seed).astype(np.int32)
    rng = np.random.default_rng(seed)
    x[:, n // 3:n // 2] = rng.integers(-30000, 30000, (nch, n // 2 - n // 3))          # a loud noisy stretch: big codes
    x[:, n // 2:n // 2 + 700] = 0                                                       # and silence: empty frames
    return x.clip(-32768, 32767).astype(np.int16)


CASES = [  # nch, quality, n, bitrate, frames per group
    (1, "High", 5000, 0, 1), (2, "High", 9000, 0, 16), (2, "Highest", 4096, 0, 3), (2, "Lowest", 7000, 0, 2),
    (2, "Low", 6000, 0, 16), (1, "Lowest", 3000, 0, 5), (4, "Middle", 3500, 0, 2), (6, "Low", 3000, 0, 4),
    (2, "High", 1024 * 12 + 17, 64000, 7), (3, "High", 2500, 0, 1), (8, "Lowest", 2100, 0, 16), (2, "High", 100, 0, 16),
]


@pytest.mark.parametrize("nch,quality,n,bitrate,group", CASES)
def test_emulated_kernels_match_the_oracle(emu, nch, quality, n, bitrate, group):
    pcm = _signal(nch, n, nch * 7 + n)
    rc, info, frames = po.hca_encode(pcm, po.hca_params(nch, n, quality=quality, bitrate=bitrate))
    assert rc == 0
    rc, want = po.hca_decode(info, frames)
    assert rc == 0
    flags, got = emu_decode(emu, info, frames, group)
    assert flags == 0
    assert np.array_equal(got, np.asarray(want).reshape(nch, n))


def test_garbage_frames_read_zeros_past_the_end_like_the_reference(emu):
    """Frames that are random bits after a valid header walk far past the frame's end (BitReader.PeekInt then yields
    zeros, BitReader.cs:55-61); scale factors are sent raw (delta bits 6/7) so that delta decoding cannot fail."""
    rng = np.random.default_rng(5)
    n = 1024 * 6
    for nch, quality in ((2, "High"), (1, "Lowest"), (2, "Lowest")):
        pcm = _signal(nch, n, 3)
        rc, info, frames = po.hca_encode(pcm, po.hca_params(nch, n, quality=quality))
        assert rc == 0
        fr = np.array(frames, np.uint8).reshape(info.frame_count, info.frame_size)
        dinfo = np.frombuffer(bytes(device_info(info)), np.int32, 27)      # [11:19] channel types, [19:27] coded counts
        for f in range(1, info.frame_count):
            bits = rng.integers(0, 2, info.frame_size * 8).astype(np.uint8)
            bits[:16] = 1                                           # sync word
            pos = 32
            for c in range(nch):                                    # raw 6-bit scale factors for every channel
                bits[pos:pos + 3] = (1, 1, int(rng.integers(0, 2)))
                pos += 3 + 6 * int(dinfo[19 + c])
                pos += 32 if dinfo[11 + c] == 2 else 6 * info.hfr_group_count      # intensity / HFR scales
            if f % 3 == 0:
                bits[info.frame_size * 4:] = 1                      # long codes at the end: far past the frame
            fr[f] = np.packbits(bits)
        rc, want = po.hca_decode(info, fr.reshape(-1))
        assert rc == 0
        for group in (1, 4):
            flags, got = emu_decode(emu, info, fr.reshape(-1), group)
            assert flags == 0
            assert np.array_equal(got, np.asarray(want).reshape(nch, n)), (nch, quality, group)


def test_dct_lane_decomposition_is_bit_exact(emu):
    """the 8-lane transform against the oracle's Dct4 on random spectra is covered through the decode above; here the
    odd frame sizes: every alignment of a frame inside its stream (frame_size % 4 = 0..3) through explicit bitrates"""
    n = 5000
    seen = set()
    for bitrate in (48000, 48047, 48094, 48141, 50000, 51234, 47001, 52000):
        pcm = _signal(2, n, bitrate)
        rc, info, frames = po.hca_encode(pcm, po.hca_params(2, n, quality="High", bitrate=bitrate))
        assert rc == 0
        seen.add(info.frame_size % 4)
        rc, want = po.hca_decode(info, frames)
        flags, got = emu_decode(emu, info, frames, 16)
        assert flags == 0 and np.array_equal(got, np.asarray(want).reshape(2, n)), bitrate
    assert len(seen) >= 3
