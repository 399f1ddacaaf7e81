"""Oracle pins for the ADX and HCA container writers.  The reference has no tests for
these two containers, so the restatement (oracle/adxhca_container_oracle.c) is pinned by header bytes derived by
hand from AdxWriter.cs:81-139 / HcaWriter.cs:57-179, the interleave vectors (test_oracle_dsp.py), the CRC-16
check value, and a build -> parse identity against readers restated from AdxReader.cs / HcaReader.cs.
The host-only vga_hca_file_header / vga_hca_write (no device work) are compared here as well."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import pyoracle as po


def adx_channels(nch, n, rng, **kw):
    pcm = rng.integers(-20000, 20000, (nch, n)).astype(np.int16)
    p = po.adx_params(**kw)
    audio, hist = po.adx_encode_batch(pcm, p)
    return [audio[c] for c in range(nch)], hist


def test_adx_header_bytes_stereo_v4():
    rng = np.random.default_rng(3)
    audio, hist = adx_channels(2, 1000, rng)
    rc, f = po.adxfile_write(audio, hist, po.adxfile_params(44100, 1000))
    assert rc == 0
    frames = -(-1000 // 32)
    assert len(f) == 40 + 18 * frames * 2 + 18                                # AudioOffset + AudioSize + FooterSize (:18,35)
    h = bytes(f[:40])
    assert h[:2] == b"\x80\x00" and struct.unpack(">h", h[2:4])[0] == 36      # BaseHeaderSize, version 4, no loop (:30)
    assert h[4:8] == bytes([3, 18, 4, 2])                                     # Linear, frame size, bit depth, channels
    assert struct.unpack(">iih", h[8:18]) == (44100, 1000, 500)
    assert h[18:20] == bytes([4, 0])
    assert h[20:24] == bytes(4)
    assert struct.unpack(">4h", h[24:32]) == (hist[0], hist[0], hist[1], hist[1])
    assert h[32:34] == bytes(2)                                               # AlignmentSamples; the loop fields behind it ...
    assert h[34:40] == b"(c)CRI"                                              # ... are overwritten here and by the audio
    body = bytes(f[40:40 + 36 * frames])
    for k in range(frames):
        assert body[36 * k:36 * k + 18] == bytes(audio[0][18 * k:18 * k + 18])
        assert body[36 * k + 18:36 * k + 36] == bytes(audio[1][18 * k:18 * k + 18])
    assert bytes(f[-18:]) == b"\x80\x01" + struct.pack(">h", 14) + bytes(14)   # footer (:133-138)


def test_adx_header_bytes_mono_v3_fixed():
    rng = np.random.default_rng(4)
    audio, hist = adx_channels(1, 640, rng, version=3, type=2, filter=1)
    rc, f = po.adxfile_write(audio, hist, po.adxfile_params(22050, 640, version=3, type=2))
    assert rc == 0
    h = bytes(f[:36])
    assert struct.unpack(">h", h[2:4])[0] == 32 and h[4] == 2 and h[7] == 1
    assert struct.unpack(">h", h[16:18])[0] == 0                              # Fixed: no high-pass frequency (:91)
    assert h[18] == 3
    assert h[20:30] == bytes(10)                                              # loop fields (all zero) run up to "(c)CRI"
    assert h[30:36] == b"(c)CRI"
    assert bytes(f[36:36 + 18 * 20]) == bytes(audio[0])


@pytest.mark.parametrize("version", [3, 4])
@pytest.mark.parametrize("nch", [1, 2])
def test_adx_looping_layout(version, nch):
    spf = 32
    loop_start_unaligned, loop_end_unaligned, n = 1000, 9000, 10000
    mult = spf * 2 if nch == 1 else spf
    align = (-loop_start_unaligned) % mult
    rng = np.random.default_rng(5)
    audio, hist = adx_channels(nch, n, rng, version=version, padding=align)
    p = po.adxfile_params(48000, n + align, True, loop_start_unaligned + align, loop_end_unaligned + align, align, version=version)
    rc, L = po.adxfile_layout(p, nch)
    assert rc == 0
    assert L.sample_count == loop_end_unaligned + align + 3 * spf            # TrimFile (:21)
    assert L.loop_start_offset % 0x800 == 0 or version == 3                  # the point of AlignmentBytes (:58-62)
    assert (L.footer_offset + L.footer_size) % 0x800 == 0                    # :35
    rc, f = po.adxfile_write(audio, hist, p)
    assert rc == 0 and len(f) == L.file_size
    rc, h, rhist, chans = po.adxfile_read(f)
    assert rc == 0 and h.looping == 1
    assert (h.loop_start_sample, h.loop_end_sample, h.inserted_samples) == (p.loop_start, p.loop_end, align)
    assert (h.loop_start_byte, h.loop_end_byte) == (L.loop_start_offset, L.loop_end_offset)
    assert bytes(f[L.header_size - 2:L.header_size + 4]) == b"(c)CRI"
    if version == 4:
        assert rhist.tolist() == hist.tolist()
    keep = L.frame_count * 18
    for a, b in zip(audio, chans):
        assert np.array_equal(a[:keep], b[:len(a[:keep])])
    assert bytes(f[L.footer_offset:L.footer_offset + 4]) == b"\x80\x01" + struct.pack(">h", L.footer_size - 4)


def test_adx_footer_follows_the_interleaver():
    """Loop end == sample count: the trimmed frame count (+3 frames) exceeds what the channels hold, the
    interleaver stops after the input blocks and the footer lands there (Interleave.cs:58-77, AdxWriter.cs:133)."""
    rng = np.random.default_rng(6)
    n = 3200
    audio, hist = adx_channels(2, n, rng)
    p = po.adxfile_params(48000, n, True, 0, n, 0)
    rc, L = po.adxfile_layout(p, 2)
    assert L.frame_count == n // 32 + 3
    rc, f = po.adxfile_write(audio, hist, p)
    assert rc == 0
    pos = L.audio_offset + (n // 32) * 18 * 2
    assert bytes(f[pos:pos + 4]) == b"\x80\x01" + struct.pack(">h", L.footer_size - 4)
    assert not f[pos + 4:].any()


@pytest.mark.parametrize("nch", [1, 2, 3, 6])
def test_adx_build_and_parse_equal(nch):
    rng = np.random.default_rng(nch)
    for n in (1, 31, 32, 33, 5000):
        audio, hist = adx_channels(nch, n, rng)
        rc, f = po.adxfile_write(audio, hist, po.adxfile_params(32000, n))
        assert rc == 0
        rc, h, rhist, chans = po.adxfile_read(f)
        assert rc == 0 and (h.channel_count, h.sample_count, h.sample_rate, h.version, h.frame_size) == (nch, n, 32000, 4, 18)
        if nch <= 2:                                                           # more channels: "(c)CRI" lands on the histories
            assert rhist.tolist() == hist.tolist()
        for a, b in zip(audio, chans):
            assert np.array_equal(a, b)


# ------------------------------------------------------------------ HCA
def hca_stream(nch=2, n=5000, **kw):
    rng = np.random.default_rng(11)
    pcm = rng.integers(-8000, 8000, (nch, n)).astype(np.int16)
    rc, info, frames = po.hca_encode(pcm, po.hca_params(nch, n, **kw))
    assert rc == 0
    return info, frames


def test_crc16_check_value():
    assert po.lib().vgo_crc16(po._u8(np.frombuffer(b"123456789", np.uint8).copy()), 9) == 0xFEE8


def test_hca_header_bytes():
    info, frames = hca_stream()
    rc, f = po.hcafile_write(info, frames)
    assert rc == 0 and len(f) == info.header_size + info.frame_size * info.frame_count
    h = bytes(f[:info.header_size])
    assert info.header_size == 96
    assert h[:8] == b"HCA\0" + struct.pack(">hh", 0x0200, 96)
    assert h[8:12] == b"fmt\0" and h[12] == 2 and h[13:16] == (48000).to_bytes(3, "big")
    assert struct.unpack(">iHH", h[16:24]) == (info.frame_count, info.inserted_samples, info.appended_samples)
    assert h[24:28] == b"comp" and struct.unpack(">h", h[28:30])[0] == info.frame_size
    assert list(h[30:38]) == [info.min_resolution, info.max_resolution, info.track_count, info.channel_config,
                              info.total_band_count, info.base_band_count, info.stereo_band_count, info.bands_per_hfr_group]
    assert h[38:40] == bytes(2)
    assert h[40:46] == b"ciph" + bytes(2)
    assert h[46:49] == b"pad" and h[49:94] == bytes(45)
    assert struct.unpack(">H", h[94:96])[0] == po.lib().vgo_crc16(po._u8(np.frombuffer(h[:94], np.uint8).copy()), 94)
    assert po.lib().vgo_crc16(po._u8(np.frombuffer(h, np.uint8).copy()), 96) == 0     # a CRC over data + CRC is 0
    assert bytes(f[96:]) == frames.tobytes()


def test_hca_header_optional_chunks_round_trip():
    info, frames = hca_stream(looping=True, loop_start=1200, loop_end=4800)
    assert info.looping == 1
    rc, f = po.hcafile_write(info, frames, comment=None, volume=0.5, encryption_type=0)
    assert rc == 0
    h = bytes(f[:info.header_size])
    assert h[40:44] == b"loop" and struct.unpack(">iihh", h[44:56]) == (info.loop_start_frame, info.loop_end_frame,
                                                                          info.pre_loop_samples, info.post_loop_samples)
    assert h[56:62] == b"ciph" + bytes(2)
    assert h[62:66] == b"rva\0" and struct.unpack(">f", h[66:70])[0] == 0.5
    assert h[70:73] == b"pad"
    rc, r, vol, enc, comment, ver = po.hcafile_read(f)
    assert rc == 0 and ver == 0x0200 and vol == 0.5 and enc == 0 and comment == ""
    for name, _ in po.HcaInfo._fields_:
        if name in ("hfr_band_count", "hfr_group_count", "use_ath_curve", "comment_length"):
            continue                                                           # derived by HcaInfo.CalculateHfrValues / not stored
        assert getattr(r, name) == getattr(info, name), name


def test_hca_comment_chunk():
    info, frames = hca_stream()
    info.comment_length = 11
    info.header_size = 128                                                     # GetNextMultiple(96 + 11, 32) (CriHcaEncoder.cs:406)
    rc, f = po.hcafile_write(info, frames, comment="hello world")
    assert rc == 0
    h = bytes(f[:128])
    assert h[46:51] == b"comm\0" and h[51:63] == b"hello world\0"
    rc, r, vol, enc, comment, ver = po.hcafile_read(f)
    assert rc == 0 and comment == "hello world"
    rc, f2 = po.hcafile_write(info, frames, comment=" \t ")                    # IsNullOrWhiteSpace -> pad (:66)
    assert bytes(f2[46:49]) == b"pad"


def test_host_header_matches_oracle():
    """vga_hca_file_header / vga_hca_write do no device work: compared on the CPU."""
    from vgaudio_amd import _lib
    for kw, comment, volume, enc, masked in (({}, None, 1.0, 0, 0), (dict(looping=True, loop_start=1200, loop_end=4800), None, 0.25, 1, 1),
                                             ({}, "a comment", 1.0, 56, 1)):
        info, frames = hca_stream(**kw)
        if comment:
            info.comment_length = len(comment)
            info.header_size = -(-(96 + len(comment)) // 32) * 32
        rc, want = po.hcafile_write(info, frames, comment=comment, volume=volume, encryption_type=enc, encrypted_ids=masked)
        assert rc == 0
        if masked:
            assert bytes(want[:4]) == b"\xc8\xc3\xc1\x00" and bytes(want[8:12]) == b"\xe6\xed\xf4\x00"   # "HCA\0", "fmt\0" | 0x80
        ci = _lib.HcaInfoC()
        C.memmove(C.byref(ci), C.byref(info), C.sizeof(ci))
        size = _lib.lib().vga_hca_file_size(C.byref(ci))
        assert size == len(want)
        got = np.zeros(size, np.uint8)
        fr = np.ascontiguousarray(frames).reshape(-1)
        cb = None if comment is None else comment.encode()
        _lib.check(_lib.lib().vga_hca_write(C.byref(ci), fr.ctypes.data_as(_lib.u8p), cb, volume, enc, masked, got.ctypes.data_as(_lib.u8p)))
        assert got.tobytes() == want.tobytes()
        hdr = np.zeros(info.header_size, np.uint8)
        _lib.check(_lib.lib().vga_hca_file_header(C.byref(ci), cb, volume, enc, masked, hdr.ctypes.data_as(_lib.u8p)))
        assert hdr.tobytes() == want[:info.header_size].tobytes()
    # chunks that do not fit HeaderSize
    ci.header_size = 48
    assert _lib.lib().vga_hca_file_header(C.byref(ci), b"x" * 40, 1.0, 0, 0, hdr.ctypes.data_as(_lib.u8p)) == _lib.InvalidOperationError.code


def test_adx_layout_matches_oracle_on_host():
    """vga_adx_file_layout_for is size math only."""
    from vgaudio_amd import _lib
    rng = np.random.default_rng(0)
    for _ in range(300):
        nch = int(rng.integers(1, 9))
        n = int(rng.integers(0, 200000))
        looping = bool(rng.integers(0, 2))
        ls = int(rng.integers(0, n + 1)); le = int(rng.integers(ls, n + 1))
        align = int(rng.integers(0, 64))
        version = int(rng.choice([3, 4])); trim = bool(rng.integers(0, 2))
        vals = (48000, n + align, int(looping), ls + align, le + align, align, 18, version, 3, 500, 0, int(trim))
        rc, want = po.adxfile_layout(po.AdxFileParams(*vals), nch)
        p = _lib.AdxFileParamsC(*vals)
        got = _lib.AdxFileLayoutC()
        assert _lib.lib().vga_adx_file_layout_for(C.byref(p), nch, C.byref(got)) == 0 and rc == 0
        for name, _t in got._fields_:
            assert getattr(got, name) == getattr(want, name), (name, vals, nch)
