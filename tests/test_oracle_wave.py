"""Oracle pins for the WAVE reader/writer: the reference's own build -> parse identities
(VGAudio.Tests/Containers/WaveTests.cs:9-43, 16-bit cases), header bytes derived by hand from WaveWriter.cs:56-129,
and the parser's rejections (WaveReader.cs:70-98).  vga_wave_parse is host-only and compared here as well."""
import ctypes as C
import struct

import numpy as np
import pytest

from oracle import pyoracle as po

FREQS = [261.63, 329.63, 392, 523.25, 659.25, 783.99, 1046.50, 130.81]      # GenerateAudio.cs:14


def sine(n, f, rate):
    i = np.arange(n, dtype=np.float64)
    return np.trunc(32767 * np.sin(2 * np.pi * f / rate * i)).astype(np.int16)


def product_parse(f):
    from vgaudio_amd import _lib
    data = np.frombuffer(bytes(f), np.uint8)
    w = _lib.WaveInfoC()
    return _lib.lib().vga_wave_parse(data.ctypes.data_as(_lib.u8p), len(data), C.byref(w)), w


SHARED = ("channel_count", "sample_rate", "bits_per_sample", "sample_count", "sample_count_declared", "looping", "loop_start",
          "loop_end", "data_offset", "data_size", "data_size_declared")


def assert_same_parse(f):
    rc, w = po.wave_parse(f)
    prc, pw = product_parse(f)
    assert (rc == 0) == (prc == 0), (rc, prc)
    if rc in (-2, -3):
        assert prc == rc                                                       # same exception class
    if rc == 0:
        for name in SHARED:
            assert getattr(w, name) == getattr(pw, name), name
    return rc, w


@pytest.mark.parametrize("nch", [1, 2, 8])
@pytest.mark.parametrize("looped", [False, True])
def test_wave_pcm16_build_and_parse_equal(nch, looped):                       # WaveTests.cs:9-18, :33-43
    pcm = [sine(48000, FREQS[i], 48000) for i in range(nch)]
    rc, f = po.wave_write(pcm, 48000, looped, 0, 48000 if looped else 0)       # WithLoop(true) = whole file
    assert rc == 0
    rc, w, chans = po.wave_read(f)
    assert rc == 0
    assert (w.channel_count, w.sample_rate, w.sample_count, w.looping, w.loop_start, w.loop_end) == (
        nch, 48000, 48000, int(looped), 0, 48000 if looped else 0)
    for a, b in zip(pcm, chans):
        assert np.array_equal(a, b)
    assert_same_parse(f)


def test_header_bytes_stereo():
    pcm = [np.arange(5, dtype=np.int16), -np.arange(5, dtype=np.int16)]
    rc, f = po.wave_write(pcm, 32000)
    f = bytes(f)
    assert len(f) == 44 + 20
    assert f[:4] == b"RIFF" and struct.unpack("<i", f[4:8])[0] == len(f) - 8 and f[8:16] == b"WAVEfmt "
    assert struct.unpack("<ihhiihh", f[16:36]) == (16, 1, 2, 32000, 32000 * 4, 4, 16)
    assert f[36:40] == b"data" and struct.unpack("<i", f[40:44])[0] == 20
    assert struct.unpack("<10h", f[44:]) == (0, 0, 1, -1, 2, -2, 3, -3, 4, -4)


def test_header_bytes_extensible_and_smpl():
    pcm = [np.full(3, c, dtype=np.int16) for c in range(6)]
    rc, f = po.wave_write(pcm, 44100, True, 1, 3)
    f = bytes(f)
    assert struct.unpack("<ihhiihh", f[16:36]) == (40, -2, 6, 44100, 44100 * 12, 12, 16)    # 0xFFFE as a short
    assert struct.unpack("<hhi", f[36:44]) == (22, 16, 0x0633)
    assert f[44:60] == bytes([1, 0, 0, 0, 0, 0, 0x10, 0, 0x80, 0, 0, 0xAA, 0, 0x38, 0x9B, 0x71])
    assert f[60:64] == b"smpl" and struct.unpack("<i", f[64:68])[0] == 0x3c
    body = struct.unpack("<15i", f[68:128])
    assert body[:7] == (0,) * 7 and body[7] == 1 and body[8:11] == (0, 0, 0) and body[11:13] == (1, 3) and body[13:] == (0, 0)
    assert f[128:132] == b"data" and struct.unpack("<i", f[132:136])[0] == 36
    assert len(f) == 136 + 36
    rc, w, chans = po.wave_read(f)
    assert rc == 0 and (w.looping, w.loop_start, w.loop_end) == (1, 1, 3)


def _riff(chunks, form=b"WAVE", size=None):
    body = form + b"".join(cid + struct.pack("<i", len(c) if n is None else n) + c + (b"\0" if len(c) & 1 else b"")
                           for cid, c, n in chunks)
    return b"RIFF" + struct.pack("<i", len(body) if size is None else size) + body


def _fmt(tag=1, nch=2, rate=8000, bits=16, align=None):
    align = (bits + 7) // 8 * nch if align is None else align
    return struct.pack("<HhiihH", tag, nch, rate, rate * align, align, bits)


def test_parser_rejections_and_quirks():
    data = bytes(range(16))
    ok = _riff([(b"fmt ", _fmt(), None), (b"data", data, None)])
    rc, w = assert_same_parse(ok)
    assert rc == 0 and w.sample_count == 4 and w.data_offset == 44
    assert assert_same_parse(b"RIFX" + ok[4:])[0] == -3                        # Not a valid RIFF file
    assert assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"data", data, None)], form=b"WAVX"))[0] == -3
    assert assert_same_parse(_riff([(b"data", data, None)]))[0] == -3          # no fmt chunk
    assert assert_same_parse(_riff([(b"fmt ", _fmt(), None)]))[0] == -3        # no data chunk
    assert assert_same_parse(_riff([(b"fmt ", _fmt(tag=2), None), (b"data", data, None)]))[0] == -3
    assert assert_same_parse(_riff([(b"fmt ", _fmt(bits=24), None), (b"data", data, None)]))[0] == -3
    assert assert_same_parse(_riff([(b"fmt ", _fmt(nch=0, align=0), None), (b"data", data, None)]))[0] == -3
    assert assert_same_parse(_riff([(b"fmt ", _fmt(align=2), None), (b"data", data, None)]))[0] == -3
    bad_guid = _fmt(tag=0xFFFE) + struct.pack("<hhI", 22, 16, 3) + bytes(16)
    assert assert_same_parse(_riff([(b"fmt ", bad_guid, None), (b"data", data, None)]))[0] == -3
    # unknown chunks and odd sizes are skipped with 2-byte alignment; a later duplicate chunk wins
    odd = _riff([(b"LIST", b"abc", None), (b"fmt ", _fmt(nch=1), None), (b"fact", struct.pack("<i", 8), None),
                 (b"data", b"\1\0", None), (b"data", data, None)])
    rc, w = assert_same_parse(odd)
    assert rc == 0 and w.sample_count == 8 and w.channel_count == 1
    # the data chunk claims more than the file holds: the channels get what is there (Interleave.cs:190),
    # WaveStructure.SampleCount what was declared (:27) -- but only if the RIFF size stops the chunk loop
    short = _riff([(b"fmt ", _fmt(), None), (b"data", data, 400)])
    rc, w = assert_same_parse(short)
    assert rc == 0 and (w.sample_count, w.sample_count_declared) == (4, 100)
    assert assert_same_parse(ok[:30])[0] != 0                                  # ends inside the fmt chunk
    assert assert_same_parse(ok[:10])[0] != 0
    assert assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"data", data, None)], size=4000))[0] != 0   # RIFF size past the end
    # 8-bit files parse (the header is valid); conversion is refused elsewhere
    rc, w = assert_same_parse(_riff([(b"fmt ", _fmt(bits=8), None), (b"data", data, None)]))
    assert rc == 0 and w.bits_per_sample == 8 and w.sample_count == 8


def test_smpl_loop_rules():
    data = bytes(64)
    def smpl(start, end, loops=1):
        return struct.pack("<9i", 0, 0, 0, 0, 0, 0, 0, loops, 0) + b"".join(struct.pack("<6i", 0, 0, start, end, 0, 0) for _ in range(loops))
    rc, w = assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"smpl", smpl(3, 9), None), (b"data", data, None)]))
    assert rc == 0 and (w.looping, w.loop_start, w.loop_end) == (1, 3, 9)
    rc, w = assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"smpl", smpl(9, 9), None), (b"data", data, None)]))
    assert rc == 0 and (w.looping, w.loop_start, w.loop_end) == (0, 0, 0)      # End > Start or it does not loop (:36)
    rc, w = assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"smpl", smpl(0, 0, loops=0), None), (b"data", data, None)]))
    assert rc == 0 and w.looping == 0
    assert assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"smpl", smpl(3, 17), None), (b"data", data, None)]))[0] == -2   # past the end
    assert assert_same_parse(_riff([(b"fmt ", _fmt(), None), (b"smpl", smpl(-1, 9), None), (b"data", data, None)]))[0] == -2


def test_random_files_parse_identically():
    rng = np.random.default_rng(5)
    base = bytearray(_riff([(b"fmt ", _fmt(), None), (b"smpl", struct.pack("<9i", 0, 0, 0, 0, 0, 0, 0, 1, 0) + struct.pack("<6i", 0, 0, 1, 5, 0, 0), None),
                            (b"data", bytes(range(40)), None)]))
    for _ in range(400):
        f = bytearray(base)
        for _k in range(int(rng.integers(1, 4))):
            f[int(rng.integers(0, len(f)))] = int(rng.integers(0, 256))
        if rng.integers(0, 4) == 0:
            f = f[:int(rng.integers(0, len(f) + 1))]
        assert_same_parse(bytes(f))
