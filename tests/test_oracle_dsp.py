"""Oracle pins for the DSP container: the interleave vectors of the reference's own
tests (VGAudio.Tests/Utilities/InterleaveTests.cs, DeinterleaveTests.cs), header bytes derived by hand from
DspWriter.cs:52-80, and the reference's build -> parse identity (Containers/DspTests.cs:9-18)."""
import json
import os
import struct

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(__file__)
KATS = json.load(open(os.path.join(HERE, "golden", "interleave_kats.json")))
FREQS = [261.63, 329.63, 392, 523.25, 659.25, 783.99, 1046.50, 130.81]      # GenerateAudio.cs:14


def sine(n, f, rate):                                                        # GenerateAudio.cs:23-33
    i = np.arange(n, dtype=np.float64)
    return np.trunc(32767 * np.sin(2 * np.pi * f / rate * i)).astype(np.int16)


def encoded(nch, n, rate=48000):
    pcm = [sine(n, FREQS[i % 8], rate) for i in range(nch)]
    coefs = [po.gc_calculate_coefficients(p) for p in pcm]
    adpcm = [po.gc_encode(p, c) for p, c in zip(pcm, coefs)]
    return adpcm, np.stack(coefs)


@pytest.mark.parametrize("case", KATS["interleave"], ids=lambda c: f"s{c['size']}o{c['output_size']}n{len(c['inputs'])}")
def test_interleave_reference_vectors(case):
    out = po.interleave([np.array(r, np.uint8) for r in case["inputs"]], case["size"], case["output_size"])
    assert out.tolist() == case["expected"]


@pytest.mark.parametrize("case", KATS["deinterleave"], ids=lambda c: f"s{c['size']}c{c['count']}o{c['output_size']}")
def test_deinterleave_reference_vectors(case):
    rc, outs = po.deinterleave(np.array(case["input"], np.uint8), case["size"], case["count"], case["output_size"])
    assert rc == 0
    assert [o.tolist() for o in outs] == case["expected"]


def test_deinterleave_rejects_indivisible_length():                          # DeinterleaveTests: throws
    rc, _ = po.deinterleave(np.arange(15, dtype=np.uint8), 2, 2)
    assert rc != 0


def test_header_bytes_stereo_non_looping():
    n, rate = 100, 32000
    adpcm, coefs = encoded(2, n, rate)
    rc, f = po.dsp_write(adpcm, coefs, po.dsp_params(rate, n), gain=[0, 0],
                         start_context=[[adpcm[0][0], 0, 0], [adpcm[1][0], 0, 0]])
    assert rc == 0
    nbytes = po.gc_sample_count_to_byte_count(n)                              # 100 samples = 7 frames + 2 -> 58 bytes
    assert nbytes == 58
    assert len(f) == 2 * (0x60 + 64)                                          # AudioDataSize rounds 58 up to 64 (:99-100)
    for i in range(2):
        h = bytes(f[0x60 * i:0x60 * (i + 1)])
        sample_count, nibbles, srate, loop, fmt, start, end, cur = struct.unpack(">iiihhiii", h[:0x1c])
        assert (sample_count, nibbles, srate, loop, fmt) == (100, 116, rate, 0, 0)      # 100 samples = 7*16 + 2 + 2 nibbles
        assert (start, end, cur) == (2, po.gc_sample_to_nibble(99), 2)
        assert list(struct.unpack(">16h", h[0x1c:0x3c])) == coefs[i].tolist()
        assert struct.unpack(">h", h[0x3c:0x3e])[0] == 0
        assert struct.unpack(">3h", h[0x3e:0x44]) == (adpcm[i][0], 0, 0)
        assert h[0x44:0x4a] == bytes(6)
        assert struct.unpack(">hh", h[0x4a:0x4e]) == (2, 0x3800 // 14)
        assert h[0x4e:] == bytes(0x60 - 0x4e)
    data = bytes(f[0xc0:])
    assert data[:58] == bytes(adpcm[0]) and data[58:64] == bytes(6)           # one (short) interleave block per channel
    assert data[64:122] == bytes(adpcm[1]) and data[122:] == bytes(6)


def test_header_bytes_mono_looping():
    n, rate = 1000, 44100
    adpcm, coefs = encoded(1, n, rate)
    p = po.dsp_params(rate, n, looping=True, loop_start=140, loop_end=900)
    rc, f = po.dsp_write(adpcm, coefs, p, loop_context=[[0x17, -5, 300]])
    assert rc == 0
    nb = po.gc_sample_count_to_byte_count(900)                                # TrimFile: SampleCount = LoopEnd (:22)
    assert len(f) == 0x60 + nb
    h = bytes(f[:0x60])
    assert struct.unpack(">iiihhiii", h[:0x1c]) == (900, po.gc_sample_count_to_nibble_count(900), rate, 1, 0,
                                                     po.gc_sample_to_nibble(140), po.gc_sample_to_nibble(900), 2)
    assert struct.unpack(">3h", h[0x44:0x4a]) == (0x17, -5, 300)
    assert struct.unpack(">hh", h[0x4a:0x4e]) == (0, 0)                       # mono: channel count and interleave 0 (:78-79)
    assert bytes(f[0x60:]) == bytes(adpcm[0][:nb])


def test_layout_quirks():
    # LoopPointAlignment shifts the header's loop points but not the audio (DspWriter.cs:29-31)
    rc, L = po.dsp_layout(po.dsp_params(48000, 5000, True, 100, 4000, loop_point_alignment=1024), 2)
    assert rc == 0 and (L.loop_start, L.loop_end, L.sample_count) == (1024, 4924, 4924)
    # TrimFile=false keeps the longer of SampleCount / LoopEnd
    rc, L = po.dsp_layout(po.dsp_params(48000, 5000, True, 100, 4000, trim_file=False), 2)
    assert rc == 0 and L.sample_count == 5000
    for bad in (0, -14, 15):                                                  # DspConfiguration.cs:31-45
        rc, _ = po.dsp_layout(po.dsp_params(48000, 5000, samples_per_interleave=bad), 2)
        assert rc != 0


@pytest.mark.parametrize("nch", [1, 2, 8])
def test_dsp_build_and_parse_equal(nch):                                     # DspTests.cs:9-18 (BuildParseTestOptions: 48000 samples at 48 kHz)
    for n, spi in ((100, 0x3800), (48000, 0x3800), (40000, 14 * 64)):
        adpcm, coefs = encoded(nch, n)
        start = [[a[0], 0, 0] for a in adpcm]
        rc, f = po.dsp_write(adpcm, coefs, po.dsp_params(48000, n, samples_per_interleave=spi), start_context=start)
        assert rc == 0
        rc, h, rcoefs, gain, sc, lc, chans = po.dsp_read(f)
        assert rc == 0
        assert (h.sample_count, h.channel_count, h.sample_rate, h.looping) == (n, nch, 48000, 0)
        assert np.array_equal(rcoefs, coefs)
        assert sc.tolist() == start and not lc.any() and not gain.any()
        for a, b in zip(adpcm, chans):
            assert np.array_equal(a, b)


def test_read_rejects_bad_headers():
    adpcm, coefs = encoded(1, 100)
    rc, f = po.dsp_write(adpcm, coefs, po.dsp_params(48000, 100))
    g = f.copy(); g[7] ^= 1                                                   # nibble count mismatch (DspReader.cs:96-99)
    assert po.dsp_read(g)[0] != 0
    g = f.copy(); g[0x0f] = 1                                                 # not ADPCM (:101-104)
    assert po.dsp_read(g)[0] != 0
