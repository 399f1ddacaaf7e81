"""CPU checks of the ADX oracle (oracle/adx_oracle.c).  The reference has NO tests for the ADX
codec (parity unpinned); these pin the restatement with hand-derived vectors,
the documented constants, and encode->decode self-consistency."""
import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import synth


def test_highpass_coefficients_48k():
    # 500 Hz high-pass at 48 kHz -> (7400, -3342)
    assert po.adx_calculate_coefficients(500, 48000).tolist() == [7400, -3342]


def test_helpers_size_math():
    L = po.lib()
    # 18-byte frames: 36 nibbles, 32 samples
    assert L.vgo_adx_sample_count_to_nibble_count(32, 18) == 36
    assert L.vgo_adx_sample_count_to_nibble_count(33, 18) == 36 + 5
    assert L.vgo_adx_sample_count_to_byte_count(33, 18) == 21
    assert L.vgo_adx_nibble_count_to_sample_count(41, 18) == 33
    for n in range(1, 2000):
        assert L.vgo_adx_nibble_count_to_sample_count(L.vgo_adx_sample_count_to_nibble_count(n, 18), 18) == n


def test_silence_frame_hand_derived():
    # all-zero input: maxDistance 0 -> scale 1, scaleOut 0, gain 0 -> 18 zero bytes (Linear)
    out = po.adx_encode(np.zeros(32, np.int16), po.adx_params())
    assert out.tolist() == [0] * 18
    # Exponential: power 0 -> scaleToWrite 12 -> header 0x00 0x0C
    out = po.adx_encode(np.zeros(32, np.int16), po.adx_params(type=4))
    assert out[:2].tolist() == [0x00, 0x0C] and not out[2:].any()
    # Fixed filter 2: header gets Filter << 5
    out = po.adx_encode(np.zeros(32, np.int16), po.adx_params(type=2, filter=2))
    assert out[:2].tolist() == [0x40, 0x00]


def test_dc_frame_hand_derived():
    # constant 1000, version 4, no padding: history = pcm[0] = 1000 (CriAdxCodec.cs:69-74).
    # coefs (7400,-3342): pred = (1000*7400>>12) + (1000*-3342>>12) = 1806 + (-816) = 990 -> distance 10
    # scale = (10-1)/7+1 = 2, scaleOut 1, gain = 32767/10; first sample: raw 10 -> 32767 -> nibble 7
    p = po.adx_params()
    out = po.adx_encode(np.full(32, 1000, np.int16), p)
    assert p.history == 1000
    assert out[:2].tolist() == [0x00, 0x01]
    assert out[2] >> 4 == 7
    dec = po.adx_decode(out, 32, po.adx_params(history=1000))
    assert np.abs(dec.astype(int) - 1000).max() <= 8


@pytest.mark.parametrize("type_,filt", [(3, 0), (4, 0), (2, 0), (2, 1), (2, 2), (2, 3)])
@pytest.mark.parametrize("version", [3, 4])
def test_roundtrip_error_bounded(type_, filt, version):
    pcm = synth.generate(2, 32 * 400 + 7)
    for c in range(2):
        p = po.adx_params(type=type_, filter=filt, version=version)
        enc = po.adx_encode(pcm[c], p)
        assert len(enc) == 401 * 18
        dec = po.adx_decode(enc, pcm.shape[1], po.adx_params(type=type_, filter=filt, version=version,
                                                              history=p.history))
        err = dec.astype(int) - pcm[c].astype(int)
        # 4-bit ADPCM with a fitted scale: error stays far below the signal
        assert np.sqrt(np.mean(err ** 2.0)) < 0.12 * np.sqrt(np.mean(pcm[c].astype(float) ** 2)) + 8


def test_padding_skips_whole_frames_and_decodes_back():
    pcm = synth.generate(1, 1000)[0]
    p = po.adx_params(padding=70)               # 2 whole frames of padding + 6 samples
    enc = po.adx_encode(pcm, p)
    assert len(enc) == -(-1070 // 32) * 18
    assert not enc[:36].any()                   # skipped frames stay zero (CriAdxCodec.cs:86)
    dec = po.adx_decode(enc, 1000, po.adx_params(padding=70))
    # the reference's decoder stops frameCount frames in: the tail it never reaches stays 0
    reached = (-(-1000 // 32)) * 32 - 6
    err = dec[:reached - 32].astype(int) - pcm[:reached - 32].astype(int)
    assert np.abs(err).max() < 3000 and not dec[reached:].any()


def test_batch_matches_single():
    pcm = synth.generate(5, 32 * 50 + 3)
    out, hist = po.adx_encode_batch(pcm, po.adx_params(), threads=3)
    for c in range(5):
        p = po.adx_params()
        assert (po.adx_encode(pcm[c], p) == out[c]).all() and hist[c] == p.history == pcm[c, 0]
