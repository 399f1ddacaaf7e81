"""Oracle pins for the ADX / HCA encryption passes and key derivations.  The reference has
no tests for them (parity unpinned by reference vectors): the restatement (oracle/crypt_oracle.c) is pinned by
properties derived by hand from the cited lines.  The key derivations are host code in the product too and are
compared here; the byte passes are compared on the GPU (tests/test_gpu_crypt.py)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po
from vgaudio_amd import _lib


def primes_between(lo, hi):
    return [n for n in range(lo, hi) if all(n % d for d in range(2, int(n ** 0.5) + 1))]


def test_adx_key_from_string_follows_the_primes_table():
    primes = primes_between(0x4000, 0x8000)[:0x400]                           # BuildPrimesTable (CriAdxKey.cs:58-65)
    assert primes[0] == 16411
    for s in ("", "a", "karaage", "(C)2005 MOSS LTD. BMW Z4", "3x5k62bg9ptbwy"):
        seed, mult, inc = (primes[0x100], primes[0x200], primes[0x300]) if s else (0, 0, 0)
        for ch in s:
            p = primes[ord(ch) + 0x80]
            seed, mult, inc = primes[seed * p % 0x400], primes[mult * p % 0x400], primes[inc * p % 0x400]
        k = po.adx_key_from_string(s) if s else po.AdxKey(0, 0, 0)
        assert (k.seed, k.mult, k.inc) == (seed, mult, inc)
        pk = _lib.AdxKeyC()
        assert _lib.lib().vga_adx_key_from_string(s.encode(), C.byref(pk)) == 0
        assert (pk.seed, pk.mult, pk.inc) == (seed, mult, inc)
    # the triple public ADX key lists give for this string
    k = po.adx_key_from_string("karaage")
    assert (k.seed, k.mult, k.inc) == (0x49e1, 0x4a57, 0x553d)


def test_adx_key_code_round_trip():
    rng = np.random.default_rng(0)
    for code in [1, 12160794, 19910623, 416383518, 683461999, 268736153152] + [int(x) for x in rng.integers(1, 1 << 42, 200)]:
        k = po.adx_key_from_code(code)
        v = code - 1
        assert (k.seed, k.mult, k.inc) == (v >> 27 & 0x7fff, (v >> 12 & 0x7ffc) | 1, (v << 1 & 0x7fff) | 1)
        pk = _lib.AdxKeyC()
        assert _lib.lib().vga_adx_key_from_code(code, C.byref(pk)) == 0
        assert (pk.seed, pk.mult, pk.inc) == (k.seed, k.mult, k.inc)
        assert _lib.lib().vga_adx_key_code(C.byref(pk)) == po.adx_key_code(k)
        # KeyCode keeps seed, mult's bits 2..14 and inc's bits 1..14: re-deriving from it reproduces the key
        k2 = po.adx_key_from_code(po.adx_key_code(k))
        assert (k2.seed, k2.mult, k2.inc) == (k.seed, k.mult, k.inc)


def test_adx_crypt_is_an_involution_and_skips_empty_frames():
    rng = np.random.default_rng(1)
    audio = [rng.integers(0, 256, 18 * 50).astype(np.uint8) for _ in range(3)]
    audio[1][18 * 7:18 * 8] = 0                                               # an empty frame stays empty (FrameNotEmpty)
    key = po.adx_key_from_string("morio")
    enc = po.adx_crypt(audio, key, 8)
    assert not enc[1][18 * 7:18 * 8].any()
    for a, e in zip(audio, enc):
        assert np.array_equal(a.reshape(-1, 18)[:, 2:], e.reshape(-1, 18)[:, 2:])      # only the two scale bytes change
        assert not np.array_equal(a, e)
    dec = po.adx_crypt(enc, key, 8)
    for a, d in zip(audio, dec):
        assert np.array_equal(a, d)
    # the LCG advances once per (frame, channel) slot, channels first (:19-38)
    x, slots = key.seed, {}
    for frame in range(50):
        for ch in range(3):
            slots[(frame, ch)] = x
            x = (x * key.mult + key.inc) & 0x7fff
    for (frame, ch), x in slots.items():
        if frame == 7 and ch == 1:
            continue
        assert enc[ch][18 * frame] == audio[ch][18 * frame] ^ (x >> 8) and enc[ch][18 * frame + 1] == audio[ch][18 * frame + 1] ^ (x & 0xff)
    nine = po.adx_crypt(audio, key, 9)
    assert all((e.reshape(-1, 18)[:, 0] <= 0x1f).all() for e in nine)                 # type 9 masks the first byte (:33)


def test_adx_test_key():
    rng = np.random.default_rng(2)
    pcm = rng.integers(-3000, 3000, (2, 32 * 200)).astype(np.int16)
    audio, hist = po.adx_encode_batch(pcm, po.adx_params())
    key, other = po.adx_key_from_string("mituba"), po.adx_key_from_string("GHM")
    enc = po.adx_crypt(list(audio), key, 8)
    assert po.adx_test_key(enc, key, 8) == 1
    assert po.adx_test_key(enc, other, 8) == 0
    assert po.adx_test_key(list(audio), po.AdxKey(0, 0, 0), 8) == 1                    # plain scales are < 0x2000: the null key fits


@pytest.mark.parametrize("key_type,code", [(0, 0), (1, 0), (56, 1), (56, 0xCC55463930DBE1AB), (56, 2424), (56, 0x7FFFFFFFFFFFFFFF)])
def test_hca_tables_are_permutations_and_host_matches(key_type, code):
    rc, dec, enc = po.hca_key_tables(key_type, code)
    assert rc == 0
    assert sorted(dec.tolist()) == list(range(256)) and dec[0] == 0 and dec[255] == 255   # ShuffleTable fixes 0 and 0xFF
    assert (dec[enc] == np.arange(256)).all() and (enc[dec] == np.arange(256)).all()
    if key_type == 0:
        assert (dec == np.arange(256)).all()
    pd, pe = np.zeros(256, np.uint8), np.zeros(256, np.uint8)
    assert _lib.lib().vga_hca_key_tables(key_type, code, pd.ctypes.data_as(_lib.u8p), pe.ctypes.data_as(_lib.u8p)) == 0
    assert np.array_equal(pd, dec) and np.array_equal(pe, enc)


def test_hca_type1_table_by_hand():
    x, out = 0, [0]                                                            # CreateDecryptionTableType1 (:80-100)
    for _ in range(256):
        x = (x * 13 + 11) % 256
        if x not in (0, 0xff):
            out.append(x)
    out.append(0xff)
    rc, dec, enc = po.hca_key_tables(1)
    assert dec.tolist() == out
    assert _lib.lib().vga_hca_key_tables(7, 0, dec.ctypes.data_as(_lib.u8p), enc.ctypes.data_as(_lib.u8p)) == _lib.ArgumentOutOfRangeError.code


def test_hca_crypt_round_trip_and_crc():
    rng = np.random.default_rng(3)
    pcm = rng.integers(-8000, 8000, (2, 6000)).astype(np.int16)
    rc, info, frames = po.hca_encode(pcm, po.hca_params(2, 6000))
    rc, dec, enc = po.hca_key_tables(56, 765765765765765)
    e = po.hca_crypt(frames, info.frame_size, enc).reshape(info.frame_count, info.frame_size)
    for f in e:
        assert po.lib().vgo_crc16(po._u8(np.ascontiguousarray(f)), info.frame_size) == 0   # CRC refreshed (:29-31)
    assert not np.array_equal(e, frames)
    d = po.hca_crypt(e, info.frame_size, dec).reshape(info.frame_count, info.frame_size)
    assert np.array_equal(d, frames)
    assert np.array_equal(e[:, :-2], enc[frames[:, :-2]])


# ---------------------------------------------------------------- key searches (remainder)
def test_hca_find_key_finds_the_key_that_encrypted_the_stream():
    """CriHcaEncryption.FindKey (CriHcaEncryption.cs:34-88): encrypt a real stream, offer the right key among wrong ones."""
    from vgaudio_amd import synth
    n = 1024 * 14
    x = synth.generate(2, n)
    x[:, :3000] = 0                                            # leading silence: FindFirstNonEmptyFrame has work to do
    rc, info, frames = po.hca_encode(x, po.hca_params(2, n))
    assert rc == 0
    codes = [0x1234, 12345678901234, 0xCC55463930DBE1AB, 765765765765765, 2]
    tables = [po.hca_key_tables(56, c)[1] for c in codes]
    enc_table = po.hca_key_tables(56, codes[2])[2]
    enc = po.hca_crypt(frames, info.frame_size, enc_table).reshape(frames.shape)
    assert po.hca_find_key(info, enc, np.stack(tables)) == 2
    assert po.hca_find_key(info, enc, np.stack(tables[:2] + tables[3:])) == -1
    assert po.hca_find_key(info, enc, np.stack([tables[2]])) == 0
    # an unencrypted stream is explained by the identity (type 0) table
    ident = po.hca_key_tables(0)[1]
    assert po.hca_find_key(info, frames, np.stack([tables[0], ident])) == 1
    # a wrong sync word is the reference's InvalidDataException
    bad = enc.copy()
    bad[4, 0] = 0x12
    assert po.hca_find_key(info, bad, np.stack(tables)) == -3


def test_adx_guess_keys_recovers_the_key_from_scales():
    """GuessAdx (VGAudio.Tools/CrackAdx/GuessAdx.cs:118-218) on reduced candidate lists: the true key is found, every
    key returned explains every scale, and a start frame > 0 is traced back to the starting seed."""
    from vgaudio_amd import synth
    rng = np.random.default_rng(8)
    pcm = synth.generate(1, 32 * 400)[0]
    audio = po.adx_encode(pcm, po.adx_params())
    mults, incs = po.adx_guess_default_candidates(8)
    assert len(mults) == 0x400 and len(incs) == 0x400 and mults[0] == 16411 and (mults == incs).all()   # the first prime after 0x4000
    m9, i9 = po.adx_guess_default_candidates(9)
    assert len(m9) == 2048 and len(i9) == 4096 and m9[0] == 1 and m9[1] == 5 and i9[1] == 3
    for start_silence in (0, 5):
        a = audio.copy()
        if start_silence:
            a[:18 * start_silence] = 0
        key = po.adx_key_from_string("crack me")
        enc = po.adx_crypt([a], key, 8)[0]
        scales = (enc.reshape(-1, 18)[:, 0].astype(np.uint16) << 8) | enc.reshape(-1, 18)[:, 1]
        start = int(np.flatnonzero(enc)[0]) // 18
        assert start == start_silence
        sub_m = np.unique(np.concatenate([rng.choice(mults, 40, replace=False), [key.mult]])).astype(np.int32)
        sub_i = np.unique(np.concatenate([rng.choice(incs, 40, replace=False), [key.inc]])).astype(np.int32)
        keys = po.adx_guess_keys(scales, start, 8, sub_m, sub_i)
        assert (key.seed, key.mult, key.inc) in keys, (start_silence, keys[:5])
        assert keys == sorted(set(keys))
        for seed, mult, inc in keys:                            # KeyIsValid: every key explains every scale
            k = po.AdxKey(seed, mult, inc)
            assert po.adx_test_key([enc], k, 8) == 1


def test_hca_byte_position_counts():
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, (3, 7 * 100)).astype(np.uint8)
    counts = po.hca_byte_position_counts(frames, 100, positions=30)
    assert counts.sum() == 30 * 21
    f = frames.reshape(3, 7, 100)
    for p in (0, 13, 29):
        assert (counts[p] == np.bincount(f[:, :, p].reshape(-1), minlength=256)).all()
