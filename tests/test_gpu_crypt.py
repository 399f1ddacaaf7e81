"""GPU parity for the ADX / HCA encryption passes: the device kernels against the oracle's
literal loops, through the C ABI, including encrypted container files."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import pyoracle as po
import vgaudio_amd
from vgaudio_amd import _lib, synth
from vgaudio_amd.adx import AdxConfiguration, AdxWriter
from vgaudio_amd.criadx import CriAdxEncryption, CriAdxFormat, CriAdxKey, CriAdxParameters
from vgaudio_amd.crihca import CriHcaEncryption, CriHcaFormat, CriHcaKey, CriHcaParameters
from vgaudio_amd.gcadpcm import Pcm16Format
from vgaudio_amd.hca import HcaConfiguration, HcaWriter

pytestmark = pytest.mark.gpu


def okey(k):
    return po.AdxKey(k.Seed, k.Mult, k.Inc)


@pytest.mark.parametrize("nch,frames", [(1, 1), (1, 1000), (2, 257), (3, 4096), (7, 100), (64, 3000)])
@pytest.mark.parametrize("etype", [8, 9])
def test_adx_crypt_matches_oracle(nch, frames, etype):
    rng = np.random.default_rng(nch * 100 + frames)
    audio = [rng.integers(0, 256, 18 * frames).astype(np.uint8) for _ in range(nch)]
    for a in audio:                                                            # some empty frames, some with a lone byte
        for f in rng.integers(0, frames, max(frames // 10, 1)):
            a[18 * f:18 * f + 18] = 0
        f = int(rng.integers(0, frames))
        a[18 * f:18 * f + 18] = 0
        a[18 * f + 17] = 1
    key = CriAdxKey("sakakit4649")
    want = po.adx_crypt(audio, okey(key), etype)
    got = [a.copy() for a in audio]
    CriAdxEncryption.EncryptDecrypt(got, key, etype, 18)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_adx_crypt_arbitrary_lcg_parameters():
    """CriAdxKey(seed, mult, inc) with values outside 15 bits: the first slot uses the raw seed (:18, :31-34)."""
    rng = np.random.default_rng(5)
    audio = [rng.integers(1, 256, 18 * 300).astype(np.uint8) for _ in range(2)]
    for seed, mult, inc in ((0x12345, 0x46b1, 0x62ad), (0x7fff, 0x10001, 3), (0, 1, 0), (0x4d06, 0x663b, 0x7d09), (-5, 77777, -12345)):
        key = CriAdxKey(seed, mult, inc)
        want = po.adx_crypt(audio, okey(key), 8)
        got = [a.copy() for a in audio]
        CriAdxEncryption.EncryptDecrypt(got, key, 8, 18)
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (seed, mult, inc)


def test_adx_other_frame_sizes_and_errors():
    rng = np.random.default_rng(6)
    for fs in (4, 10, 34):
        audio = [rng.integers(0, 256, fs * 77).astype(np.uint8) for _ in range(2)]
        key = CriAdxKey(19910623)
        want = po.adx_crypt(audio, okey(key), 9, fs)
        got = [a.copy() for a in audio]
        CriAdxEncryption.EncryptDecrypt(got, key, 9, fs)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
    with pytest.raises(_lib.ArgumentError):
        CriAdxEncryption.EncryptDecrypt([np.zeros(19, np.uint8)], CriAdxKey(1), 8, 18)      # not whole frames


def test_adx_find_key():
    pcm = synth.generate(2, 32 * 3000)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    names = ["GHM", "GHMSC", "karaage", "mituba", "morio", "ranatus"]
    keys = [CriAdxKey(n) for n in names] + [CriAdxKey(0x4133, 0x5a01, 0x5723)]
    for target in (2, 5, 6):
        enc = [np.array(c.Audio, dtype=np.uint8, copy=True) for c in fmt.Channels]
        CriAdxEncryption.EncryptDecrypt(enc, keys[target], 8, 18)
        found = CriAdxEncryption.FindKey(enc, 8, 18, keys)
        want = next((k for k in keys if po.adx_test_key(enc, okey(k), 8)), None)
        assert found is want and found is keys[target]
    enc = [np.array(c.Audio, dtype=np.uint8, copy=True) for c in fmt.Channels]
    CriAdxEncryption.EncryptDecrypt(enc, CriAdxKey("CS-GGNX+"), 8, 18)
    assert CriAdxEncryption.FindKey(enc, 8, 18, keys) is None
    codes = [CriAdxKey(c) for c in (12160794, 19910623, 416383518)]
    enc9 = [np.array(c.Audio, dtype=np.uint8, copy=True) for c in fmt.Channels]
    CriAdxEncryption.EncryptDecrypt(enc9, codes[1], 9, 18)
    found = CriAdxEncryption.FindKey(enc9, 9, 18, codes)
    want = next((k for k in codes if po.adx_test_key(enc9, okey(k), 9)), None)
    assert found is want


def test_adx_encrypted_file():
    pcm = synth.generate(2, 20000)
    fmt = CriAdxFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000))
    key = CriAdxKey("morio")
    cfg = AdxConfiguration(EncryptionKey=key, EncryptionType=8)
    f = AdxWriter(cfg).GetFile(fmt)
    enc = po.adx_crypt([c.Audio for c in fmt.Channels], okey(key), 8)
    p = po.adxfile_params(48000, 20000, encryption_type=8)
    rc, want = po.adxfile_write(enc, [c.History for c in fmt.Channels], p)
    assert rc == 0 and f == want.tobytes()
    assert f[19] == 8
    plain = AdxWriter().GetFile(fmt)                                           # the format's own audio was not touched (:123-127)
    assert plain != f and po.adxfile_read(plain)[0] == 0


@pytest.mark.parametrize("key", [CriHcaKey.Type0, CriHcaKey.Type1, 0xCC55463930DBE1AB, 88888888])
def test_hca_crypt_matches_oracle(key):
    pcm = synth.generate(2, 30000)
    fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(pcm), 48000), CriHcaParameters())
    k = CriHcaKey(key)
    frames = np.ascontiguousarray(fmt.AudioData, dtype=np.uint8).copy()
    want = po.hca_crypt(frames, fmt.Hca.FrameSize, k.EncryptionTable).reshape(frames.shape)
    CriHcaEncryption.Crypt(fmt.Hca, frames, k, False)
    assert np.array_equal(frames, want)
    CriHcaEncryption.Crypt(fmt.Hca, frames, k, True)
    assert np.array_equal(frames, np.asarray(fmt.AudioData))


def test_hca_crypt_device_batch_and_odd_sizes():
    rng = np.random.default_rng(9)
    rc, dec, enc = po.hca_key_tables(56, 123456789)
    for ns, fc, fs in ((3, 17, 682), (1, 5, 3), (2, 9, 65), (4, 3, 4096), (2, 11, 1023)):
        pitch = fc * fs + 5
        host = rng.integers(0, 256, (ns, pitch)).astype(np.uint8)
        d = torch.from_numpy(host.copy()).cuda()
        _lib.check(_lib.lib().vga_hca_crypt_device(d.data_ptr(), pitch, ns, fc, fs, enc.ctypes.data_as(_lib.u8p),
                                                   torch.cuda.current_stream().cuda_stream))
        got = d.cpu().numpy()
        for s in range(ns):
            want = po.hca_crypt(host[s, :fc * fs], fs, enc)
            assert np.array_equal(got[s, :fc * fs], want)
            assert np.array_equal(got[s, fc * fs:], host[s, fc * fs:])
    assert _lib.lib().vga_hca_crypt_device(d.data_ptr(), pitch, 1, 1, 5000, enc.ctypes.data_as(_lib.u8p), None) == _lib.ArgumentError.code


def test_hca_encrypted_file():
    pcm = synth.generate(2, 20000)
    key = CriHcaKey(0x0123456789ABCDEF)
    f = HcaWriter(HcaConfiguration(EncryptionKey=key)).GetFile(Pcm16Format(list(pcm), 48000))
    rc, info, frames = po.hca_encode(pcm, po.hca_params(2, 20000))
    enc = po.hca_crypt(frames, info.frame_size, key.EncryptionTable)
    rc, want = po.hcafile_write(info, enc, encryption_type=56, encrypted_ids=True)
    assert rc == 0 and f == want.tobytes()
    assert f[:4] == bytes([0xC8, 0xC3, 0xC1, 0x00])
    rc, r, vol, etype, comment, ver = po.hcafile_read(f)                       # the reader masks the ids off again
    assert rc == 0 and etype == 56 and r.frame_count == info.frame_count


# ---------------------------------------------------------------- key searches (remainder)
def test_hca_find_key_matches_oracle():
    """CriHcaEncryption.FindKey / TestKey (CriHcaEncryption.cs:34-88) over candidate lists, against the oracle's literal loop."""
    from vgaudio_amd.crihca import CriHcaEncryption, CriHcaFormat, CriHcaKey, CriHcaParameters
    from vgaudio_amd.gcadpcm import Pcm16Format
    rng = np.random.default_rng(77)
    for nch, quality, n, silence in ((2, 2, 1024 * 14, 3000), (1, 5, 1024 * 12, 0), (2, 4, 1024 * 6, 1024 * 3), (2, 2, 1024 * 3, 0)):
        x = synth.generate(nch, n)
        x[:, :silence] = 0
        fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(x), 48000), CriHcaParameters(Quality=quality))
        q = {1: "Highest", 2: "High", 3: "Middle", 4: "Low", 5: "Lowest"}[quality]
        rc, info, frames = po.hca_encode(x, po.hca_params(nch, n, quality=q))
        assert rc == 0 and np.array_equal(frames, fmt.AudioData)
        codes = [int(c) for c in rng.integers(1, 2 ** 56, 12)]
        keys = [CriHcaKey(c) for c in codes]
        true = 7
        enc = fmt.AudioData.copy()
        CriHcaEncryption.Crypt(fmt.Hca, enc, keys[true], False)
        tables = np.stack([k.DecryptionTable for k in keys])
        want = po.hca_find_key(info, enc, tables)
        assert want == true
        got = CriHcaEncryption.FindKey(fmt.Hca, enc, keys)
        assert got is keys[true]
        # without the right key: whatever the literal loop says (normally none)
        others = keys[:true] + keys[true + 1:]
        want = po.hca_find_key(info, enc, np.stack([k.DecryptionTable for k in others]))
        got = CriHcaEncryption.FindKey(fmt.Hca, enc, others)
        assert (got is None and want == -1) or (want >= 0 and got is others[want])
        # unencrypted audio and the type-0 key
        ident = CriHcaKey(CriHcaKey.Type0)
        assert CriHcaEncryption.FindKey(fmt.Hca, fmt.AudioData, [keys[0], ident]) is ident
    bad = enc.copy()
    bad[1, 0] = 0x12
    with pytest.raises(vgaudio_amd.InvalidDataError):
        CriHcaEncryption.FindKey(fmt.Hca, bad, keys)
    # TestKey walks a key's frames in order and FindKey the keys in order (CriHcaEncryption.cs:34-63): a wrong sync word
    # in frame 2 throws only for a key that gets that far.  Wrong keys are rejected at their first frame, so a list
    # without the right key yields null; the right key throws; a key placed before it that unpacks everything wins first.
    bad = enc.copy()
    bad[2, 0] = 0x00                                    # 0x00 and 0xFF are fixed points of every substitution table
    others = keys[:true] + keys[true + 1:]
    assert po.hca_find_key(info, bad, np.stack([k.DecryptionTable for k in others])) == -1
    assert CriHcaEncryption.FindKey(fmt.Hca, bad, others) is None
    assert po.hca_find_key(info, bad, tables) == -3
    with pytest.raises(vgaudio_amd.InvalidDataError):
        CriHcaEncryption.FindKey(fmt.Hca, bad, keys)


def test_hca_find_key_random_frames_agree_with_oracle():
    """Random bytes behind a valid sync word: validity is decided by the bit walk alone; every key must get the oracle's verdict."""
    from vgaudio_amd.crihca import CriHcaEncoder, CriHcaEncryption, CriHcaKey, CriHcaParameters
    rng = np.random.default_rng(5)
    enc = CriHcaEncoder.InitializeNew(CriHcaParameters(ChannelCount=2, SampleRate=48000, SampleCount=1024 * 8, Quality=4))
    hca = enc.Hca
    rc, info = po.hca_init(po.hca_params(2, 1024 * 8, quality="Low"))
    frames = rng.integers(0, 256, (hca.FrameCount, hca.FrameSize)).astype(np.uint8)
    frames[:, :2] = 0xFF
    frames[:, 2] &= 0x0F                                        # small noise levels: more frames survive
    keys = [CriHcaKey(int(c)) for c in rng.integers(1, 2 ** 56, 40)] + [CriHcaKey(CriHcaKey.Type0), CriHcaKey(CriHcaKey.Type1)]
    for k in range(len(keys)):                                  # one candidate at a time: the verdict for every key
        want = po.hca_find_key(info, frames, keys[k].DecryptionTable[None, :])
        got = CriHcaEncryption.FindKey(hca, frames, [keys[k]])
        assert (want == 0) == (got is keys[k]), k


def test_adx_guess_keys_matches_oracle_and_finds_the_key():
    """GuessAdx (VGAudio.Tools/CrackAdx/GuessAdx.cs:118-218): reduced candidate lists against the oracle's literal loops,
    then the reference's full candidate sets (4096 x 1024 x 1024 triples for type 8) on the GPU alone."""
    from vgaudio_amd.criadx import AdxFile, CriAdxKey, GuessAdx
    rng = np.random.default_rng(21)
    pcm = synth.generate(1, 32 * 600)[0]
    audio = po.adx_encode(pcm, po.adx_params())
    for etype, key in ((8, CriAdxKey("crack me")), (9, CriAdxKey(0x123456789))):
        mults, incs = po.adx_guess_default_candidates(etype)
        gm, gi = GuessAdx.DefaultCandidates(etype)
        assert np.array_equal(mults, gm) and np.array_equal(incs, gi)
        for start_silence in (0, 4):
            a = audio.copy()
            a[:18 * start_silence] = 0
            enc = po.adx_crypt([a], po.AdxKey(key.Seed, key.Mult, key.Inc), etype)[0]
            f = AdxFile(enc, 18)
            assert f.StartFrame == start_silence
            sub_m = np.unique(np.concatenate([rng.choice(mults, 24, replace=False), [key.Mult]])).astype(np.int32)
            sub_i = np.unique(np.concatenate([rng.choice(incs, 24, replace=False), [key.Inc]])).astype(np.int32)
            want = po.adx_guess_keys(f.Scales, f.StartFrame, etype, sub_m, sub_i)
            got = [(k.Seed, k.Mult, k.Inc) for k in GuessAdx.Run(f, etype, sub_m, sub_i)]
            assert got == want and (key.Seed & (0x7fff if etype == 8 else 0x1fff), key.Mult, key.Inc) in [(s & (0x7fff if etype == 8 else 0x1fff), m, i) for s, m, i in got]
    # the full search, type 8
    key = CriAdxKey("crack me")
    enc = po.adx_crypt([audio], po.AdxKey(key.Seed, key.Mult, key.Inc), 8)[0]
    found = [(k.Seed, k.Mult, k.Inc) for k in GuessAdx.Run(AdxFile(enc, 18), 8)]
    assert (key.Seed, key.Mult, key.Inc) in found and len(found) < 50
    for s, m, i in found:
        assert po.adx_test_key([enc], po.AdxKey(s, m, i), 8) == 1


def test_hca_byte_position_counts_match_oracle():
    import ctypes as C
    import torch
    from vgaudio_amd import _lib
    rng = np.random.default_rng(2)
    ns, fc, fs = 5, 37, 100
    frames = rng.integers(0, 256, (ns, fc * fs + 12)).astype(np.uint8)
    d = torch.from_numpy(frames).cuda()
    counts = np.zeros((30, 256), dtype=np.uint32)
    _lib.check(_lib.lib().vga_hca_byte_position_counts_device(d.data_ptr(), frames.shape[1], ns, fc, fs, 30, counts.ctypes.data,
                                                              torch.cuda.current_stream().cuda_stream))
    want = po.hca_byte_position_counts(frames[:, :fc * fs], fs, 30)
    assert np.array_equal(counts, want)
