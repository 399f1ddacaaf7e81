"""The HIP path against the COMMITTED golden vectors (tests/golden/codec_vectors.*): data that two separately
written restatements of the reference (oracle/liboracle.so and oracle/pyref) agreed on bit for bit before it was
written (tests/golden/make_codec_fixtures.py).  Nothing from oracle/ runs here: inputs and expected outputs are
read from the fixture, the kernels are driven through the C ABI by the reference-shaped mirror classes.
Also holds the exact-size BASELINE configs[0] test (1 mono channel x 48 kHz x 10 s)."""
import hashlib
import json
import os

import numpy as np
import pytest

from vgaudio_amd import synth
from vgaudio_amd.criadx import CriAdxCodec, CriAdxParameters
from vgaudio_amd.crihca import CriHcaDecoder, CriHcaFormat, CriHcaParameters
from vgaudio_amd.gcadpcm import (GcAdpcmCoefficients, GcAdpcmDecoder, GcAdpcmEncoder, GcAdpcmFormat, GcAdpcmParameters,
                                 Pcm16Format)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
Q = {"Highest": 1, "High": 2, "Middle": 3, "Low": 4, "Lowest": 5}


@pytest.fixture(scope="module")
def vec():
    return np.load(os.path.join(GOLD, "codec_vectors.npz")), json.load(open(os.path.join(GOLD, "codec_vectors.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_gc_golden_vectors(vec):
    arrays, manifest = vec
    names = manifest["gc"]["signals"]
    n = manifest["gc"]["sample_count"]
    pcm = [arrays[f"gc_{k}_pcm"] for k in names]
    coefs = GcAdpcmCoefficients.CalculateCoefficients(pcm)
    for i, k in enumerate(names):
        assert coefs[i].tolist() == arrays[f"gc_{k}_coefs"].tolist(), k
    enc = GcAdpcmEncoder.Encode(pcm, coefs)
    for i, k in enumerate(names):
        assert np.array_equal(enc[i], arrays[f"gc_{k}_adpcm"]), k
    dec = GcAdpcmDecoder.Decode(enc, coefs, GcAdpcmParameters(SampleCount=n))
    for i, k in enumerate(names):
        assert np.array_equal(dec[i], arrays[f"gc_{k}_decoded"]), k
    # one channel at a time as well (small batches take the time-piece / single-workgroup paths)
    for k in names:
        c = GcAdpcmCoefficients.CalculateCoefficients(arrays[f"gc_{k}_pcm"])
        assert c.tolist() == arrays[f"gc_{k}_coefs"].tolist(), k
        assert np.array_equal(GcAdpcmEncoder.Encode(arrays[f"gc_{k}_pcm"], c), arrays[f"gc_{k}_adpcm"]), k
    # coefficients that wrap int32 in the predictor (cold paths of the encoder)
    got = GcAdpcmEncoder.Encode(arrays["gc_noise_fs_pcm"], arrays["gc_hostile_coefs"])
    assert np.array_equal(got, arrays["gc_hostile_adpcm"])


def test_gc_config0_exact_size(vec):
    """BASELINE configs[0]: single mono 48 kHz 10 s PCM16 -> GC-ADPCM, through GcAdpcmFormat.EncodeFromPcm16
    (one vga_gcadpcm_encode_batch call with nch = 1) and back; digests from the fixture."""
    arrays, manifest = vec
    m = manifest["gc_config0"]
    x = synth.generate(1, m["sample_count"])[0]
    assert _sha(x) == m["input_sha256"]
    fmt = GcAdpcmFormat().EncodeFromPcm16(Pcm16Format([x], 48000))
    ch = fmt.Channels[0]
    assert ch.Coefs.tolist() == m["coefs"]
    adpcm = ch.GetAdpcmAudio()
    assert len(adpcm) == m["adpcm_bytes"]
    assert np.array_equal(adpcm[:256], arrays["gc_config0_adpcm_head"]) and np.array_equal(adpcm[-256:], arrays["gc_config0_adpcm_tail"])
    assert _sha(adpcm) == m["adpcm_sha256"]
    back = fmt.ToPcm16()
    assert _sha(back.Channels[0]) == m["decoded_sha256"]


def test_adx_golden_vectors(vec):
    arrays, manifest = vec
    names = manifest["adx"]["signals"]
    n = manifest["adx"]["sample_count"]
    pcm = [arrays[f"adx_{k}_pcm"] for k in names]
    m = dict(type="Type", filter="Filter", version="Version", frame_size="FrameSize", padding="Padding", sample_rate="SampleRate")
    for k, case in enumerate(manifest["adx"]["cases"]):
        kw = {m[a]: b for a, b in case["params"].items()}
        cfg = CriAdxParameters(**kw)
        enc = CriAdxCodec.Encode(pcm, cfg)
        for i, name in enumerate(names):
            assert np.array_equal(enc[i], arrays[f"adx_{k}_{name}_bytes"]), (kw, name)
            assert int(np.atleast_1d(cfg.History)[i]) == case["history"][name], (kw, name)
        dkw = {a: b for a, b in kw.items() if a != "Filter"}
        dec = CriAdxCodec.Decode(enc, n, CriAdxParameters(**dkw))
        for i, name in enumerate(names):
            assert np.array_equal(dec[i], arrays[f"adx_{k}_{name}_decoded"]), (kw, name)


def test_hca_golden_vectors(vec):
    arrays, manifest = vec
    for case in manifest["hca"]["cases"]:
        x = arrays[f"hca_{case['name']}_pcm"]
        pcm = Pcm16Format(list(x), 48000)
        if case.get("looping"):
            pcm.Looping, pcm.LoopStart, pcm.LoopEnd = True, case["loop_start"], case["loop_end"]
        fmt = CriHcaFormat().EncodeFromPcm16(pcm, CriHcaParameters(Quality=Q[case["quality"]]))
        for k, v in case["info"].items():
            assert getattr(fmt.Hca.c, k) == v, (case["name"], k)
        want = arrays[f"hca_{case['name']}_frames"]
        assert fmt.AudioData.shape == want.shape
        bad = np.argwhere(fmt.AudioData != want)
        assert bad.size == 0, (case["name"], bad[0].tolist(), len(bad))
        dec = CriHcaDecoder.Decode(fmt.Hca, want)
        assert np.array_equal(np.stack(dec), arrays[f"hca_{case['name']}_decoded"]), case["name"]


def test_full_length_digests_of_one_channel_and_one_stream():
    """60 s of audio (2 880 000 samples: 205 714 GC frames, 90 000 ADX frames, 2 813 HCA frames) through every codec, against
    digests both restatements agreed on (tests/golden/make_full_length_digests.py): channel 0 of configs[1] and configs[2],
    stream 0 of configs[3]."""
    d = json.load(open(os.path.join(GOLD, "full_length_digests.json")))
    n = d["sample_count"]
    x = synth.generate(2, n)
    assert _sha(x[0]) == d["gc"]["input_sha256"] and _sha(x) == d["hca"]["input_sha256"]
    coefs = GcAdpcmCoefficients.CalculateCoefficients(x[0])
    assert [int(v) for v in coefs] == d["gc"]["coefs"]
    adpcm = GcAdpcmEncoder.Encode(x[0], coefs)
    assert _sha(adpcm) == d["gc"]["adpcm_sha256"]
    assert _sha(GcAdpcmDecoder.Decode(adpcm, coefs, GcAdpcmParameters(SampleCount=n))) == d["gc"]["decoded_sha256"]
    cfg = CriAdxParameters()
    bytes_ = CriAdxCodec.Encode(x[0], cfg)
    assert _sha(bytes_) == d["adx"]["bytes_sha256"] and int(cfg.History) == d["adx"]["history"]
    assert _sha(CriAdxCodec.Decode(bytes_, n, CriAdxParameters())) == d["adx"]["decoded_sha256"]
    fmt = CriHcaFormat().EncodeFromPcm16(Pcm16Format(list(x), 48000), CriHcaParameters())
    assert fmt.AudioData.shape == (d["hca"]["frame_count"], d["hca"]["frame_size"]) and _sha(fmt.AudioData) == d["hca"]["frames_sha256"]
    dec = CriHcaDecoder.Decode(fmt.Hca, fmt.AudioData)
    assert _sha(np.stack(dec).astype(np.int16)) == d["hca"]["decoded_sha256"]
