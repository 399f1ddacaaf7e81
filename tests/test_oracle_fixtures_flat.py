"""The flat export of the golden fixtures (tests/golden/codec_vectors.bin + codec_vectors_index.json, read by
tools/dotnet_check/CheckVectors.cs) holds exactly the arrays of codec_vectors.npz; and the full-length digests
(tests/golden/full_length_digests.json: one configs[1] / configs[2] channel, one configs[3] stream, 2 880 000 samples) are
what the C restatement produces (the Python restatement agreed when the file was written: make_full_length_digests.py)."""
import hashlib
import json
import os

import numpy as np

from oracle import pyoracle as po
from vgaudio_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_flat_export_equals_the_npz():
    z = np.load(os.path.join(GOLD, "codec_vectors.npz"))
    index = json.load(open(os.path.join(GOLD, "codec_vectors_index.json")))["arrays"]
    blob = open(os.path.join(GOLD, "codec_vectors.bin"), "rb").read()
    assert sorted(index) == sorted(z.files)
    for name, e in index.items():
        a = np.frombuffer(blob, dtype=np.dtype(e["dtype"]).newbyteorder("<"), count=int(np.prod(e["shape"], dtype=np.int64)),
                          offset=e["offset"]).reshape(e["shape"])
        assert e["offset"] % 8 == 0 and np.array_equal(a, z[name]) and a.dtype.name == z[name].dtype.name, name


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_full_length_digests_match_the_c_restatement():
    d = json.load(open(os.path.join(GOLD, "full_length_digests.json")))
    n = d["sample_count"]
    assert n == 2_880_000
    x = synth.generate(2, n)
    assert _sha(x[0]) == d["gc"]["input_sha256"] and _sha(x) == d["hca"]["input_sha256"]
    coefs = po.gc_calculate_coefficients(x[0])
    adpcm = po.gc_encode(x[0], coefs)
    assert [int(v) for v in coefs] == d["gc"]["coefs"] and _sha(adpcm) == d["gc"]["adpcm_sha256"]
    assert _sha(po.gc_decode(adpcm, coefs, n)) == d["gc"]["decoded_sha256"]
    p = po.adx_params()
    w = po.adx_encode(x[0], p)
    assert _sha(w) == d["adx"]["bytes_sha256"] and int(p.history) == d["adx"]["history"]
    assert _sha(po.adx_decode(w, n, po.adx_params())) == d["adx"]["decoded_sha256"]
    rc, info, frames = po.hca_encode(x, po.hca_params(2, n))
    assert rc == 0 and _sha(frames) == d["hca"]["frames_sha256"] and info.frame_count == d["hca"]["frame_count"]
    rc, dec = po.hca_decode(info, frames)
    assert rc == 0 and _sha(np.asarray(dec, np.int16)) == d["hca"]["decoded_sha256"]
