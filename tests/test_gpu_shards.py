"""BASELINE configs[4] on one GPU, shard by shard: 32768 mono channels x 48 kHz x 60 s sharded over 8 GPUs means rank r
encodes channels [4096 r, 4096 r + 4096).  Here one GPU encodes each of the eight shards in turn; sampled channels of
every shard are encoded by the oracle in the test and compared bit for bit, and EVERY channel of the shard (bitstream +
coefficients) is held to the digests the ORACLE produced for it in the build container, without a GPU
(tests/golden/gc_shard_oracle_digests.json + gc_channel_oracle_digests.npy, written by
tests/golden/make_gc_shard_oracle_digests.py; provenance "oracle") -- the values `bench.py --gpus 8` checks every rank's
output against.  tests/golden/gc_shard_digests.json is the same list as a round-3 GPU wrote it (kept: the two must agree).

    VGA_WRITE_SHARD_DIGESTS=1 python -m pytest tests/test_gpu_shards.py -m gpu      (re)writes gpurun_out/gc_shard_digests.json
"""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "gc_shard_oracle_digests.json")
GOLD_CHANNELS = os.path.join(ROOT, "tests", "golden", "gc_channel_oracle_digests.npy")
GOLD_GPU = os.path.join(ROOT, "tests", "golden", "gc_shard_digests.json")
CHANNELS, SAMPLES, SHARDS = 4096, 2_880_000, 8
SAMPLED = (0, 511, 1024, 1999, 2048, 3071, 3500, 4095)


def test_every_shard_of_config5_matches_oracle_samples_and_committed_digests():
    import torch
    from oracle import pyoracle as po
    from vgaudio_amd import _lib, device as vdev, distributed as vdist, synth
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    nb = vdev.gc_byte_count(SAMPLES)
    pcm = vdev.alloc_pcm(CHANNELS, SAMPLES, dev)
    adpcm = vdev.alloc_adpcm(CHANNELS, SAMPLES, dev)
    ws = torch.empty(L.vga_gcadpcm_coefs_workspace_bytes(CHANNELS, SAMPLES), dtype=torch.uint8, device=dev)
    writing = os.environ.get("VGA_WRITE_SHARD_DIGESTS") == "1"
    gold = None if writing else json.load(open(GOLD))
    per_channel = None
    if gold is not None:
        assert gold["provenance"] == "oracle"
        assert gold["channels_per_shard"] == CHANNELS and gold["samples_per_channel"] == SAMPLES and len(gold["shards"]) == SHARDS
        per_channel = np.load(GOLD_CHANNELS)
        gpu_written = json.load(open(GOLD_GPU))["shards"]
        assert [dict(s) for s in gold["shards"]] == [dict(s) for s in gpu_written], "oracle-written and GPU-written digests differ"
    found = []
    for r in range(SHARDS):
        first = r * CHANNELS
        vdev.synth_pcm(CHANNELS, SAMPLES, dev, first_channel=first, out=pcm)
        coefs = vdev.gc_coefs(pcm, SAMPLES, workspace=ws)
        vdev.gc_encode(pcm, SAMPLES, coefs, out=adpcm)
        torch.cuda.synchronize()
        # sampled channels against the oracle (the generator on the host is bit-identical to the device's)
        host = np.stack([synth.generate(1, SAMPLES, first_channel=first + c)[0] for c in SAMPLED])
        assert np.array_equal(host, pcm[list(SAMPLED), :SAMPLES].cpu().numpy()), r
        want_coefs, want_adpcm = po.gc_encode_batch(host, threads=8)
        assert np.array_equal(np.asarray(want_coefs).reshape(len(SAMPLED), 16), coefs[list(SAMPLED)].cpu().numpy().reshape(len(SAMPLED), 16)), r
        assert np.array_equal(np.asarray(want_adpcm)[:, :nb], adpcm[list(SAMPLED), :nb].cpu().numpy()), r
        # the whole shard: the 64-bit positional digest the gather uses, and SHA-256 over coefficients then rows
        digest = vdist.rows_digest(adpcm, nb, coefs, first)
        sha = hashlib.sha256()
        sha.update(coefs.cpu().numpy().tobytes())
        for c0 in range(0, CHANNELS, 512):
            sha.update(adpcm[c0:c0 + 512, :nb].cpu().numpy().tobytes())
        found.append({"rank": r, "first_channel": first, "rows_digest": "0x%016x" % digest, "sha256": sha.hexdigest()})
        if gold is not None:
            inner = vdist.row_digests(adpcm, nb, coefs).cpu().numpy().view(np.uint64)
            differ = np.nonzero(inner != per_channel[first:first + CHANNELS])[0]
            assert differ.size == 0, "shard %d: channels that differ from the oracle: %s" % (r, (first + differ[:16]).tolist())
            assert found[-1] == gold["shards"][r], (found[-1], gold["shards"][r])
    if writing:
        out = {"what": "GC-ADPCM coefficients + bitstream of every shard of BASELINE configs[4] (vgaudio_amd.synth channels "
                       "first_channel .. first_channel + 4095, 2 880 000 samples each), written by tests/test_gpu_shards.py on one MI355X; "
                       "rows_digest = vgaudio_amd.distributed.rows_digest, sha256 over the int16 coefficients [4096][16] followed by the "
                       "rows' data bytes [4096][byte count]",
               "channels_per_shard": CHANNELS, "samples_per_channel": SAMPLES, "bytes_per_row": nb, "shards": found}
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gc_shard_digests.json"), "w"), indent=1)
