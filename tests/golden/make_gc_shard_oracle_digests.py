"""Writes tests/golden/gc_shard_oracle_digests.json + gc_channel_oracle_digests.npy: the ORACLE's GC-ADPCM output for
every channel of BASELINE configs[1] / configs[4] (32 768 synthetic channels x 2 880 000 samples = eight shards of 4096).

    python tests/golden/make_gc_shard_oracle_digests.py [--threads 6] [--shards 0,1,...] [--samples 2880000]

Runs in the build container, no GPU, nothing of vgaudio_amd's HIP library: PCM from oracle/synth_oracle.c (held to
vgaudio_amd/synth.py by tests/test_oracle_gcadpcm.py and, below, on a few channels per shard), coefficients + bitstream
from oracle/gcadpcm_oracle.c (`vgo_gc_encode_batch` = GcAdpcmFormat.EncodeFromPcm16's per-channel work,
Formats/GcAdpcm/GcAdpcmFormat.cs:58-74,129-135).  Per shard: the 64-bit positional digest the final gather uses
(vgaudio_amd.distributed.rows_digest -- plain torch integer arithmetic on the CPU), SHA-256 over the coefficients then
the rows, and one 64-bit digest per channel (row_digests) so that a GPU run that differs names the channel.
About 0.95 core-seconds per channel: 65 min for all eight shards on 8 cores.  Resumable: a finished shard is kept.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT_JSON = os.path.join(ROOT, "tests", "golden", "gc_shard_oracle_digests.json")
OUT_NPY = os.path.join(ROOT, "tests", "golden", "gc_channel_oracle_digests.npy")
CHANNELS, SHARDS = 4096, 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--shards", default=",".join(str(r) for r in range(SHARDS)))
    ap.add_argument("--samples", type=int, default=2_880_000)
    ap.add_argument("--chunk", type=int, default=256)
    a = ap.parse_args()
    import torch
    from oracle import pyoracle as po
    from vgaudio_amd import distributed as vdist, synth
    n = a.samples
    nb = po.gc_sample_count_to_byte_count(n)
    pitch = (nb + 15) // 16 * 16
    state = {"shards": {}}
    if os.path.exists(OUT_JSON):
        old = json.load(open(OUT_JSON))
        if old.get("samples_per_channel") == n:
            state["shards"] = {str(s["rank"]): s for s in old["shards"]}
    per_channel = np.load(OUT_NPY) if os.path.exists(OUT_NPY) and state["shards"] else np.zeros(CHANNELS * SHARDS, dtype=np.uint64)
    rows = np.zeros((CHANNELS, pitch), dtype=np.uint8)
    coefs = np.zeros((CHANNELS, 16), dtype=np.int16)
    pcm = np.empty((a.chunk, n), dtype=np.int16)
    for r in [int(x) for x in a.shards.split(",") if x != ""]:
        if str(r) in state["shards"]:
            print("shard", r, "already done", flush=True)
            continue
        first = r * CHANNELS
        t0 = time.time()
        for c0 in range(0, CHANNELS, a.chunk):
            k = min(a.chunk, CHANNELS - c0)
            po.synth_generate(k, n, first_channel=first + c0, threads=a.threads, out=pcm)
            if c0 % 1024 == 0:      # the C generator against its definition, one channel per 1024
                assert np.array_equal(pcm[7], synth.generate(1, n, first_channel=first + c0 + 7)[0]), (r, c0)
            wc, wa = po.gc_encode_batch(pcm[:k], threads=a.threads)
            coefs[c0:c0 + k] = wc
            rows[c0:c0 + k, :nb] = wa
            print("shard %d: %4d / %d channels, %.0f s" % (r, c0 + k, CHANNELS, time.time() - t0), flush=True)
        trows, tcoefs = torch.from_numpy(rows), torch.from_numpy(coefs)
        inner = vdist.row_digests(trows, nb, tcoefs)
        digest = vdist.combine_row_digests(inner, first)
        assert digest == vdist.rows_digest(trows, nb, tcoefs, first)
        sha = hashlib.sha256()
        sha.update(coefs.tobytes())
        for c0 in range(0, CHANNELS, 512):
            sha.update(np.ascontiguousarray(rows[c0:c0 + 512, :nb]).tobytes())
        per_channel[first:first + CHANNELS] = inner.numpy().view(np.uint64)
        state["shards"][str(r)] = {"rank": r, "first_channel": first, "rows_digest": "0x%016x" % digest, "sha256": sha.hexdigest()}
        out = {"what": "GC-ADPCM coefficients + bitstream of every shard of BASELINE configs[4] (vgaudio_amd.synth channels "
                       "first_channel .. first_channel + 4095, 2 880 000 samples each) as the CPU ORACLE produces them "
                       "(oracle/gcadpcm_oracle.c), written by tests/golden/make_gc_shard_oracle_digests.py in the build container, "
                       "no GPU involved; rows_digest = vgaudio_amd.distributed.rows_digest, sha256 over the int16 coefficients "
                       "[4096][16] followed by the rows' data bytes [4096][byte count]; gc_channel_oracle_digests.npy holds "
                       "vgaudio_amd.distributed.row_digests of every channel (uint64 [32768])",
               "provenance": "oracle", "channels_per_shard": CHANNELS, "samples_per_channel": n, "bytes_per_row": nb,
               "shards": [state["shards"][k] for k in sorted(state["shards"], key=int)]}
        json.dump(out, open(OUT_JSON, "w"), indent=1)
        np.save(OUT_NPY, per_channel)
        print("shard", r, "done in %.0f s:" % (time.time() - t0), state["shards"][str(r)], flush=True)


if __name__ == "__main__":
    main()
