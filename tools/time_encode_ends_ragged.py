#!/usr/bin/env python3
"""When do the persistent workgroups of the ragged GC-ADPCM encoder end on bench.py's mixed-lengths set (10 008 files of 1-120 s)?
Needs a -DVGA_DEBUG_TIMESTAMPS build of gc_encode_kernel.hip (VARIED=gc_encode_kernel tools/build_variants.sh ts:"-DVGA_DEBUG_TIMESTAMPS")."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    raw = C.CDLL(_lib.SO_PATH)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0xBA7C4)                     # bench.py measure_mixed_lengths
    target = 4096 * 2_880_000
    lens, total = [], 0
    while total < target:
        n = int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0))))
        lens.append(n)
        total += n
    rb = vdev.GcRaggedBatch(lens, dev)
    pcm = rb.synth(first_channel=1 << 20)
    coefs = rb.coefs(pcm)
    out = rb.alloc_adpcm()
    for _ in range(2):
        rb.encode(pcm, coefs, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rb.encode(pcm, coefs, out=out)
    e1.record()
    torch.cuda.synchronize()
    ts = np.zeros(1 << 16, dtype=np.uint64)
    raw.vga_debug_encode_timestamps(ts.ctypes.data_as(C.c_void_p), ts.size)
    t = ts.reshape(-1, 2).astype(np.float64) / 1e5
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    end = t[:, 1] - t0
    q = np.percentile(end, [0, 1, 5, 25, 50, 75, 95, 99, 100])
    print(json.dumps({"launch_ms": round(e0.elapsed_time(e1), 2), "workgroups": int(t.shape[0]),
                      "end_ms_percentiles_0_1_5_25_50_75_95_99_100": [round(float(x), 1) for x in q], "mean_end_ms": round(float(end.mean()), 1)}))
    stats = (C.c_ulonglong * 8)()
    raw.vga_testing_gc_encode_stats(stats, 1)
    rb.encode(pcm, coefs, out=out)
    raw.vga_testing_gc_encode_stats(stats, 0)
    print(json.dumps({"seams_closed_in_piece": stats[0], "seams_left_open": stats[1], "frames_reencoded_by_seams": stats[2], "wave_frames": stats[3],
                      "channels_walked_by_chain": stats[5], "pieces": stats[6], "total_frames": int(sum((n + 13) // 14 for n in lens))}))


if __name__ == "__main__":
    main()
