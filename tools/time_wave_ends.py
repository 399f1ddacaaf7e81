#!/usr/bin/env python3
"""When do the waves of gc_coefs_kernel / gc_encode_kernel end?  Needs a -DVGA_DEBUG_TIMESTAMPS build:
    tools/build_variants.sh ts:"-DVGA_DEBUG_TIMESTAMPS"      (VARIED="gc_encode_kernel gcadpcm_kernels")
    VGAUDIO_HIP_LIBRARY=tools/variants/libvga_ts.so python tools/time_wave_ends.py [--channels 4096] [--seconds 60]
Prints the distribution of start / end times (ms after the first start) of every workgroup: a kernel whose waves end
long before its last wave leaves issue slots empty at the end."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dist(x):
    q = np.percentile(x, [0, 5, 25, 50, 75, 95, 100])
    return {"min": round(float(q[0]), 2), "p5": round(float(q[1]), 2), "p25": round(float(q[2]), 2), "median": round(float(q[3]), 2),
            "p75": round(float(q[4]), 2), "p95": round(float(q[5]), 2), "max": round(float(q[6]), 2), "mean": round(float(np.mean(x)), 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--persistent", type=int, default=0, help="vga_testing_gc_encoder_persistent_this_thread: 0 launcher, 1 grid, 2 persistent")
    ap.add_argument("--pieces", type=int, default=0)
    ap.add_argument("--signal", default="", help="a class of vgaudio_amd/signals.py instead of the synthetic generator (e.g. slow_channel_93: every row the same)")
    a = ap.parse_args()
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    raw = C.CDLL(_lib.SO_PATH)
    dev = torch.device("cuda:0")
    n = int(a.seconds * 48000)
    nch = a.channels
    if a.signal:
        from vgaudio_amd import signals
        pcm = signals.device(a.signal, nch, n, dev)
    else:
        pcm = vdev.synth_pcm(nch, n, dev)
    for _ in range(2):
        coefs = vdev.gc_coefs(pcm, n)
    torch.cuda.synchronize()
    ts = np.zeros(3 * 8192, dtype=np.uint64)
    raw.vga_debug_coefs_timestamps(ts.ctypes.data_as(C.c_void_p), ts.size)
    t = ts[:3 * min(nch, 8192)].reshape(-1, 3).astype(np.float64) / 1e5        # 100 MHz ticks -> ms
    t0 = t[:, 0].min()
    print(json.dumps({"kernel": "gc_coefs_kernel", "waves": int(t.shape[0]), "start_ms": dist(t[:, 0] - t0), "pass0_end_ms": dist(t[:, 1] - t0),
                      "end_ms": dist(t[:, 2] - t0), "life_ms": dist(t[:, 2] - t[:, 0])}))
    # where the spread lives: workgroup i (one wave) goes to XCD i % 8, CU (i / 8) % 32 of it; the sixteen waves of a CU are
    # i, i + 256, ...; end times by XCD, by CU (its last wave), and by launch order (i / 256: the order in which a CU received them)
    if t.shape[0] == 4096:
        e = t[:, 2] - t0
        by_cu = e.reshape(16, 256)
        print(json.dumps({"end_ms_by_xcd_mean": [round(float(e[x::8].mean()), 2) for x in range(8)],
                          "end_ms_by_xcd_max": [round(float(e[x::8].max()), 2) for x in range(8)],
                          "last_wave_of_a_cu_ms": dist(by_cu.max(axis=0)), "first_wave_of_a_cu_ms": dist(by_cu.min(axis=0)),
                          "mean_wave_of_a_cu_ms": dist(by_cu.mean(axis=0)),
                          "end_ms_by_arrival_order_mean": [round(float(by_cu[k].mean()), 2) for k in range(16)]}))
    if not hasattr(raw, "vga_debug_encode_timestamps"):     # a build of gcadpcm_kernels.hip only
        return
    out = vdev.alloc_adpcm(nch, n, dev)
    L.vga_testing_gc_encoder_persistent_this_thread(a.persistent)
    L.vga_testing_gc_encoder_segments_this_thread(a.pieces)
    for _ in range(2):
        vdev.gc_encode(pcm, n, coefs, out=out)
    torch.cuda.synchronize()
    ts = np.zeros(1 << 16, dtype=np.uint64)
    raw.vga_debug_encode_timestamps(ts.ctypes.data_as(C.c_void_p), ts.size)
    t = ts.reshape(-1, 2).astype(np.float64) / 1e5
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    print(json.dumps({"kernel": "gc_encode_kernel", "workgroups": int(t.shape[0]), "start_ms": dist(t[:, 0] - t0), "end_ms": dist(t[:, 1] - t0),
                      "life_ms": dist(t[:, 1] - t[:, 0])}))
    # per channel group x (the four pieces of a group share a CU: workgroup i goes to XCD i % 8, CU (i / 8) % 32): when its
    # last piece ended, and the sum of its pieces' lifetimes -- what the CU would need if it were never idle
    groups = (nch + 15) // 16
    if a.persistent != 2 and t.shape[0] % groups == 0:
        ee = (t[:, 1] - t0).reshape(-1, groups)
        print(json.dumps({"pieces": int(ee.shape[0]), "last_piece_of_a_group_ends_ms": dist(ee.max(axis=0)),
                          "first_piece_of_a_group_ends_ms": dist(ee.min(axis=0)), "mean_over_pieces_ms": dist(ee.mean(axis=0))}))
        np.save(os.path.join(ROOT, "gpurun_out", "r04", "encode_wg_end_ms.npy"), ee)
    # per channel group (blockIdx.x): mean end time over its pieces, the eight slowest and fastest groups
    e = t[:, 1] - t0
    order = np.argsort(e)
    print(json.dumps({"slowest_workgroups": [int(i) for i in order[-8:]], "their_end_ms": [round(float(e[i]), 1) for i in order[-8:]],
                      "fastest_workgroups": [int(i) for i in order[:8]], "their_end_ms_": [round(float(e[i]), 1) for i in order[:8]]}))


if __name__ == "__main__":
    main()
