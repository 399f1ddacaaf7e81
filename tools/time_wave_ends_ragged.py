#!/usr/bin/env python3
"""When do the waves of gc_coefs_kernel start and end in the RAGGED launch of bench.py's mixed-lengths set (10 008 files of
1-120 s, longest first)?  Needs a -DVGA_DEBUG_TIMESTAMPS build of gcadpcm_kernels.hip:
    VARIED=gcadpcm_kernels tools/build_variants.sh ts:"-DVGA_DEBUG_TIMESTAMPS"
    VGAUDIO_HIP_LIBRARY=tools/variants/libvga_ts.so python tools/time_wave_ends_ragged.py
Work slot i (workgroup i) holds the i-th longest file: prints start / end by slot decile and the launch's last enders."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=0, help="only the N longest files of the set")
    a = ap.parse_args()
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
    if a.top:
        lens = sorted(lens, reverse=True)[:a.top]
    rb = vdev.GcRaggedBatch(lens, dev)
    pcm = rb.synth(first_channel=1 << 20)
    ws = torch.empty(max(rb.workspace_bytes, 16), dtype=torch.uint8, device=dev)
    for _ in range(2):
        coefs = rb.coefs(pcm, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    coefs = rb.coefs(pcm, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    n = len(lens)
    ts = np.zeros(3 * 16384, dtype=np.uint64)
    raw.vga_debug_coefs_timestamps(ts.ctypes.data_as(C.c_void_p), ts.size)
    t = ts[:3 * n].reshape(-1, 3).astype(np.float64) / 1e5   # 100 MHz ticks -> ms; row = work slot = rank by length
    t0 = t[:, 0].min()
    start, end = t[:, 0] - t0, t[:, 2] - t0
    order = np.argsort(-np.asarray(lens), kind="stable")
    secs = np.asarray(lens)[order] / 48000.0
    print(json.dumps({"files": n, "launch_ms": round(e0.elapsed_time(e1), 2), "last_end_ms": round(float(end.max()), 2)}))
    for lo in range(0, n, n // 10 + 1):
        hi = min(n, lo + n // 10 + 1)
        print(json.dumps({"slots": [lo, hi], "seconds": [round(float(secs[hi - 1]), 1), round(float(secs[lo]), 1)],
                          "start_ms": [round(float(start[lo:hi].min()), 2), round(float(start[lo:hi].max()), 2)],
                          "end_ms": [round(float(end[lo:hi].min()), 2), round(float(np.median(end[lo:hi])), 2), round(float(end[lo:hi].max()), 2)],
                          "life_ms_median": round(float(np.median(end[lo:hi] - start[lo:hi])), 2)}))
    last = np.argsort(end)[-12:]
    print(json.dumps({"last_enders_slot": [int(i) for i in last], "seconds": [round(float(secs[i]), 1) for i in last],
                      "start_ms": [round(float(start[i]), 2) for i in last], "end_ms": [round(float(end[i]), 2) for i in last]}))
    # how busy is the chip: waves alive over time
    grid = np.linspace(0, end.max(), 41)
    alive = [(int(((start <= g) & (end > g)).sum())) for g in grid]
    print(json.dumps({"t_ms": [round(float(g), 1) for g in grid], "waves_alive": alive}))


if __name__ == "__main__":
    main()
