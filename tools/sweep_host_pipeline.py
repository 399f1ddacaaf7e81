#!/usr/bin/env python3
"""Sweeps the shape of the host pipeline (feeders, drainers, chunk size, slot bytes) for vga_gcadpcm_encode_batch at a
given channel count and prints wall time + the breakdown per shape.  GPU box only.
    python tools/sweep_host_pipeline.py [--channels 4096] [--seconds 60]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--shapes", default="8,4,0,0;4,2,0,33554432;3,2,0,33554432;2,2,0,67108864;6,3,0,16777216;4,2,512,33554432;4,2,2048,33554432")
    args = ap.parse_args()
    import numpy as np
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    nch, n = args.channels, int(args.seconds * 48000)
    nb = vdev.gc_byte_count(n)
    pcm = vdev.synth_pcm(nch, n, dev)
    host = np.empty((nch, n), dtype=np.int16)
    for c0 in range(0, nch, 256):
        host[c0:c0 + 256] = pcm[c0:c0 + 256, :n].cpu().numpy()
    del pcm
    torch.cuda.empty_cache()
    outs = np.zeros((nch, nb), dtype=np.uint8)
    cf = np.zeros(nch * 16, dtype=np.int16)
    pp = (_lib.i16p * nch)(*[host[c].ctypes.data_as(_lib.i16p) for c in range(nch)])
    op = (_lib.u8p * nch)(*[outs[c].ctypes.data_as(_lib.u8p) for c in range(nch)])
    names = ["total", "setup", "feeders_memcpy_sum", "feeders_wait_slot_sum", "feeders_issue_sum", "slowest_feeder", "caller_wait_upload",
             "caller_launch", "caller_tail_sync", "drainers_wait_compute_sum", "drainers_wait_download_sum", "drainers_memcpy_sum",
             "slowest_drainer", "feeders", "drainers", "chunks", "units_per_chunk", "device_alloc", "entry_point",
             "feeders_chunk_boundary_sum", "feeders_final_sync_sum", "drainers_register_sum"]
    ref = None
    for shape in args.shapes.split(";"):
        f, d, ch, sl, *rest = (int(v) for v in shape.split(","))
        L.vga_testing_host_pipeline_this_thread(f, d, ch, sl)
        L.vga_testing_host_pipeline_tail_this_thread(rest[0] if rest else 0)        # optional 5th field: the last chunk's size
        best, bd = None, None
        for _ in range(3):
            t0 = time.perf_counter()
            _lib.check(L.vga_gcadpcm_encode_batch(pp, nch, n, 0, 0, cf.ctypes.data_as(_lib.i16p), op))
            dt = time.perf_counter() - t0
            if best is None or dt < best:
                st = (C.c_double * 32)()
                k = L.vga_testing_last_pipeline_stats(st, 32)
                best, bd = dt, {names[i]: (int(st[i]) if 13 <= i <= 16 else round(st[i] * 1e3, 1)) for i in range(k)}
        import zlib
        h = zlib.crc32(outs.tobytes()[:1 << 28])
        ref = h if ref is None else ref
        print(json.dumps({"shape": shape, "wall_ms": round(best * 1e3, 1), "Msamples/s": round(nch * n / best / 1e6, 1),
                          "same_output": h == ref, "breakdown_ms": bd}), flush=True)
    L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)


if __name__ == "__main__":
    main()
