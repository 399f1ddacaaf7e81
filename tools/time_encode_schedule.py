#!/usr/bin/env python3
"""GC-ADPCM encode with persistent workgroups: launch time against the piece schedule (VGA_HIP_GC_SCHEDULE =
"rounds of big items,rounds of short items,frames of a short item").  GPU box only.
    python tools/time_encode_schedule.py [--channels 4096] [--schedules 3,4 4,3 ...]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[4096])
    ap.add_argument("--schedules", nargs="+", default=["2,2,4096", "3,2,4096", "2,3,4096", "2,2,6144", "3,3,3072", "2,4,3072", "1,4,4096", "4,2,4096", "2,1,4096", "2,2,3072", "3,2,6144"])
    ap.add_argument("--seconds", type=float, default=60.0)
    args = ap.parse_args()
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n = int(args.seconds * 48000)
    for nch in args.channels:
        pcm = vdev.synth_pcm(nch, n, dev)
        coefs = vdev.gc_coefs(pcm, n)
        out = vdev.alloc_adpcm(nch, n, dev)
        row = {"channels": nch}
        ref = None
        L.vga_testing_gc_encoder_persistent_this_thread(2)
        for sch in args.schedules:
            os.environ["VGA_HIP_GC_SCHEDULE"] = sch
            vdev.gc_encode(pcm, n, coefs, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                vdev.gc_encode(pcm, n, coefs, out=out)
            e1.record()
            torch.cuda.synchronize()
            row[sch] = round(e0.elapsed_time(e1) / 3, 2)
            h = int(out.view(torch.int64).sum().item())
            ref = h if ref is None else ref
            assert h == ref, "the schedule changed the output"
        L.vga_testing_gc_encoder_persistent_this_thread(0)
        print(json.dumps(row), flush=True)
        del pcm, out


if __name__ == "__main__":
    main()
