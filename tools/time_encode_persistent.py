#!/usr/bin/env python3
"""GC-ADPCM encode: one workgroup per (channel group, piece) against persistent workgroups taking the items from a queue
(vga_testing_gc_encoder_persistent_this_thread), over piece counts.  GPU box only.
    python tools/time_encode_persistent.py [--channels 4096 1024] [--pieces 0 16 32 64]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[4096, 2048, 1024, 512])
    ap.add_argument("--pieces", type=int, nargs="+", default=[0, 8, 16, 32, 64])
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
        ref = None
        for mode, name in ((1, "grid"), (2, "persistent"), (0, "launcher")):
            row = {"channels": nch, "mode": name}
            L.vga_testing_gc_encoder_persistent_this_thread(mode)
            for pieces in (args.pieces if mode else [0]):
                L.vga_testing_gc_encoder_segments_this_thread(pieces)
                vdev.gc_encode(pcm, n, coefs, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    vdev.gc_encode(pcm, n, coefs, out=out)
                e1.record()
                torch.cuda.synchronize()
                row["auto" if pieces == 0 else str(pieces)] = round(e0.elapsed_time(e1) / 3, 2)
                h = int(out.view(torch.int64).sum().item())
                ref = h if ref is None else ref
                assert h == ref, "the mode / piece count changed the output"
            print(json.dumps(row), flush=True)
        L.vga_testing_gc_encoder_segments_this_thread(0)
        L.vga_testing_gc_encoder_persistent_this_thread(0)
        del pcm, out


if __name__ == "__main__":
    main()
