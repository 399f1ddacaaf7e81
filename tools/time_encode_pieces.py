#!/usr/bin/env python3
"""GC-ADPCM encode time against the number of time pieces, for channel counts that do not fill the chip (the host
pipeline's chunks).  GPU box only.   python tools/time_encode_pieces.py [--channels 384 512 640 1024] [--pieces 0 8 16 32 64]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[256, 384, 512, 640, 1024])
    ap.add_argument("--pieces", type=int, nargs="+", default=[0, 4, 8, 16, 32, 64])
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
        for pieces in args.pieces:
            L.vga_testing_gc_encoder_segments_this_thread(pieces)
            vdev.gc_encode(pcm, n, coefs, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                vdev.gc_encode(pcm, n, coefs, out=out)
            e1.record()
            torch.cuda.synchronize()
            row["auto" if pieces == 0 else str(pieces)] = round(e0.elapsed_time(e1) / 3, 2)
            h = int(out.to(torch.int64).sum().item())
            ref = h if ref is None else ref
            assert h == ref, "the piece count changed the output"
        L.vga_testing_gc_encoder_segments_this_thread(0)
        print(json.dumps(row), flush=True)
        del pcm, out


if __name__ == "__main__":
    main()
