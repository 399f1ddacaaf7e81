#!/usr/bin/env python3
"""Ragged GC-ADPCM encode (bench.py's mixed_lengths set: mono files, log-uniform 1-120 s, 11.8 G samples) against the piece
schedule (VGA_HIP_GC_SCHEDULE = "rounds of big items,rounds of short items,frames of a short item"; the plan is made when
the ragged batch is created).  GPU box only.   python tools/time_ragged_schedule.py [--schedules 3,2,4096 ...]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--schedules", nargs="+", default=["3,2,4096", "4,2,4096", "6,2,4096", "8,2,4096", "3,4,4096", "6,4,3072", "12,2,4096", "4,0,4096"])
    ap.add_argument("--with-coefs", action="store_true")
    args = ap.parse_args()
    import torch
    from vgaudio_amd import device as vdev
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0xBA7C4)
    lens, total = [], 0
    while total < 4096 * 2_880_000:
        n = int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0))))
        lens.append(n)
        total += n
    row = {"files": len(lens), "samples": total}
    pcm = coefs = None
    ref = None
    for sch in args.schedules:
        if sch == "default":
            os.environ.pop("VGA_HIP_GC_SCHEDULE", None)
        else:
            os.environ["VGA_HIP_GC_SCHEDULE"] = sch
        rb = vdev.GcRaggedBatch(lens, dev)
        if pcm is None:
            pcm = rb.synth(first_channel=1 << 20)
            coefs = rb.coefs(pcm)
            out = rb.alloc_adpcm()
        rb.encode(pcm, coefs, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            rb.encode(pcm, coefs, out=out)
        e1.record()
        torch.cuda.synchronize()
        row[sch] = round(e0.elapsed_time(e1) / 3, 2)
        if args.with_coefs:                              # as bench.py times it: the coefficient kernel before every encode
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            for k in range(3):
                rb.coefs(pcm)
                ev[2 * k].record()
                rb.encode(pcm, coefs, out=out)
                ev[2 * k + 1].record()
            torch.cuda.synchronize()
            row[sch + " after coefs"] = round(sum(ev[2 * k].elapsed_time(ev[2 * k + 1]) for k in range(3)) / 3, 2)
        h = int(out[:out.numel() // 8 * 8].view(torch.int64).sum().item())
        ref = h if ref is None else ref
        assert h == ref, "the schedule changed the output"
        rb.close()
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
