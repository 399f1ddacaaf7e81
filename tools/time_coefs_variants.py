#!/usr/bin/env python3
"""Times the two coefficient-search kernels (1: one wave per channel; 2: workgroups of four channels + a summing wave; 0: the launcher's choice) at
a given shape and checks that they agree.  GPU box only.
    python tools/time_coefs_variants.py [--channels 4096] [--seconds 60]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, nargs="+", default=[4096, 3072, 2048, 1536, 1024, 128, 8, 1])
    ap.add_argument("--seconds", type=float, default=60.0)
    args = ap.parse_args()
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    n = int(args.seconds * 48000)
    for nch in args.channels:
        pcm = vdev.synth_pcm(nch, n, dev)
        ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=dev)
        out = {}
        res = {}
        for variant in (1, 2, 3, 0):
            L.vga_testing_gc_coefs_variant_this_thread(variant)
            for _ in range(2):
                c = vdev.gc_coefs(pcm, n, workspace=ws)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                c = vdev.gc_coefs(pcm, n, workspace=ws)
            e1.record()
            torch.cuda.synchronize()
            out[variant] = round(e0.elapsed_time(e1) / 3, 3)
            res[variant] = c.clone()
        L.vga_testing_gc_coefs_variant_this_thread(0)
        print(json.dumps({"channels": nch, "samples": n, "one_wave_per_channel_ms": out[1],
                          "four_channels_and_summing_wave_ms": out[2], "five_waves_on_one_channel_ms": out[3],
                          "launcher_choice_ms": out[0],
                          "same_coefficients": bool(torch.equal(res[0], res[1]) and torch.equal(res[0], res[2]) and
                                                    torch.equal(res[0], res[3]))}), flush=True)
        del pcm, ws


if __name__ == "__main__":
    main()
