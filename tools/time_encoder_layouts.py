#!/usr/bin/env python3
"""A/B of the two GC-ADPCM encoder wave layouts (vga_testing_gc_encoder_layout_this_thread: 4 = lane per (channel,
predictor, candidate), 8 = lane per (channel, predictor)) at BASELINE configs[1] and a few smaller batches: HIP-event
time of vga_gcadpcm_encode_device and a byte comparison of the outputs.  GPU box only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev  # noqa: E402

L = _lib.lib()
d = torch.device("cuda:0")
n = 2880000
out = {}
for nch in (4096, 1024, 64, 1):
    pcm = vdev.synth_pcm(nch, n, d)
    coefs = vdev.gc_coefs(pcm, n)
    res = {}
    outs = {}
    for layout in (4, 8):
        L.vga_testing_gc_encoder_layout_this_thread(layout)
        o = vdev.alloc_adpcm(nch, n, d)
        for _ in range(2):
            vdev.gc_encode(pcm, n, coefs, out=o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            vdev.gc_encode(pcm, n, coefs, out=o)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[layout] = round(min(ts), 3)
        outs[layout] = o
    out[nch] = {"ms_layout4": res[4], "ms_layout8": res[8], "identical": bool(torch.equal(outs[4], outs[8]))}
    del pcm, outs
L.vga_testing_gc_encoder_layout_this_thread(0)
print(json.dumps(out))
# time pieces per channel at configs[1], layout 8
nch = 4096
pcm = vdev.synth_pcm(nch, n, d)
coefs = vdev.gc_coefs(pcm, n)
o = vdev.alloc_adpcm(nch, n, d)
seg = {}
for segments in (0, 1, 2, 3, 4, 6, 8):
    L.vga_testing_gc_encoder_segments_this_thread(segments)
    for _ in range(2):
        vdev.gc_encode(pcm, n, coefs, out=o)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        vdev.gc_encode(pcm, n, coefs, out=o)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    seg[segments] = round(min(ts), 2)
L.vga_testing_gc_encoder_segments_this_thread(0)
print(json.dumps({"segments_ms": seg}))
