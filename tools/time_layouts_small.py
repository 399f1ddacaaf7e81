#!/usr/bin/env python3
"""GC-ADPCM encode with the two wave layouts (vga_testing_gc_encoder_layout_this_thread 4 / 8) for batches of 1 to 1024
channels x 60 s: (ms layout 4, ms layout 8, same bytes).  GPU box only."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); d = torch.device("cuda:0"); n = 2880000
out = {}
for nch in (1, 8, 16, 64, 96, 128, 256, 384, 512, 1024):
    pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n)
    res = {}; outs = {}
    for layout in (4, 8):
        L.vga_testing_gc_encoder_layout_this_thread(layout)
        o = vdev.alloc_adpcm(nch, n, d)
        for _ in range(2): vdev.gc_encode(pcm, n, coefs, out=o)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); vdev.gc_encode(pcm, n, coefs, out=o); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        res[layout] = round(min(ts), 2); outs[layout] = o
    out[nch] = (res[4], res[8], bool(torch.equal(outs[4], outs[8])))
    print(nch, out[nch], flush=True)
L.vga_testing_gc_encoder_layout_this_thread(0)
