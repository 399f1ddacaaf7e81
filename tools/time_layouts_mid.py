#!/usr/bin/env python3
"""GC-ADPCM encode (the launcher's own layout choice) for mid-size batches x 60 s.  GPU box only."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vgaudio_amd import device as vdev
d = torch.device("cuda:0"); n = 2880000
for nch in (256, 512, 1024, 2048):
    pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); o = vdev.alloc_adpcm(nch, n, d)
    for _ in range(2): vdev.gc_encode(pcm, n, coefs, out=o)
    torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); vdev.gc_encode(pcm, n, coefs, out=o); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(nch, round(min(ts), 2), flush=True)
