#!/usr/bin/env python3
"""Is the ragged GC-ADPCM encoder slower per frame, or is it the mix of lengths?  4096 channels x 60 s through the ragged entry
point with ONE channel a frame shorter (so that the ragged kernels run) against the same batch through the equal-length
kernels.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev  # noqa: E402

dev = torch.device("cuda:0")
nch, n = 4096, 2_880_000


def timed(f, reps=3):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return round(min(ts), 2)


pcm = vdev.synth_pcm(nch, n, dev)
coefs = vdev.gc_coefs(pcm, n)
out = vdev.alloc_adpcm(nch, n, dev)
print("equal-length kernels: encode", timed(lambda: vdev.gc_encode(pcm, n, coefs, out=out)), "ms", flush=True)
del pcm, out
for short in (14, 14 * 4096, 14 * 100000):
    lens = [n] * nch
    lens[nch // 2] = n - short
    rb = vdev.GcRaggedBatch(lens, dev)
    p = rb.synth(first_channel=0)
    c = rb.coefs(p)
    o = rb.alloc_adpcm()
    print(f"ragged kernels, one channel {short // 14} frames shorter: encode", timed(lambda: rb.encode(p, c, out=o)), "ms; coefs",
          timed(lambda: rb.coefs(p)), "ms", flush=True)
    rb.close()
    del p, o
