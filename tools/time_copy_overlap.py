#!/usr/bin/env python3
"""Do host<->device copies overlap with a chip-filling kernel on this box?  Times a pinned 8 GiB H2D copy, the GC-ADPCM
encode of configs[1], and both at once on two streams.  Run with HSA_ENABLE_SDMA unset / 0 / 1.  GPU box only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import device as vdev  # noqa: E402

d = torch.device("cuda:0")
nch, n = 4096, 2880000
pcm = vdev.synth_pcm(nch, n, d)
coefs = vdev.gc_coefs(pcm, n)
out = vdev.alloc_adpcm(nch, n, d)
vdev.gc_encode(pcm, n, coefs, out=out)
pin = torch.empty(8 << 30, dtype=torch.uint8).pin_memory()
dst = torch.empty(8 << 30, dtype=torch.uint8, device=d)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()


def t(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def copy():
    with torch.cuda.stream(s1):
        dst.copy_(pin, non_blocking=True)


def back():
    with torch.cuda.stream(s1):
        pin.copy_(dst, non_blocking=True)


def kern():
    with torch.cuda.stream(s2):
        vdev.gc_encode(pcm, n, coefs, out=out)
        vdev.gc_encode(pcm, n, coefs, out=out)


res = {"env": {k: v for k, v in os.environ.items() if k.startswith(("HSA_", "HIP_", "GPU_", "ROC"))},
       "h2d_8GiB_ms": round(min(t(copy) for _ in range(2)), 1), "d2h_8GiB_ms": round(min(t(back) for _ in range(2)), 1),
       "two_encodes_ms": round(min(t(kern) for _ in range(2)), 1)}
res["h2d_and_kernels_ms"] = round(min(t(lambda: (copy(), kern())) for _ in range(2)), 1)
res["kernels_then_h2d_issue_order_ms"] = round(min(t(lambda: (kern(), copy())) for _ in range(2)), 1)
print(json.dumps(res))
