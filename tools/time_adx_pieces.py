#!/usr/bin/env python3
"""ADX encode at BASELINE configs[2] for several time-piece counts (vga_testing_gc_encoder_segments_this_thread also
steers the ADX encoder's pieces).  GPU box only."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev  # noqa: E402

L = _lib.lib()
d = torch.device("cuda:0")
nch, n = 4096, 2880000
pcm = vdev.synth_pcm(nch, n, d)
p = _lib.AdxParams()
L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
pitch = (nb + 15) // 16 * 16
hist = torch.zeros(nch, dtype=torch.int16, device=d)
st = lambda: torch.cuda.current_stream().cuda_stream
res, ref = {}, None
for segments in (0, 4, 8, 12, 16, 24, 32, 48, 64):
    L.vga_testing_gc_encoder_segments_this_thread(segments)
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    for _ in range(2):
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st()))
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st()))
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ref = adx if ref is None else ref
    res[segments] = {"ms": round(min(ts), 2), "same": bool(torch.equal(adx, ref))}
L.vga_testing_gc_encoder_segments_this_thread(0)
print(json.dumps(res))
