#!/usr/bin/env python3
"""ADX encode / decode (device-resident, 18-byte frames, 60 s at 48 kHz) against the number of channels.  GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev  # noqa: E402

L = _lib.lib()
d = torch.device("cuda:0")
n = 2880000
p = _lib.AdxParams()
L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
pitch = (nb + 15) // 16 * 16
st = torch.cuda.current_stream().cuda_stream
for nch in (1, 8, 64, 256, 512, 1024, 2048, 4096):
    pcm = vdev.synth_pcm(nch, n, d)
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
    hist = torch.zeros(nch, dtype=torch.int16, device=d)
    dec = vdev.alloc_pcm(nch, n, d)
    status = torch.zeros(1, dtype=torch.int32, device=d)

    def enc():
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))

    def dco():
        _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), dec.data_ptr(), dec.stride(0), status.data_ptr(), st))

    out = []
    for f in (enc, dco):
        f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            f()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / 3)
    print("nch %5d x %d  encode %8.2f ms  decode %7.2f ms   encode Msamples/s %.0f" % (nch, n, out[0], out[1], nch * n / out[0] / 1e3))
    del pcm, adx, dec
    torch.cuda.empty_cache()
