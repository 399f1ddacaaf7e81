#!/usr/bin/env python3
"""ADX encode at BASELINE configs[2] with every seam forced open: adx_encode_fs18_fixup_kernel then runs each piece
(11 250 frames) to its end, so its duration / frames is the seam run's time per frame (run under tools/prof_kernels.sh).
GPU box only."""
import ctypes as C
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
adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
ref = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), ref.data_ptr(), pitch, hist.data_ptr(), st))
L.vga_testing_force_open_seams_this_thread(1)
_lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
L.vga_testing_force_open_seams_this_thread(0)
torch.cuda.synchronize()
print("same bytes with every seam forced open:", bool(torch.equal(adx, ref)))
