#!/usr/bin/env python3
"""Three ADX encodes at BASELINE configs[2] (4096 x 60 s) and nothing else: the process to put under rocprofv3 when only
the encoder's kernels are wanted (tools/pmc_adx_encode.sh).  GPU box only."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev  # noqa: E402

L = _lib.lib()
d = torch.device("cuda:0")
nch, n = int(os.environ.get("NCH", "4096")), 2880000
pcm = vdev.synth_pcm(nch, n, d)
p = _lib.AdxParams()
L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
pitch = (nb + 15) // 16 * 16
hist = torch.zeros(nch, dtype=torch.int16, device=d)
adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
st = torch.cuda.current_stream().cuda_stream
L.vga_testing_gc_encoder_segments_this_thread(int(os.environ.get("VGA_ADX_SEGMENTS", "0")))   # 0 = the library's choice
for _ in range(int(os.environ.get("CALLS", "3"))):
    _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
torch.cuda.synchronize()
