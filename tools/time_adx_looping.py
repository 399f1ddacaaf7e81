"""ADX encode / decode of configs[2]'s shape with and without a loop's alignment padding (CriAdxFormat.cs:59-62: LoopStart = 1000,
mono: Padding = 24), device-resident; a sample of channels against the oracle."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import _lib, device as vdev
from oracle import pyoracle as po
L = _lib.lib(); d = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 2880000
pcm = vdev.synth_pcm(nch, n, d)
for padding in (0, 24, 57):
    p = _lib.AdxParams(); L.vga_adx_default_params(C.byref(p)); p.padding = padding
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p)); pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d); hist = torch.zeros(nch, dtype=torch.int16, device=d)
    back = vdev.alloc_pcm(nch, n, d); status = torch.zeros(1, dtype=torch.int32, device=d)
    enc = lambda: _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
    dec = lambda: _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), st))
    def t(f, reps=3):
        f(); torch.cuda.synchronize(); ts = []
        for _ in range(reps):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        return min(ts)
    te = t(enc); td = t(dec, 1 if padding else 3)
    idx = [0, 1, nch // 2, nch - 1]
    host = pcm[idx, :n].cpu().numpy()
    op = po.adx_params(padding=padding)
    want, whist = po.adx_encode_batch(host, op, threads=4)
    ok_e = np.array_equal(adx[idx, :nb].cpu().numpy(), want) and np.array_equal(hist[idx].cpu().numpy(), whist)
    wdec = po.adx_decode_batch(want, n, po.adx_params(padding=padding), threads=4)
    ok_d = np.array_equal(back[idx, :n].cpu().numpy(), wdec)
    print("padding %2d: encode %.2f ms  decode %.2f ms  bit-exact vs oracle (4 channels): encode %s decode %s" % (padding, te, td, ok_e, ok_d), flush=True)
