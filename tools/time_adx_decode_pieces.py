"""ADX decode at configs[2] (4096 channels x 60 s) for several numbers of time pieces per channel (0 = the launcher's choice),
through vga_testing_gc_encoder_segments_this_thread.  Honours VGAUDIO_HIP_LIBRARY (tools/variants/)."""
import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); d = torch.device("cuda:0"); n = 2880000; nch = 4096
st = torch.cuda.current_stream().cuda_stream
pcm = vdev.synth_pcm(nch, n, d)
back = vdev.alloc_pcm(nch, n, d)
p = _lib.AdxParams(); L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p)); pitch = (nb + 15) // 16 * 16
adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d); hist = torch.zeros(nch, dtype=torch.int16, device=d)
status = torch.zeros(1, dtype=torch.int32, device=d)
_lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
def t(f):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for seg in (0, 4, 8, 12, 16, 24, 32, 64):
    L.vga_testing_gc_encoder_segments_this_thread(seg)
    ms = t(lambda: _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), st)))
    print("segments", seg, "adx_decode %.2f ms" % ms, int(back.view(torch.int64).sum().item()))
L.vga_testing_gc_encoder_segments_this_thread(0)
