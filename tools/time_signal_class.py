"""One signal class (vgaudio_amd/signals.py) through GC-ADPCM decode and ADX encode / decode at BASELINE configs[1]'s shape, for
rocprofv3 --kernel-trace --stats: which kernel of the launch is the slow one on signals whose seams behave differently.
    python tools/time_signal_class.py clipped_square [channels]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev, signals  # noqa: E402

cls = sys.argv[1]
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = 2880000
d = torch.device("cuda:0")
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
pcm = signals.device(cls, nch, n, d)
coefs = vdev.gc_coefs(pcm, n)
adpcm = vdev.gc_encode(pcm, n, coefs)
back = vdev.alloc_pcm(nch, n, d)
p = _lib.AdxParams()
L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
pitch = (nb + 15) // 16 * 16
adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d)
hist = torch.zeros(nch, dtype=torch.int16, device=d)
status = torch.zeros(1, dtype=torch.int32, device=d)


def timed(name, fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    print("%-16s %-12s %9.2f ms" % (cls, name, a.elapsed_time(b)), flush=True)


timed("gc decode", lambda: vdev.gc_decode(adpcm, coefs, n, out=back))
timed("adx encode", lambda: _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st)))
timed("adx decode", lambda: _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), st)))
