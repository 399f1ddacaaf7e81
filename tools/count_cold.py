"""How often does an encoder wave take the cold block (third trips / rare paths)?  Needs the counting build:
    tools/build_variants.sh "count:-DVGA_X_COUNT"   (then run this on the GPU box)"""
import sys, ctypes as C, os, torch
os.environ["VGAUDIO_HIP_LIBRARY"] = os.path.join(os.getcwd(), "tools/variants/libvga_count.so")
sys.path.insert(0, os.getcwd())
from vgaudio_amd import device as vdev, _lib
d = torch.device('cuda:0'); nch, n = 4096, 720000
pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); out = vdev.alloc_adpcm(nch, n, d)
vdev.gc_encode(pcm, n, coefs, out=out); torch.cuda.synchronize()
L = _lib.lib(); buf = (C.c_ulonglong * 8)()
L.vga_debug_counters.argtypes = [C.c_void_p]
print("rc", L.vga_debug_counters(buf)); v = list(buf)
print("wave-frames", v[0], "cold", v[1], "rare", v[2], "resume", v[3], "wide", v[4], "resume lanes", v[5],
      "cold frac %.3f" % (v[1] / max(v[0], 1)), "resume per pair %.4f" % (v[5] / max(v[0], 1) / 32))
