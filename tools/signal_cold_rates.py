"""The GC-ADPCM encoder's cold-block rate (wave-frames that took the cold block: third trips of the retry loop,
GcAdpcmEncoder.cs:127-170, the bump loop, inexact sums) per signal class, from a library built with -DVGA_GC_STATS
(vgaudio_amd/libvgaudio_hip_stats.so; the product does not count: five instructions per cold block).  Run by bench.py's
signal_sensitivity block in a process of its own (VGAUDIO_HIP_LIBRARY points at the stats library); prints one JSON object.
    VGAUDIO_HIP_LIBRARY=vgaudio_amd/libvgaudio_hip_stats.so python tools/signal_cold_rates.py [channels] [seconds]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from vgaudio_amd import _lib, device as vdev, signals  # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(round(float(sys.argv[2]) * 48000)) if len(sys.argv) > 2 else 2880000
d = torch.device("cuda:0")
L = _lib.lib()
raw = (C.c_ulonglong * 8)()
pcm = vdev.alloc_pcm(nch, n, d)
adpcm = vdev.alloc_adpcm(nch, n, d)
out = {}
for cls in ("synthetic",) + tuple(signals.CLASSES):
    if cls == "synthetic":
        vdev.synth_pcm(nch, n, d, out=pcm)
    else:
        signals.device(cls, nch, n, d, out=pcm)
    coefs = vdev.gc_coefs(pcm, n)
    L.vga_testing_gc_encode_stats(None, 1)
    vdev.gc_encode(pcm, n, coefs, out=adpcm)
    if L.vga_testing_gc_encode_stats(raw, 1) != 0 or int(raw[7]) != 1 or int(raw[3]) == 0:
        out[cls] = None
    else:
        out[cls] = round(int(raw[4]) / int(raw[3]), 4)
print(json.dumps(out))
