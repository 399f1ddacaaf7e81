"""Why does the GC-ADPCM encoder need 18 ms for 96 channels and 6.8 ms for 64 (profiles/r04_d_channel_scaling.log)?  The
synthetic generator's channel c has tone c % 96 of its table, and tone 93 (11.9 kHz) is the one whose seams between time pieces
take ~1750 frames to close (LABNOTES 8.4): the first 64 channels do not hold it, any 96 consecutive ones do.  Times the encoder
for windows of the generator's channels with and without it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev

d = torch.device("cuda:0")
n = 2880000


def t(f):
    f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts)


for nch, first, note in ((64, 0, "channels 0-63: without tone 93"), (64, 32, "channels 32-95: with it"), (64, 96, "channels 96-159: without"),
                         (96, 0, "channels 0-95: with"), (92, 0, "channels 0-91: without"), (92, 94, "channels 94-185: without"),
                         (256, 0, "channels 0-255: three of them"), (256, 94, "channels 94-349: two of them")):
    pcm = vdev.synth_pcm(nch, n, d, first_channel=first)
    coefs = vdev.gc_coefs(pcm, n)
    out = vdev.alloc_adpcm(nch, n, d)
    te = t(lambda: vdev.gc_encode(pcm, n, coefs, out=out))
    print("nch %4d  first channel %4d  encode %7.2f ms   (%s)" % (nch, first, te, note), flush=True)
    del pcm, out
