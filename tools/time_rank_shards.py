"""Encode time of every rank's shard of BASELINE configs[4] (8 x 4096 channels) on one GPU: the time pieces' seams depend
on the data, and a channel whose seam stays open is re-encoded serially by the repair launch."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev
d = torch.device('cuda:0'); nch, n = 4096, 2880000
pcm = vdev.alloc_pcm(nch, n, d); out = vdev.alloc_adpcm(nch, n, d)
for rank in range(8):
    vdev.synth_pcm(nch, n, d, first_channel=rank * nch, out=pcm)
    coefs = vdev.gc_coefs(pcm, n)
    vdev.gc_encode(pcm, n, coefs, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); vdev.gc_encode(pcm, n, coefs, out=out); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("rank %d shard: encode %.2f ms" % (rank, min(ts)), flush=True)
