"""How does gc_encode time scale with the number of channels (waves per SIMD)?"""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev
d = torch.device('cuda:0'); n = 720000
for nch in (2048, 4096, 8192):
    pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); out = vdev.alloc_adpcm(nch, n, d)
    vdev.gc_encode(pcm, n, coefs, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); vdev.gc_encode(pcm, n, coefs, out=out); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print("nch %6d  encode ms %.2f   Msamples/s %.0f" % (nch, min(ts), nch * n / min(ts) / 1e3), flush=True)
    del pcm, out
