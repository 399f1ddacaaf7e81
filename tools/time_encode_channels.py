"""How do the GC-ADPCM kernels' times scale with the number of channels?  (Few channels are cut into time pieces that
run side by side, LABNOTES.md 4.3; BASELINE configs[0] is 1 channel x 480 000 samples, configs[1] 4096 x 2 880 000.)"""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev
d = torch.device('cuda:0')
def t(f):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
SHAPES = tuple((int(a), 2880000) for a in sys.argv[1:]) or ((1, 480000), (1, 2880000), (2, 2880000), (8, 2880000), (64, 2880000), (96, 2880000), (256, 2880000), (384, 2880000),
          (512, 2880000), (1024, 2880000), (2048, 2880000), (4096, 2880000))
for nch, n in SHAPES:
    pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); out = vdev.alloc_adpcm(nch, n, d)
    dec = vdev.alloc_pcm(nch, n, d)
    tc = t(lambda: vdev.gc_coefs(pcm, n)); te = t(lambda: vdev.gc_encode(pcm, n, coefs, out=out))
    td = t(lambda: vdev.gc_decode(out, coefs, n, out=dec))
    print("nch %5d x %8d  coefs %8.2f ms  encode %8.2f ms  decode %7.2f ms   encode Msamples/s %.0f" %
          (nch, n, tc, te, td, nch * n / te / 1e3), flush=True)
    del pcm, out, dec
