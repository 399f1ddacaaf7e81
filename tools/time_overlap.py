"""Does the coefficient kernel of step k+1 fit into the issue slots the encoder of step k leaves idle?
Sequential (one stream) vs software-pipelined (two streams, double-buffered coefficients/records)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import device as vdev, _lib
d = torch.device('cuda:0'); nch, n = 4096, 2880000
pcm = vdev.synth_pcm(nch, n, d); out = vdev.alloc_adpcm(nch, n, d)
L = _lib.lib()
wsb = L.vga_gcadpcm_coefs_workspace_bytes(nch, n)
ws = [torch.empty(wsb, dtype=torch.uint8, device=d) for _ in range(2)]
def seq(K):
    for _ in range(K):
        c = vdev.gc_coefs(pcm, n, workspace=ws[0]); vdev.gc_encode(pcm, n, c, out=out)
def pipe(K, sa, sb):
    evs = []
    for k in range(K):
        with torch.cuda.stream(sa):
            c = vdev.gc_coefs(pcm, n, workspace=ws[k & 1])
            e = torch.cuda.Event(); e.record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(e)
            vdev.gc_encode(pcm, n, c, out=out)
            c.record_stream(sb)
            if k >= 1:
                pass
        evs.append(c)
    return evs
seq(3); torch.cuda.synchronize()
for K in (5,):
    t0 = time.perf_counter(); seq(K); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("sequential  %.1f ms/step" % ((t1 - t0) / K * 1e3))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    pipe(2, sa, sb); torch.cuda.synchronize()
    t0 = time.perf_counter(); keep = pipe(K, sa, sb); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("pipelined   %.1f ms/step" % ((t1 - t0) / K * 1e3))
