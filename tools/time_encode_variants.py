"""Times gc_encode on every library under tools/variants/ (plus the product library) at BASELINE
configs[1] and prints a checksum of the output so variants can be compared for bit-exactness.
Each library is loaded in its own process (VGAUDIO_HIP_LIBRARY)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from vgaudio_amd import device as vdev
d = torch.device('cuda:0'); nch, n = 4096, 2880000
pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); out = vdev.alloc_adpcm(nch, n, d)
tc = []
for _ in range(2):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); coefs = vdev.gc_coefs(pcm, n); b.record(); torch.cuda.synchronize()
    tc.append(a.elapsed_time(b))
vdev.gc_encode(pcm, n, coefs, out=out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); vdev.gc_encode(pcm, n, coefs, out=out); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
h = int(out.view(torch.int64).sum().item()) if out.numel() %% 8 == 0 else int(out.to(torch.int64).sum().item())
hc = int(coefs.to(torch.int64).mul(torch.arange(1, coefs.numel() + 1, device=d).view_as(coefs) %% 1000003).sum().item())
print("encode ms %%.2f  checksum %%d | coefs ms %%.2f  checksum %%d" %% (min(ts), h, min(tc), hc))
''' % ROOT
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "libvga_*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["VGAUDIO_HIP_LIBRARY"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-40s %s" % (os.path.basename(lib) if lib else "product", (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
