"""Times GC-ADPCM decode and the ADX encode / decode at configs[1] / configs[2] shapes (4096 channels x 60 s) on the product
library and every library under tools/variants/ (VGAUDIO_HIP_LIBRARY), with checksums of the outputs: for the
frames-per-LDS-tile experiments (tools/build_variants.sh with VARIED="gc_decode_kernel adx_kernels")."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, ctypes as C, torch
sys.path.insert(0, %r)
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); d = torch.device("cuda:0"); n = 2880000; nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st = torch.cuda.current_stream().cuda_stream
pcm = vdev.synth_pcm(nch, n, d); coefs = vdev.gc_coefs(pcm, n); adpcm = vdev.gc_encode(pcm, n, coefs)
back = vdev.alloc_pcm(nch, n, d)
def t(f):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
gd = t(lambda: vdev.gc_decode(adpcm, coefs, n, out=back))
cg = int(back.view(torch.int64).sum().item()) if back.numel() %% 4 == 0 else 0
p = _lib.AdxParams(); L.vga_adx_default_params(C.byref(p))
nb = L.vga_adx_encoded_byte_count(n, C.byref(p)); pitch = (nb + 15) // 16 * 16
adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=d); hist = torch.zeros(nch, dtype=torch.int16, device=d)
status = torch.zeros(1, dtype=torch.int32, device=d)
ae = t(lambda: _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st)))
ad = t(lambda: _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), back.data_ptr(), back.stride(0), status.data_ptr(), st)))
ca = int(adx.view(torch.int64).sum().item()); cd = int(back.view(torch.int64).sum().item()) if back.numel() %% 4 == 0 else 0
print("gc_decode %%.2f ms | adx_encode %%.2f ms  adx_decode %%.2f ms | checksums %%d %%d %%d status %%d" %% (gd, ae, ad, cg, ca, cd, int(status.item())))
''' % ROOT
for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "libvga_*.so"))):
    env = dict(os.environ)
    if lib:
        env["VGAUDIO_HIP_LIBRARY"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD] + sys.argv[1:], env=env, capture_output=True, text=True, timeout=300)
    print("%-28s %s" % (os.path.basename(lib) if lib else "product", (r.stdout.strip() or r.stderr.strip()[-400:])), flush=True)
