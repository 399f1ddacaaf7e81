"""Times hca_encode and hca_decode (config 4 shape) on the product library and every library under tools/variants/."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, ctypes as C, torch
sys.path.insert(0, %r)
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); dev = torch.device("cuda:0"); n = 2880000; ns = 1024
hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0); info = _lib.HcaInfoC()
_lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
spcm = vdev.synth_pcm(ns * 2, n, dev); ch_pitch = spcm.stride(0)
fpitch = (info.frame_count * info.frame_size + 8 + 15) // 16 * 16
frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
enc = lambda: _lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * ch_pitch, ch_pitch, ns, n, C.byref(info), frames.data_ptr(), fpitch, status.data_ptr(), st))
wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); out = torch.zeros_like(spcm)
dec = lambda: _lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, out.data_ptr(), 2 * ch_pitch, ch_pitch, ws.data_ptr(), wsb, status.data_ptr(), st))
def t(f):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(2):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
e = t(enc); d = t(dec)
print("hca_encode ms %%.1f  hca_decode ms %%.1f  checksum %%d %%d" %% (e, d, int(frames.to(torch.int64).sum().item()), int(out.to(torch.int64).sum().item())))
''' % ROOT
for lib in [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "libvga_*.so"))):
    env = dict(os.environ)
    if lib:
        env["VGAUDIO_HIP_LIBRARY"] = lib
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-28s %s" % (os.path.basename(lib) if lib else "product", (r.stdout.strip() or r.stderr.strip()[-300:])), flush=True)
