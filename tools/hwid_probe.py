"""Wave placement census: which SIMD does wave 0 / wave 1 of each workgroup land on?"""
import ctypes as C, collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import _lib
_lib._preload_torch_hip_runtime()
L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libhwid.so"))
L.hwid_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
for blocks, threads, lds in ((1024, 128, 29696), (2048, 128, 29696), (1024, 64, 29696), (1024, 192, 29696), (1024, 256, 29696)):
    wpb = threads // 64
    out = np.zeros((blocks * wpb, 2), dtype=np.uint32)
    rc = L.hwid_probe(blocks, threads, lds, out.ctypes.data)
    hw = out[:, 0]; xcc = out[:, 1] & 0xF
    wave_slot = hw & 0xF; simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    cuid = xcc.astype(np.int64) * 4096 + se.astype(np.int64) * 256 + sh * 16 + cu
    print(f"blocks {blocks} x {threads} thr, lds {lds}: rc {rc}, distinct CUs {len(set(cuid.tolist()))}")
    for w in range(wpb):
        sel = np.arange(blocks) * wpb + w
        print(f"   wave {w}: SIMD histogram {np.bincount(simd[sel], minlength=4).tolist()}")
    # serial waves (wave 0) per (CU, SIMD)
    cnt = collections.Counter(zip(cuid[::wpb].tolist(), simd[::wpb].tolist()))
    print("   wave-0 count per (CU,SIMD): histogram", sorted(collections.Counter(cnt.values()).items()))
    per_cu = collections.Counter(cuid[::wpb].tolist())
    print("   workgroups per CU: histogram", sorted(collections.Counter(per_cu.values()).items()))
