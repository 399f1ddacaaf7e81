"""Does independent work fill the issue slots the persistent GC-ADPCM encoder leaves idle?

The encoder's persistent workgroups hold 3 x 168 of a SIMD's 512 VGPRs and 4 x 36.9 of a CU's 160 KB of LDS: no wave of
another kernel fits next to them.  With fewer workgroups per CU (VGA_HIP_GC_WGS_PER_CU, a -DVGA_TUNING build:
tools/build_variants.sh tune:"-DVGA_TUNING") the coefficient kernel of ANOTHER batch can run underneath.  Timed here, all at
BASELINE configs[1]'s shape: the encoder alone with 4 / 3 / 2 workgroups per CU, the coefficient search alone, and both at
once on two streams (encoder launched first, and the other way round).  If the pair takes about as long as the encoder
alone, splitting the batch and searching part k + 1 under the encode of part k is worth building; if it takes the sum, the
SIMDs are saturated and the two kernels can only run one after the other.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "tools", "variants", "libvga_tune.so")
if os.path.exists(lib):
    os.environ.setdefault("VGAUDIO_HIP_LIBRARY", lib)
import torch  # noqa: E402

from vgaudio_amd import device as vdev  # noqa: E402

nch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 2880000
d = torch.device("cuda:0")
pcm_a = vdev.synth_pcm(nch, n, d)
pcm_b = vdev.synth_pcm(nch, n, d, first_channel=nch)
ws_a = torch.empty(nch * ((n + 13) // 14) * 16 + 4096, dtype=torch.uint8, device=d)
ws_b = torch.empty_like(ws_a)
coefs_a = vdev.gc_coefs(pcm_a, n, workspace=ws_a)
out = vdev.alloc_adpcm(nch, n, d)
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        rel = [m[0].elapsed_time(m[1]) for m in marks]
        if best is None or wall < best[0]:
            best = (wall, rel)
    return best


def enc_alone():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s1):
        a.record()
        vdev.gc_encode(pcm_a, n, coefs_a, out=out)
        b.record()
    return [(a, b)]


def coefs_alone():
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s2):
        a.record()
        vdev.gc_coefs(pcm_b, n, workspace=ws_b)
        b.record()
    return [(a, b)]


def both(enc_first):
    def run():
        a, b, c, e = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        def enc():
            with torch.cuda.stream(s1):
                a.record()
                vdev.gc_encode(pcm_a, n, coefs_a, out=out)
                b.record()
        def co():
            with torch.cuda.stream(s2):
                c.record()
                vdev.gc_coefs(pcm_b, n, workspace=ws_b)
                e.record()
        (enc(), co()) if enc_first else (co(), enc())
        return [(a, b), (c, e), (a, e) if enc_first else (c, b)]
    return run


def checksum():
    return int(out.view(torch.int64).sum().item())


os.environ["VGA_HIP_GC_WGS_PER_CU"] = "4"
timed(enc_alone, 1)
ref = checksum()
w, r = timed(coefs_alone)
print("coefficient search alone                      wall %7.2f ms   kernel %7.2f" % (w, r[0]), flush=True)
for wgs in (4, 3, 2):
    os.environ["VGA_HIP_GC_WGS_PER_CU"] = str(wgs)
    w, r = timed(enc_alone)
    print("encoder alone, %d workgroups per CU             wall %7.2f ms   kernel %7.2f   %s" % (wgs, w, r[0], "same bytes" if checksum() == ref else "DIFFERENT BYTES"), flush=True)
    for enc_first in (True, False):
        w, r = timed(both(enc_first))
        print("  + coefficient search of another batch (%s)  wall %7.2f ms   encoder %7.2f   coefs %7.2f   first start to last end %7.2f   %s"
              % ("encoder launched first" if enc_first else "coefs launched first  ", w, r[0], r[1], r[2], "same bytes" if checksum() == ref else "DIFFERENT BYTES"), flush=True)
