"""Debug aid (round 5): the edge-input encode test's cases through both encoder wave layouts, first differing frame per case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po
from vgaudio_amd import _lib, synth
from vgaudio_amd.gcadpcm import GcAdpcmEncoder
src = open(os.path.join(ROOT, "tests", "test_gpu_gcadpcm.py")).read()
ns = {}
exec(src[src.index("def _edge_channels"):src.index("def test_coefs_match_oracle_synthetic")], {"np": np, "synth": synth}, ns)
L = _lib.lib()
for n in (4200, 4203):
    rng = np.random.default_rng(11)
    chans = ns["_edge_channels"](n, rng)
    names = list(chans)
    pcm = [chans[k] for k in names]
    real = np.stack([po.gc_calculate_coefficients(p) for p in pcm])
    bounded = rng.integers(-16383, 16384, real.shape).astype(np.int16)
    wrapping = rng.integers(-32768, 32768, real.shape).astype(np.int16)
    for label, coefs in (("real", real), ("bounded", bounded), ("wrapping", wrapping)):
        for layout in (4, 8):
            L.vga_testing_gc_encoder_layout_this_thread(layout)
            try:
                got = GcAdpcmEncoder.Encode(pcm, coefs)
            finally:
                L.vga_testing_gc_encoder_layout_this_thread(0)
            for i, name in enumerate(names):
                want = po.gc_encode(pcm[i], coefs[i])
                bad = np.flatnonzero(got[i] != want)
                if bad.size:
                    f = int(bad[0]) // 8
                    print(n, label, "layout", layout, name, "first byte", int(bad[0]), "frame", f, "got", got[i][8 * f:8 * f + 8].tolist(), "want", want[8 * f:8 * f + 8].tolist(), "differing bytes", bad.size, flush=True)
print("done")
