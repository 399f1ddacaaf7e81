"""Analysis, not a test: the GC-ADPCM encoder's data-dependent rates on the signal classes of vgaudio_amd/signals.py, from the
oracle's frame encoder on the CPU -- how many (frame, predictor) pairs take a third quantise pass (GcAdpcmEncoder.cs:127-170),
how many frames have one, how often that predictor wins the frame.  Next to the device-side counters of bench.py's
signal_sensitivity block (vga_testing_gc_encode_stats).
    python tests/host/analysis/signal_class_stats.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "third_trip_stats.so")
subprocess.run(["gcc", "-O2", "-fwrapv", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "gc_third_trip_stats.c"), "-lm", "-lpthread"], check=True)
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
from vgaudio_amd import signals, synth  # noqa: E402

L = ctypes.CDLL(SO)


class S(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint64) for n in "frames pair_frames third_pairs frames_with_third third_wins third_wins_strict fourth_pairs".split()] + [("hist", ctypes.c_uint64 * 16)]


n = 14 * 20000
for cls in ("synthetic",) + signals.CLASSES:
    tot = S()
    p_none = []
    for ch in range(0, 64, 8):
        pcm = (synth.generate(1, n, first_channel=ch) if cls == "synthetic" else signals.host(cls, 1, n, first_channel=ch))[0].copy()
        coefs = np.zeros(16, np.int16)
        L.vgo_gc_calculate_coefficients(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p))
        st = S()
        L.third_trip_stats(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st))
        for f, _ in S._fields_[:-1]:
            setattr(tot, f, getattr(tot, f) + getattr(st, f))
        p_none.append(1.0 - st.frames_with_third / st.frames)
    wave = 1.0 - float(np.prod(p_none))            # eight channels share a wave: any of them with a third trip costs the wave one
    print("%-18s third pairs %6.3f %%   frames with one %6.2f %%   (eight channels a wave: %5.1f %% of wave-frames)   won by it %6.3f %% of frames   fourth trips %d"
          % (cls, 100 * tot.third_pairs / tot.pair_frames, 100 * tot.frames_with_third / tot.frames, 100 * wave,
             100 * tot.third_wins / tot.frames, tot.fourth_pairs), flush=True)
