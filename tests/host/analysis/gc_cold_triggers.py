"""Analysis, not a test: per signal class (vgaudio_amd/signals.py), the share of frames in which some predictor of ONE channel
trips each of the conditions that send gc_encode_kernel's frame to its cold block (eight channels share a wave on the GPU: the
wave goes when any of them does).  python tests/host/analysis/gc_cold_triggers.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "gc_cold_triggers.so")
subprocess.run(["g++", "-O2", "-fwrapv", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "gc_cold_triggers.cpp")], check=True)
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
from oracle import pyoracle as po  # noqa: E402
from vgaudio_amd import signals, synth  # noqa: E402

L = ctypes.CDLL(SO)
names = "frames any_rare any_resume any_wide any_cold !coef_ok bump_A bump_B cap_A_ov>3 cap_B_ov>3(used) cap_B_ov>3(unused) winner_wide".split()
n = 14 * 6000
for cls in ("synthetic",) + signals.CLASSES:
    tot = np.zeros(16, dtype=np.uint64)
    for ch in range(0, 64, 8):
        pcm = (synth.generate(1, n, first_channel=ch) if cls == "synthetic" else signals.host(cls, 1, n, first_channel=ch))[0].copy()
        coefs = np.ascontiguousarray(po.gc_calculate_coefficients(pcm), dtype=np.int16)
        st = np.zeros(16, dtype=np.uint64)
        L.cold_triggers(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
        tot += st
    print("%-18s" % cls, "  ".join("%s %.1f%%" % (names[i], 100.0 * tot[i] / tot[0]) for i in range(1, 12)), flush=True)
