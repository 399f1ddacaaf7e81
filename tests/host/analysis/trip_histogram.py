"""Analysis, not a test: the distribution of quantise passes per (frame, predictor) on the bench's synthetic channels
(GcAdpcmEncoder.cs:127-170's do-while) -- what fraction of the passes the encoder's waves execute is wanted by the lane
that executes it.
    python tests/host/analysis/trip_histogram.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "third_trip_stats.so")
subprocess.run(["gcc", "-O2", "-fwrapv", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "gc_third_trip_stats.c"), "-lm", "-lpthread"], check=True)
sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
from vgaudio_amd import synth  # noqa: E402

L = ctypes.CDLL(SO)
n = 14 * 20000
hist, fmax = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
for ch in range(0, 128, 3):
    pcm = synth.generate(1, n, first_channel=ch)[0].copy()
    coefs = np.zeros(16, np.int16)
    L.vgo_gc_calculate_coefficients(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p))
    h, m = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    L.trip_histogram(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p), h.ctypes.data_as(ctypes.c_void_p),
                     m.ctypes.data_as(ctypes.c_void_p))
    hist += h
    fmax += m
tot = hist.sum()
print("passes per (frame, predictor):", {t: "%.2f%%" % (100.0 * int(hist[t]) / int(tot)) for t in range(1, 8) if hist[t]})
print("mean passes per pair: %.3f" % (sum(t * int(hist[t]) for t in range(16)) / int(tot)))
print("slowest predictor of a frame:", {t: "%.2f%%" % (100.0 * int(fmax[t]) / int(fmax.sum())) for t in range(1, 8) if fmax[t]})
