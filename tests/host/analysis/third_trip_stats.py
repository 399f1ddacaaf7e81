"""Analysis, not a test: on the bench's synthetic channels, how often does a predictor that needs a third quantise pass
(GcAdpcmEncoder.cs:127-170) win the frame (:66-76)?  Numbers in LABNOTES.md section 6, item 1.
    python tests/host/analysis/third_trip_stats.py"""
import os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "third_trip_stats.so")
subprocess.run(["gcc", "-O2", "-fwrapv", "-shared", "-fPIC", "-o", SO, os.path.join(HERE, "gc_third_trip_stats.c"), "-lm", "-lpthread"], check=True)
import ctypes, numpy as np, sys
sys.path.insert(0, os.path.join(HERE, '..', '..', '..'))
from vgaudio_amd import synth
L=ctypes.CDLL(SO)
class S(ctypes.Structure):
    _fields_=[(n,ctypes.c_uint64) for n in "frames pair_frames third_pairs frames_with_third third_wins third_wins_strict fourth_pairs".split()]+[("hist",ctypes.c_uint64*16)]
n=14*20000
tot=S()
for ch in list(range(0,128,3)):
    pcm=synth.generate(1,n,first_channel=ch)[0].copy()
    coefs=np.zeros(16,np.int16)
    L.vgo_gc_calculate_coefficients(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p))
    st=S()
    L.third_trip_stats(pcm.ctypes.data_as(ctypes.c_void_p), n, coefs.ctypes.data_as(ctypes.c_void_p), ctypes.byref(st))
    print(ch, st.frames, "third pairs %.3f%%"%(100*st.third_pairs/st.pair_frames), "frames with third %.1f%%"%(100*st.frames_with_third/st.frames), "third wins / frames %.3f%%"%(100*st.third_wins/st.frames), "wins per third-frame %.1f%%"%(100*st.third_wins/max(1,st.frames_with_third)), "4th", st.fourth_pairs)
    for f,_ in S._fields_[:-1]: setattr(tot,f,getattr(tot,f)+getattr(st,f))
print("TOTAL third pairs %.3f%% ; frames with third %.2f%% ; frames won by a third-trip predictor %.3f%%"%(100*tot.third_pairs/tot.pair_frames,100*tot.frames_with_third/tot.frames,100*tot.third_wins/tot.frames))
