#!/usr/bin/env python3
"""Times gc_coefs at configs[1] (4096 x 60 s) with every library given (tools/build_variants.sh builds of the
VGA_COEFS_PRIO experiment), one subprocess per library, and prints a digest of the coefficients next to the time.
GPU box only.   python tools/time_coefs_prio.py tools/variants/libvga_*.so"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    nch, n = int(os.environ.get("VGA_T_CH", "4096")), 60 * 48000
    pcm = vdev.synth_pcm(nch, n, dev)
    ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=dev)
    for _ in range(2):
        c = vdev.gc_coefs(pcm, n, workspace=ws)
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = vdev.gc_coefs(pcm, n, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        times.append(round(e0.elapsed_time(e1), 3))
    print(json.dumps({"library": os.path.basename(_lib.SO_PATH), "channels": nch, "coefs_ms": times,
                      "sha256": hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest()[:16]}), flush=True)


if __name__ == "__main__":
    if os.environ.get("VGA_T_CHILD"):
        child()
    else:
        for lib in [None] + sys.argv[1:]:
            env = dict(os.environ, VGA_T_CHILD="1")
            if lib:
                env["VGAUDIO_HIP_LIBRARY"] = os.path.abspath(lib)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=600)
