#!/usr/bin/env python3
"""Times the five-waves-per-channel coefficient kernel (variant 3) on every library given: uniform batches of 1 / 256 / 768
channels x 60 s and the ragged launches of 16 / 1536 / 4096 files (log-uniform 1-120 s), with a digest of the coefficients.
GPU box only.   python tools/time_coefs_solo.py tools/variants/libvga_solo*.so"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    L.vga_testing_gc_coefs_variant_this_thread(3)
    out = {"library": os.path.basename(_lib.SO_PATH)}
    h = hashlib.sha256()

    def timed(f):
        for _ in range(2):
            c = f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = f()
        e1.record()
        torch.cuda.synchronize()
        h.update(c.cpu().numpy().tobytes())
        return round(e0.elapsed_time(e1), 2)
    n = 60 * 48000
    for nch in (1, 256, 768):
        pcm = vdev.synth_pcm(nch, n, dev)
        ws = torch.empty(max(L.vga_gcadpcm_coefs_workspace_bytes(nch, n), 16), dtype=torch.uint8, device=dev)
        out[f"uniform_{nch}_ms"] = timed(lambda: vdev.gc_coefs(pcm, n, workspace=ws))
        del pcm, ws
    rng = np.random.default_rng(5)
    for nfiles in (16, 1536, 4096):
        lens = [int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0)))) for _ in range(nfiles)]
        rb = vdev.GcRaggedBatch(lens, dev)
        pcm = rb.synth(first_channel=77)
        ws = torch.empty(max(rb.workspace_bytes, 16), dtype=torch.uint8, device=dev)
        out[f"ragged_{nfiles}_ms"] = timed(lambda: rb.coefs(pcm, workspace=ws))
        rb.close()
        del pcm, ws
    out["sha256"] = h.hexdigest()[:16]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("VGA_T_CHILD"):
        child()
    else:
        for lib in [None] + sys.argv[1:]:
            env = dict(os.environ, VGA_T_CHILD="1")
            if lib:
                env["VGAUDIO_HIP_LIBRARY"] = os.path.abspath(lib)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=600)
