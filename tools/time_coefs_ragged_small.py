"""Ragged GC-ADPCM coefficient search on batches of N files (log-uniform 1-120 s): one wave per channel (variant 1) against five
waves per channel (variant 3) and the launcher's choice.  GPU box only."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
for nfiles in (16, 768, 1000, 1536, 2304, 3072, 4096, 5000, 6000, 8000):
    lens = [int(np.exp(rng.uniform(np.log(48000.0), np.log(120 * 48000.0)))) for _ in range(nfiles)]
    rb = vdev.GcRaggedBatch(lens, dev); pcm = rb.synth(first_channel=77)
    ws = torch.empty(max(rb.workspace_bytes, 16), dtype=torch.uint8, device=dev)
    out = {}
    for variant in (1, 3, 0):
        L.vga_testing_gc_coefs_variant_this_thread(variant)
        for _ in range(2): c = rb.coefs(pcm, workspace=ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = rb.coefs(pcm, workspace=ws); e1.record(); torch.cuda.synchronize()
        out[variant] = (round(e0.elapsed_time(e1), 2), c.clone())
    L.vga_testing_gc_coefs_variant_this_thread(0)
    print(nfiles, "files: one wave per channel", out[1][0], "ms; five waves per channel", out[3][0], "ms; launcher's choice", out[0][0], "ms; same coefficients",
          bool(torch.equal(out[0][1], out[1][1]) and torch.equal(out[3][1], out[1][1])), flush=True)
    rb.close()
