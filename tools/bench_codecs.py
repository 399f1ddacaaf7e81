#!/usr/bin/env python3
"""Secondary measurements (not the headline bench): device-resident kernel times for the other
BASELINE configs' shapes -- GC-ADPCM decode, CRI ADX encode/decode (config 3), CRI HCA
encode/decode (config 4).  HIP-event timing on torch's current stream; prints one JSON object.

    python tools/bench_codecs.py [--channels 4096] [--seconds 60] [--streams 1024]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=3):
    import torch
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in evs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--streams", type=int, default=1024)
    args = ap.parse_args()
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    nch, n = args.channels, int(args.seconds * 48000)
    out = {}

    pcm = vdev.synth_pcm(nch, n, dev)
    coefs = vdev.gc_coefs(pcm, n)
    adpcm = vdev.gc_encode(pcm, n, coefs)
    dec = vdev.alloc_pcm(nch, n, dev)
    ms = timed(lambda: vdev.gc_decode(adpcm, coefs, n, out=dec))
    out["gc_decode"] = {"ms": round(ms, 2), "Msamples/s": round(nch * n / ms / 1e3, 1),
                        "GB/s_algorithmic": round((2 + 8 / 14) * nch * n / ms / 1e6, 1)}
    del adpcm, dec, coefs

    # ADX (config 3)
    p = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(p))
    nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
    pitch = (nb + 15) // 16 * 16
    adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=dev)
    hist = torch.zeros(nch, dtype=torch.int16, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    dec = vdev.alloc_pcm(nch, n, dev)
    ms_e = timed(lambda: _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(),
                                                            adx.stride(0), hist.data_ptr(), st())))
    ms_d = timed(lambda: _lib.check(L.vga_adx_decode_device(adx.data_ptr(), adx.stride(0), nb, nch, n, C.byref(p),
                                                            dec.data_ptr(), dec.stride(0), status.data_ptr(), st())))
    for k, ms in (("adx_encode", ms_e), ("adx_decode", ms_d)):
        out[k] = {"ms": round(ms, 2), "Msamples/s": round(nch * n / ms / 1e3, 1),
                  "GB/s_algorithmic": round(2.5625 * nch * n / ms / 1e6, 1)}
    del adx, dec, pcm

    # HCA (config 4): nstreams stereo streams
    ns = args.streams
    hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0)
    info = _lib.HcaInfoC()
    _lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
    spcm = vdev.synth_pcm(ns * 2, n, dev)                     # [ns*2, pitch]: stream-major planar
    ch_pitch = spcm.stride(0)
    fbytes = info.frame_count * info.frame_size
    fpitch = (fbytes + 8 + 15) // 16 * 16
    frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=dev)
    ms_e = timed(lambda: _lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * ch_pitch, ch_pitch, ns, n, C.byref(info),
                                                            frames.data_ptr(), fpitch, status.data_ptr(), st())))
    wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dec = torch.zeros_like(spcm)
    ms_d = timed(lambda: _lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, dec.data_ptr(),
                                                            2 * ch_pitch, ch_pitch, ws.data_ptr(), wsb, status.data_ptr(), st())))
    for k, ms in (("hca_encode", ms_e), ("hca_decode", ms_d)):
        out[k] = {"ms": round(ms, 2), "Mchannel-samples/s": round(ns * 2 * n / ms / 1e3, 1),
                  "GB/s_algorithmic": round((2 + 682 / 2048) * ns * 2 * n / ms / 1e6, 1)}
    del spcm, frames, dec, ws
    # End to end through the host-buffer C ABI (what the P/Invoke shim calls): H2D + coefficient search +
    # encode + D2H inside one vga_gcadpcm_encode_batch call, pageable host memory, one HIP stream.
    import time
    import numpy as np
    e2e_ch = min(512, nch)
    host = vdev.synth_pcm(e2e_ch, n, dev)[:, :n].cpu().numpy()
    rows = [np.ascontiguousarray(host[c]) for c in range(e2e_ch)]
    nb = L.vga_gcadpcm_sample_count_to_byte_count(n)
    outs = [np.zeros(nb, dtype=np.uint8) for _ in range(e2e_ch)]
    cf = np.zeros(e2e_ch * 16, dtype=np.int16)
    pp = (_lib.i16p * e2e_ch)(*[r.ctypes.data_as(_lib.i16p) for r in rows])
    op = (_lib.u8p * e2e_ch)(*[o.ctypes.data_as(_lib.u8p) for o in outs])
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        _lib.check(L.vga_gcadpcm_encode_batch(pp, e2e_ch, n, 0, 0, cf.ctypes.data_as(_lib.i16p), op))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["gc_encode_batch_host_e2e"] = {"channels": e2e_ch, "ms": round(best * 1e3, 1),
                                       "Msamples/s": round(e2e_ch * n / best / 1e6, 1),
                                       "host_bytes_moved": e2e_ch * (2 * n + nb)}
    out["hca_status"] = int(status.item())
    out["shape"] = {"adpcm_channels": nch, "hca_streams": ns, "samples": n}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
