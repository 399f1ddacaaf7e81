#!/usr/bin/env python3
"""Times the three ToPcm16 entry points through host pointers (vga_gcadpcm_decode_batch, vga_adx_decode_batch,
vga_hca_decode_batch) at the BASELINE sizes, with the pipeline's default shape and with direct (page-locked) downloads,
and checks the PCM against the device-resident decoders.  GPU box only.
    python tools/time_decode_batches.py [--channels 4096] [--streams 1024] [--seconds 60] [--modes "0,0,0,0;1,1,0,-1"]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--modes", default="0,0,0,0;1,1,0,-1")
    ap.add_argument("--codecs", default="gc,adx,hca")
    ap.add_argument("--pieces", type=int, default=0, help="force the decoders' time pieces per channel (0 = the launcher's choice)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from vgaudio_amd import _lib, device as vdev
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n = int(args.seconds * 48000)
    modes = [tuple(int(v) for v in m.split(",")) for m in args.modes.split(";")]
    L.vga_testing_gc_encoder_segments_this_thread(args.pieces)

    def timed(name, call, in_bytes, out_bytes, check):
        for mode in modes:
            L.vga_testing_host_pipeline_this_thread(*mode)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                call()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            print(json.dumps({"call": name, "pipeline": ",".join(str(v) for v in mode), "ms": round(best * 1e3, 1),
                              "host_GB_in": round(in_bytes / 1e9, 2), "host_GB_out": round(out_bytes / 1e9, 2),
                              "identical_to_device_path": bool(check())}), flush=True)
        L.vga_testing_host_pipeline_this_thread(0, 0, 0, 0)

    def rows(a, ptype):
        return (ptype * a.shape[0])(*[a[i].ctypes.data_as(ptype) for i in range(a.shape[0])])

    def to_host(t, width, dtype):
        out = np.empty((t.shape[0], width), dtype=dtype)
        for c0 in range(0, t.shape[0], 256):
            out[c0:c0 + 256] = t[c0:c0 + 256, :width].cpu().numpy()
        return out

    if "gc" in args.codecs:
        nch = args.channels
        pcm = vdev.synth_pcm(nch, n, dev)
        coefs = vdev.gc_coefs(pcm, n)
        adpcm = vdev.gc_encode(pcm, n, coefs)
        nb = vdev.gc_byte_count(n)
        want, _ = vdev.gc_decode(adpcm, coefs, n)
        h_adpcm = to_host(adpcm, nb, np.uint8)
        h_coefs = coefs.cpu().numpy().reshape(-1).copy()
        out = np.zeros((nch, n), dtype=np.int16)
        ip, op = rows(h_adpcm, _lib.u8p), rows(out, _lib.i16p)
        del pcm, adpcm
        torch.cuda.empty_cache()
        timed("vga_gcadpcm_decode_batch", lambda: _lib.check(L.vga_gcadpcm_decode_batch(ip, h_coefs.ctypes.data_as(_lib.i16p), nch, n, None, None, op)),
              nch * nb, nch * n * 2, lambda: np.array_equal(out[::97], want[::97, :n].cpu().numpy()))
        del want, out, h_adpcm
        torch.cuda.empty_cache()
    if "adx" in args.codecs:
        nch = args.channels
        pcm = vdev.synth_pcm(nch, n, dev)
        p = _lib.AdxParams()
        L.vga_adx_default_params(C.byref(p))
        nb = L.vga_adx_encoded_byte_count(n, C.byref(p))
        pitch = (nb + 15) // 16 * 16
        adx = torch.zeros((nch, pitch), dtype=torch.uint8, device=dev)
        hist = torch.zeros(nch, dtype=torch.int16, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        want = vdev.alloc_pcm(nch, n, dev)
        _lib.check(L.vga_adx_encode_device(pcm.data_ptr(), pcm.stride(0), nch, n, C.byref(p), adx.data_ptr(), pitch, hist.data_ptr(), st))
        _lib.check(L.vga_adx_decode_device(adx.data_ptr(), pitch, nb, nch, n, C.byref(p), want.data_ptr(), want.stride(0), status.data_ptr(), st))
        torch.cuda.synchronize()
        h_adx = to_host(adx, nb, np.uint8)
        out = np.zeros((nch, n), dtype=np.int16)
        ip, op = rows(h_adx, _lib.u8p), rows(out, _lib.i16p)
        del pcm, adx
        torch.cuda.empty_cache()
        timed("vga_adx_decode_batch", lambda: _lib.check(L.vga_adx_decode_batch(ip, nb, nch, n, C.byref(p), op)),
              nch * nb, nch * n * 2, lambda: np.array_equal(out[::97], want[::97, :n].cpu().numpy()))
        del want, out, h_adx
        torch.cuda.empty_cache()
    if "hca" in args.codecs:
        ns = args.streams
        hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0)
        info = _lib.HcaInfoC()
        _lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
        spcm = vdev.synth_pcm(ns * 2, n, dev)
        chp = spcm.stride(0)
        fbytes = info.frame_count * info.frame_size
        fpitch = (fbytes + 8 + 15) // 16 * 16
        frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * chp, chp, ns, n, C.byref(info), frames.data_ptr(), fpitch, status.data_ptr(), st))
        wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        want = torch.zeros_like(spcm)
        _lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, want.data_ptr(), 2 * chp, chp, ws.data_ptr(), wsb, status.data_ptr(), st))
        torch.cuda.synchronize()
        h_fr = to_host(frames, fbytes, np.uint8)
        out = np.zeros((ns * 2, n), dtype=np.int16)
        ip, op = rows(h_fr, _lib.u8p), rows(out, _lib.i16p)
        del spcm, frames, ws
        torch.cuda.empty_cache()
        timed("vga_hca_decode_batch", lambda: _lib.check(L.vga_hca_decode_batch(C.byref(info), ip, ns, op)),
              ns * fbytes, ns * 2 * n * 2, lambda: np.array_equal(out[::53], want[::53, :n].cpu().numpy()))


if __name__ == "__main__":
    main()
