#!/usr/bin/env python3
"""Measurements for the SURVEY 8f rows (containers, WAVE transposes, encryption passes): device-resident kernel
times and the HBM traffic they stand for (bytes read + written / time).  Prints one JSON object.

    python tools/bench_containers.py
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=5):
    import torch
    fn()
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
    import torch
    from vgaudio_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = lambda: torch.cuda.current_stream().cuda_stream
    out = {}

    def rec(name, ms, nbytes, **kw):
        out[name] = dict(ms=round(ms, 3), GB_per_s=round(nbytes / ms / 1e6, 1), bytes_moved=int(nbytes), **kw)

    # what a plain device copy reaches on this box (read + write), for scale
    src = torch.empty(1 << 32, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    ms = timed(lambda: dst.copy_(src))
    rec("hbm_copy_ceiling", ms, 2 * src.numel(), note="torch copy_ of 4 GiB")
    del src, dst

    n = 60 * 48000
    # DSP image: 1024 channels x 60 s of GC-ADPCM (1.69 GB image)
    nch = 1024
    nb = L.vga_gcadpcm_sample_count_to_byte_count(n)
    pitch = (nb + 15) // 16 * 16
    adpcm = torch.randint(0, 256, (nch, pitch), dtype=torch.uint8, device=dev)
    coefs = torch.zeros((nch, 16), dtype=torch.int16, device=dev)
    p = _lib.DspParamsC(48000, n, 0, 0, 0, 0x3800, 1, 1)
    lay = _lib.DspLayoutC()
    _lib.check(L.vga_dsp_layout_for(C.byref(p), nch, C.byref(lay)))
    image = torch.empty(lay.file_size, dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(L.vga_dsp_write_device(adpcm.data_ptr(), pitch, nb, coefs.data_ptr(), None, None, None, nch,
                                                         C.byref(p), image.data_ptr(), st())))
    rec("dsp_image", ms, nch * nb + 2 * lay.file_size, channels=nch, note="memset + header + interleave (read once, image written twice)")
    del adpcm, image

    # ADX image: 255 channels x 60 s (413 MB image), 18-byte frames -> 2-byte granules
    nch = 255
    ap = _lib.AdxParams()
    L.vga_adx_default_params(C.byref(ap))
    anb = L.vga_adx_encoded_byte_count(n, C.byref(ap))
    apitch = (anb + 15) // 16 * 16
    audio = torch.randint(0, 256, (nch, apitch), dtype=torch.uint8, device=dev)
    hist = torch.zeros(nch, dtype=torch.int16, device=dev)
    fp = _lib.AdxFileParamsC(48000, n, 0, 0, 0, 0, 18, 4, 3, 500, 0, 1)
    al = _lib.AdxFileLayoutC()
    _lib.check(L.vga_adx_file_layout_for(C.byref(fp), nch, C.byref(al)))
    image = torch.empty(al.file_size, dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(L.vga_adx_write_device(audio.data_ptr(), apitch, anb, hist.data_ptr(), nch, C.byref(fp),
                                                         image.data_ptr(), st())))
    rec("adx_image", ms, nch * anb + 2 * al.file_size, channels=nch)
    del image

    # ADX encryption pass in place, config-3 shape (4096 channels)
    nch = 4096
    audio = torch.randint(0, 256, (nch, apitch), dtype=torch.uint8, device=dev)
    key = _lib.AdxKeyC()
    L.vga_adx_key_from_string(b"karaage", C.byref(key))
    ms = timed(lambda: _lib.check(L.vga_adx_crypt_device(audio.data_ptr(), apitch, anb, nch, C.byref(key), 8, 18, st())))
    rec("adx_crypt", ms, nch * anb, channels=nch, note="reads every frame (emptiness test), writes 2 of 18 bytes")
    del audio

    # WAVE transposes: 64 channels x 60 s (369 MB of samples)
    nch = 64
    inter = torch.randint(0, 256, (n * nch * 2,), dtype=torch.uint8, device=dev)
    ppitch = (n + 7) // 8 * 8
    planar = torch.empty((nch, ppitch), dtype=torch.int16, device=dev)
    ms = timed(lambda: _lib.check(L.vga_wave_deinterleave_pcm16_device(inter.data_ptr(), n, nch, planar.data_ptr(), ppitch, st())))
    rec("wave_deinterleave", ms, 4 * n * nch, channels=nch)
    wp = _lib.WaveParamsC(48000, n, 0, 0, 0)
    size = L.vga_wave_file_size(C.byref(wp), nch)
    wfile = torch.empty(size, dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(L.vga_wave_write_pcm16_device(planar.data_ptr(), ppitch, nch, C.byref(wp), wfile.data_ptr(), st())))
    rec("wave_write", ms, 4 * n * nch, channels=nch, note="includes a synchronous 136-byte header upload")
    del inter, planar, wfile

    # HCA encryption pass, config-4 shape (1024 stereo streams, 2813 frames of 682 bytes)
    ns, fc, fs = 1024, 2813, 682
    fpitch = (fc * fs + 15) // 16 * 16
    frames = torch.randint(0, 256, (ns, fpitch), dtype=torch.uint8, device=dev)
    import numpy as np
    dec, enc = np.zeros(256, np.uint8), np.zeros(256, np.uint8)
    _lib.check(L.vga_hca_key_tables(56, 0xCC55463930DBE1AB, dec.ctypes.data_as(_lib.u8p), enc.ctypes.data_as(_lib.u8p)))
    ms = timed(lambda: _lib.check(L.vga_hca_crypt_device(frames.data_ptr(), fpitch, ns, fc, fs, enc.ctypes.data_as(_lib.u8p), st())))
    rec("hca_crypt", ms, 2 * ns * fc * fs, streams=ns, note="includes the 256-byte table upload and a stream sync")
    info = _lib.HcaInfoC()
    hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0)
    _lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
    fsz = L.vga_hca_file_size(C.byref(info))
    files = torch.empty((ns, fsz + 14), dtype=torch.uint8, device=dev)
    ms = timed(lambda: _lib.check(L.vga_hca_write_device(C.byref(info), frames.data_ptr(), fpitch, ns, None, 1.0, 0, 0,
                                                         files.data_ptr(), fsz + 14, st())))
    rec("hca_images", ms, 2 * ns * info.frame_count * info.frame_size, streams=ns)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
