import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from vgaudio_amd import _lib, device as vdev
L = _lib.lib(); dev = torch.device("cuda:0"); n = 2880000; ns = 1024
hp = _lib.HcaParamsC(2, 0, 0, 2, 48000, n, 0, 0, 0); info = _lib.HcaInfoC()
_lib.check(L.vga_hca_encoder_initialize(C.byref(hp), C.byref(info)))
spcm = vdev.synth_pcm(ns * 2, n, dev); ch_pitch = spcm.stride(0)
fpitch = (info.frame_count * info.frame_size + 8 + 15) // 16 * 16
frames = torch.zeros((ns, fpitch), dtype=torch.uint8, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.check(L.vga_hca_encode_device(spcm.data_ptr(), 2 * ch_pitch, ch_pitch, ns, n, C.byref(info), frames.data_ptr(), fpitch, status.data_ptr(), st))
wsb = L.vga_hca_decode_workspace_bytes(C.byref(info), ns); ws = torch.empty(wsb, dtype=torch.uint8, device=dev); out = torch.zeros_like(spcm)
dec = lambda: _lib.check(L.vga_hca_decode_device(C.byref(info), frames.data_ptr(), fpitch, ns, out.data_ptr(), 2 * ch_pitch, ch_pitch, ws.data_ptr(), wsb, status.data_ptr(), st))
for g in (0, 8, 12, 16, 24, 32, 48, 64):
    L.vga_testing_hca_frames_per_group_this_thread(g)
    dec(); torch.cuda.synchronize(); ts = []
    for _ in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); dec(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("frames per group", g, "decode ms %.2f" % min(ts), "checksum", int(out.to(torch.int64).sum().item()), flush=True)
L.vga_testing_hca_frames_per_group_this_thread(0)
