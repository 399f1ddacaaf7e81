"""Device-resident entry points: torch CUDA(HIP) tensors in, tensors out, on torch's
current stream.  torch is plumbing only (HBM allocations + streams); all compute
is the C ABI's *_device functions (include/vgaudio_hip.h).
"""
import numpy as np
import torch

from . import _lib, synth
from ._lib import check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype):
    if not (t.is_cuda and t.dtype == dtype and t.dim() == 2 and t.stride(1) == 1):
        raise _lib.ArgumentError(f"need a 2-D {dtype} CUDA tensor with unit inner stride")


def gc_byte_count(n):
    return _lib.lib().vga_gcadpcm_sample_count_to_byte_count(int(n))


def alloc_adpcm(nch, sample_count, device):
    """[nch, pitch] uint8 with pitch rounded up to 16 (the kernels store 8-byte frames)."""
    pitch = (gc_byte_count(sample_count) + 15) // 16 * 16
    return torch.zeros((nch, max(pitch, 16)), dtype=torch.uint8, device=device)


def alloc_pcm(nch, n, device):
    pitch = (n + 7) // 8 * 8
    return torch.zeros((nch, max(pitch, 8)), dtype=torch.int16, device=device)


def gc_coefs(pcm, length, workspace=None):
    """pcm: [nch, pitch] int16 CUDA tensor (first `length` samples of each row are valid)."""
    _req(pcm, torch.int16)
    nch = pcm.shape[0]
    L = _lib.lib()
    need = L.vga_gcadpcm_coefs_workspace_bytes(nch, length)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=pcm.device)
    coefs = torch.empty((nch, 16), dtype=torch.int16, device=pcm.device)
    check(L.vga_gcadpcm_coefs_device(pcm.data_ptr(), pcm.stride(0), nch, length, coefs.data_ptr(),
                                     workspace.data_ptr(), workspace.numel(), _stream()))
    return coefs


def gc_encode(pcm, sample_count, coefs, out=None, hist1=None, hist2=None):
    _req(pcm, torch.int16)
    nch = pcm.shape[0]
    if out is None:
        out = alloc_adpcm(nch, sample_count, pcm.device)
    check(_lib.lib().vga_gcadpcm_encode_device(
        pcm.data_ptr(), pcm.stride(0), nch, sample_count, coefs.data_ptr(),
        hist1.data_ptr() if hist1 is not None else None, hist2.data_ptr() if hist2 is not None else None,
        out.data_ptr(), out.stride(0), _stream()))
    return out


def gc_decode(adpcm, coefs, sample_count, out=None, hist1=None, hist2=None):
    _req(adpcm, torch.uint8)
    nch = adpcm.shape[0]
    if out is None:
        out = alloc_pcm(nch, sample_count, adpcm.device)
    status = torch.zeros(1, dtype=torch.int32, device=adpcm.device)
    check(_lib.lib().vga_gcadpcm_decode_device(
        adpcm.data_ptr(), adpcm.stride(0), coefs.data_ptr(), nch, sample_count,
        hist1.data_ptr() if hist1 is not None else None, hist2.data_ptr() if hist2 is not None else None,
        out.data_ptr(), out.stride(0), status.data_ptr(), _stream()))
    return out, status


def synth_pcm(nch, n, device, first_channel=0, out=None):
    """Synthetic PCM16 generated on the device; bit-identical to synth.generate()."""
    if out is None:
        out = alloc_pcm(nch, n, device)
    params = np.array([synth.channel_params(first_channel + k) for k in range(nch)], dtype=np.uint32)
    d_params = torch.from_numpy(params.view(np.int32)).to(device)
    check(_lib.lib().vga_synth_pcm16_device(out.data_ptr(), out.stride(0), nch, n, first_channel,
                                            d_params.data_ptr(), _stream()))
    return out


class GcRaggedBatch:
    """A device-resident ragged GC-ADPCM batch (vga_gcadpcm_ragged, include/vgaudio_hip.h): channels of different
    lengths in packed device buffers.  `pcm_offsets` / `adpcm_offsets` (numpy int64) say where each channel's row
    starts in the flat tensors alloc_pcm() / alloc_adpcm() return."""

    def __init__(self, sample_counts, device):
        import ctypes as C
        self.counts = np.ascontiguousarray(sample_counts, dtype=np.int32)
        self.nch = int(self.counts.shape[0])
        self.device = device
        L = _lib.lib()
        h = C.c_void_p()
        with torch.cuda.device(device):
            check(L.vga_gcadpcm_ragged_create(self.counts.ctypes.data_as(C.POINTER(C.c_int)), self.nch, C.byref(h)))
        self.handle = h
        self.pcm_offsets = np.zeros(self.nch, dtype=np.int64)
        self.adpcm_offsets = np.zeros(self.nch, dtype=np.int64)
        check(L.vga_gcadpcm_ragged_offsets(h, self.pcm_offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                           self.adpcm_offsets.ctypes.data_as(C.POINTER(C.c_int64))))
        self.pcm_samples = int(L.vga_gcadpcm_ragged_pcm_samples(h))
        self.adpcm_bytes = int(L.vga_gcadpcm_ragged_adpcm_bytes(h))
        self.workspace_bytes = int(L.vga_gcadpcm_ragged_coefs_workspace_bytes(h))
        self.byte_counts = np.array([L.vga_gcadpcm_sample_count_to_byte_count(int(n)) for n in self.counts], dtype=np.int64)

    def close(self):
        if self.handle is not None:
            _lib.lib().vga_gcadpcm_ragged_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def alloc_pcm(self):
        return torch.zeros(self.pcm_samples, dtype=torch.int16, device=self.device)

    def alloc_adpcm(self):
        return torch.zeros(self.adpcm_bytes, dtype=torch.uint8, device=self.device)

    def put_pcm(self, flat, channels):
        """copies host channels (list of int16 arrays) into the packed tensor"""
        host = np.zeros(self.pcm_samples, dtype=np.int16)
        for c, a in enumerate(channels):
            host[self.pcm_offsets[c]:self.pcm_offsets[c] + self.counts[c]] = a
        flat.copy_(torch.from_numpy(host))
        return flat

    def synth(self, first_channel=0, out=None):
        """the bench's synthetic channels first_channel + c, each cut to its own length, generated in place"""
        flat = out if out is not None else self.alloc_pcm()
        L = _lib.lib()
        params = torch.from_numpy(np.array([synth.channel_params(first_channel + c) for c in range(self.nch)],
                                           dtype=np.uint32).reshape(self.nch, 4).view(np.int32)).to(self.device)
        for c in range(self.nch):                      # one launch per channel: set-up code, not the timed path
            n = int(self.counts[c])
            if n:
                check(L.vga_synth_pcm16_device(flat.data_ptr() + 2 * int(self.pcm_offsets[c]), max(n, 8), 1, n, first_channel + c,
                                               params[c].data_ptr(), _stream()))
        return flat

    def coefs(self, pcm, workspace=None):
        L = _lib.lib()
        if workspace is None or workspace.numel() < self.workspace_bytes:
            workspace = torch.empty(max(self.workspace_bytes, 16), dtype=torch.uint8, device=self.device)
        coefs = torch.empty((self.nch, 16), dtype=torch.int16, device=self.device)
        check(L.vga_gcadpcm_coefs_device_v(self.handle, pcm.data_ptr(), coefs.data_ptr(), workspace.data_ptr(), workspace.numel(),
                                           _stream()))
        return coefs

    def encode(self, pcm, coefs, out=None, hist1=None, hist2=None):
        if out is None:
            out = self.alloc_adpcm()
        check(_lib.lib().vga_gcadpcm_encode_device_v(
            self.handle, pcm.data_ptr(), coefs.data_ptr(), hist1.data_ptr() if hist1 is not None else None,
            hist2.data_ptr() if hist2 is not None else None, out.data_ptr(), _stream()))
        return out

    def decode(self, adpcm, coefs, out=None, hist1=None, hist2=None):
        if out is None:
            out = self.alloc_pcm()
        status = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(_lib.lib().vga_gcadpcm_decode_device_v(
            self.handle, adpcm.data_ptr(), coefs.data_ptr(), hist1.data_ptr() if hist1 is not None else None,
            hist2.data_ptr() if hist2 is not None else None, out.data_ptr(), status.data_ptr(), _stream()))
        return out, status

    def rows(self, flat, offsets, sizes):
        """the channels' rows of a packed tensor as a list of numpy arrays"""
        host = flat.cpu().numpy()
        return [host[int(o):int(o) + int(n)] for o, n in zip(offsets, sizes)]
