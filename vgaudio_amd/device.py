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
