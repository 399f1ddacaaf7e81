"""Signal classes for the sensitivity measurements (bench.py `signal_sensitivity`, tests/test_gpu_signal_classes.py).

The encoders' speed depends on the data -- third trips of GcAdpcmEncoder's retry loop, how soon the seams between time pieces
close -- and the synthetic generator (synth.py) is one family of signals.  These are the others the path is measured and
checked on: the reference's own benchmark tone, both ends of the loudness range, silence, a clipped wave, and the slowest
channel of the synthetic set on every row.  Every class is counter-based integer arithmetic (or a table built once in f64 on
the host), so `host()` (numpy) and `device()` (torch on the GPU) produce the same bits.

  sine440           (short)(32767 sin(2 pi 440 i / 48000)): VGAudio.Benchmark/AdpcmBenchmarks/EncodeBenchmarks.cs:8-24; channel c
                    starts 37 c samples into the tone
  white_full_scale  uniform over the whole int16 range, a hash of (channel, sample)
  noise_3lsb        uniform in [-3, 3]
  silence           zeros (GenerateAdpcmEmpty, VGAudio.Tests/GenerateAudio.cs:73-88: zero coefficients, zero bytes)
  clipped_square    +32767 / -32768, period 32 + c % 97 samples
  slow_channel_93   channel 93 of the synthetic set (an 11.9 kHz tone whose seams take ~1750 frames to close) on every row
"""
import numpy as np

from . import synth

CLASSES = ("sine440", "white_full_scale", "noise_3lsb", "silence", "clipped_square", "slow_channel_93")
SINE_PERIOD = 1200                     # 440 Hz at 48 kHz: eleven cycles in 1200 samples
_M32 = 0xFFFFFFFF


def sine_table():
    i = np.arange(SINE_PERIOD, dtype=np.float64)
    return (32767.0 * np.sin(2.0 * np.pi * 440.0 * i / 48000.0)).astype(np.int16)      # C#'s (short): truncation toward zero


def _hash32_np(c, i):
    """uint32 hash of (channel, sample index): arrays broadcast; 32-bit multiplies done in uint64 and masked"""
    idx = (i.astype(np.uint64) + c.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(_M32)
    h = (idx * np.uint64(0x85EBCA77)) & np.uint64(_M32)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0xC2B2AE3D)) & np.uint64(_M32)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0x27D4EB2F)) & np.uint64(_M32)
    h ^= h >> np.uint64(16)
    return h


def host(cls, nch, n, first_channel=0):
    """[nch, n] int16 on the host"""
    c = (np.arange(nch, dtype=np.int64) + first_channel)[:, None]
    i = np.arange(n, dtype=np.int64)[None, :]
    if cls == "sine440":
        return sine_table()[(i + 37 * c) % SINE_PERIOD]
    if cls == "white_full_scale":
        return ((_hash32_np(c, i) & np.uint64(0xFFFF)).astype(np.int64) - 32768).astype(np.int16)
    if cls == "noise_3lsb":
        return ((_hash32_np(c, i) % np.uint64(7)).astype(np.int64) - 3).astype(np.int16)
    if cls == "silence":
        return np.zeros((nch, n), dtype=np.int16)
    if cls == "clipped_square":
        p = 32 + c % 97
        return np.where((i % p) < p // 2, 32767, -32768).astype(np.int16)
    if cls == "slow_channel_93":
        return np.repeat(synth.generate(1, n, first_channel=93), nch, axis=0)
    raise ValueError(cls)


def device(cls, nch, n, dev, first_channel=0, out=None, chunk=32):
    """[nch, pitch] int16 on the GPU (pitch as device.alloc_pcm), the same bits as host()"""
    import torch

    from . import device as vdev
    if out is None:
        out = vdev.alloc_pcm(nch, n, dev)
    if cls == "silence":
        out.zero_()
        return out
    if cls == "slow_channel_93":
        row = vdev.synth_pcm(1, n, dev, first_channel=93)
        out[:, :row.shape[1]] = row
        return out
    i = torch.arange(n, dtype=torch.int64, device=dev)[None, :]
    table = torch.from_numpy(sine_table()).to(dev) if cls == "sine440" else None
    for c0 in range(0, nch, chunk):
        c1 = min(c0 + chunk, nch)
        c = (torch.arange(c0, c1, dtype=torch.int64, device=dev) + first_channel)[:, None]
        if cls == "sine440":
            v = table[(i + 37 * c) % SINE_PERIOD]
        elif cls == "clipped_square":
            p = 32 + c % 97
            v = torch.where((i % p) < p // 2, 32767, -32768).to(torch.int16)
        else:
            idx = (i + c * 0x9E3779B1) & _M32
            h = (idx * 0x85EBCA77) & _M32
            h = h ^ (h >> 15)
            h = (h * 0xC2B2AE3D) & _M32
            h = h ^ (h >> 13)
            h = (h * 0x27D4EB2F) & _M32
            h = h ^ (h >> 16)
            v = ((h & 0xFFFF) - 32768 if cls == "white_full_scale" else (h % 7) - 3).to(torch.int16)
        out[c0:c1, :n] = v
    return out
