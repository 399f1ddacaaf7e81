"""Integer-only, counter-based synthetic PCM16 generator.

Host (numpy) and device (csrc/synth_kernels.hip) produce identical bits: every
operation is 32/64-bit integer arithmetic on (channel, sample index), no libm.

    x(c, i) = sat16( ((tri(i*f_c) * A_c >> 15) + (tri(i*3*f_c + phi_c) * (A_c/3) >> 15)
                      + lp4(noise(c, i))) * env(c, i) >> 15 )

tri  : 32-bit-phase triangle wave in [-32768, 32767]
f_c  : one of 96 semitone-spaced phase increments (55 Hz .. ~13.3 kHz at 48 kHz)
A_c  : 4000 + hash(c) % 20001
lp4  : mean of 4 consecutive splitmix64-hashed noise values in [-2048, 2047]
env  : slow triangle LFO in [8192, 32767] so frames span loud and quiet passages
"""
import numpy as np

SEED = 0x5EED
N_FREQ = 96
_MASK64 = (1 << 64) - 1


_FT = None


def freq_table():
    """96 phase increments; inc[0] = 55 Hz @ 48 kHz; each step * 69433/65536 (~2^(1/12))."""
    global _FT
    if _FT is None:
        inc = [4921183]  # round(55 / 48000 * 2**32)
        for _ in range(N_FREQ - 1):
            inc.append((inc[-1] * 69433) >> 16)
        _FT = np.array(inc, dtype=np.uint32)
    return _FT


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _tri(phase_u32):
    q = (phase_u32 >> np.uint32(15)).astype(np.int64)  # 0 .. 131071
    return np.where(q < 65536, q - 32768, 98303 - q)


def channel_params(c):
    """(f_inc, phi, A, lfo_inc) for channel c -- all uint32/ints."""
    with np.errstate(over="ignore"):
        h = int(_splitmix64(np.uint64((SEED << 32) ^ c)))
    ft = freq_table()
    f_inc = int(ft[c % N_FREQ])
    phi = (h >> 32) & 0xFFFFFFFF
    amp = 4000 + (h & 0xFFFFFFFF) % 20001
    lfo = 2000 + ((h >> 20) & 0x3FFF)  # period ~ 2^32/lfo samples (~5-45 s)
    return f_inc, phi, amp, lfo


def generate(nch, n, first_channel=0, first_sample=0):
    """Return int16 [nch, n] synthetic PCM."""
    out = np.empty((nch, n), dtype=np.int16)
    i = (np.arange(n, dtype=np.uint64) + np.uint64(first_sample))
    i32 = i.astype(np.uint32)
    with np.errstate(over="ignore"):
        for k in range(nch):
            c = first_channel + k
            f_inc, phi, amp, lfo = channel_params(c)
            p1 = i32 * np.uint32(f_inc)
            p2 = i32 * np.uint32((3 * f_inc) & 0xFFFFFFFF) + np.uint32(phi)
            s = (_tri(p1) * amp >> 15) + (_tri(p2) * (amp // 3) >> 15)
            base = np.uint64(((SEED << 32) ^ c) & _MASK64) * np.uint64(0x100000001B3)
            nz = np.zeros(n, dtype=np.int64)
            for d in range(4):
                idx = i - np.uint64(d)  # wraps for i < d, still deterministic
                hh = _splitmix64(base ^ idx)
                nz += (hh & np.uint64(4095)).astype(np.int64) - 2048
            s = s + (nz >> 2)
            env = 20480 + (_tri(i32 * np.uint32(lfo)) * 12287 >> 15)  # 8193 .. 32766
            s = (s * env) >> 15
            out[k] = np.clip(s, -32768, 32767).astype(np.int16)
    return out


def sine(n, freq=440.0, sample_rate=48000):
    """The reference's own benchmark/test shape:
    (short)(short.MaxValue * Math.Sin(2*pi*f*i/sr))  (Tests/GenerateAudio.cs:23-33,
    Benchmark/AdpcmBenchmarks/EncodeBenchmarks.cs:17)."""
    c = 2 * np.pi * freq / sample_rate
    return np.trunc(32767 * np.sin(c * np.arange(n))).astype(np.int16)
