"""The integer identities vgaudio_amd/csrc/adx_kernels.hip relies on (adx_quantise_step, adx_prescan30, the nibble
gathering of adx_encode_frame_packed), checked on the CPU against the reference's own arithmetic as the oracle restates it
(CriAdxCodec.cs:107-171): exhaustively where the domain is small, on random operands elsewhere.  No GPU, no product code:
the kernel's formulas are written out again here in numpy int64 and wrapped to 32 bits where the kernel is."""
import numpy as np

M32 = (1 << 32) - 1


def _trunc_div(a, b):
    """C# integer division (truncates toward zero)"""
    q = np.abs(a) // b
    return np.where(a < 0, -q, q)


def test_scale_short_to_nibble_is_a_floor_division_with_one_reciprocal():
    # CriAdxCodec.cs:167-171: sign = Math.Sign(sample); sample = (sample + 2340 * sign) / 4681; Clamp4
    s = np.arange(-32768, 32768, dtype=np.int64)
    ref = np.clip(_trunc_div(s + 2340 * np.sign(s), 4681), -8, 7)
    # the kernel: u = (uint32)(s * 57346 + 35107 * 57346) >> 28, q = u - 7
    u = ((s * 57346 + 35107 * 57346) & M32) >> 28
    assert u.min() >= 0 and u.max() <= 14
    assert np.array_equal(u - 7, ref)
    # and the product the kernel forms in 32 bits never needs more than 32 (as an unsigned value)
    assert ((s + 35107) * 57346).max() < (1 << 32) and (s + 35107).min() > 0
    # the reciprocal: exact for every quotient the comment claims (k <= 48)
    t = np.arange(0, 49 * 4681, dtype=np.int64)
    assert np.array_equal((t * 57346) >> 28, t // 4681)


def test_raw_distance_with_one_multiply_add():
    # rawDistance = x - ((b c0 >> 12) + (a c1 >> 12))  (:124-126, >> is C#'s arithmetic shift = floor)
    rng = np.random.default_rng(7)
    n = 2_000_000
    x = rng.integers(-32768, 32768, n)
    a = rng.integers(-32768, 32768, n)
    b = rng.integers(-32768, 32768, n)
    for c0, c1 in ((7400, -3342), (0x1CC0, -3328), (0x1880, -3520), (16384, -16384), (-16384, 16384), (0x0F00, 0)):
        ref = x - (((b * c0) >> 12) + ((a * c1) >> 12))
        xb = x - ((a * c1) >> 12)
        k = (xb << 12) + 4095
        v = k - b * c0
        assert np.abs(v).max() < (1 << 31) and np.abs(k).max() < (1 << 31)          # the kernel's int32 holds it
        assert np.array_equal(v >> 12, ref)


def test_reconstruction_with_the_offset_nibble():
    # Clamp16(Clamp16(scale * q) + predicted) == Clamp16(scale * (q + 7) + (predicted - 7 scale)); scale <= 4096, |q| <= 7
    rng = np.random.default_rng(8)
    n = 1_000_000
    scale = rng.integers(1, 4097, n)
    q = rng.integers(-7, 8, n)
    pred = rng.integers(-(1 << 19), 1 << 19, n)
    ref = np.clip(np.clip(scale * q, -32768, 32767) + pred, -32768, 32767)
    assert np.array_equal(np.clip(scale * (q + 7) + (pred - 7 * scale), -32768, 32767), ref)


def test_nibbles_gathered_eight_to_a_dword():
    # frame bytes 2..17: byte = (q[2k] << 4) | (q[2k + 1] & 0xF) (:143-146); the kernel gathers u = q + 7 a nibble at a
    # time, first sample on top, adds 0x11111111, flips 0x88888888 and reverses the bytes
    rng = np.random.default_rng(9)
    q = rng.integers(-7, 8, (100_000, 8))
    want = np.zeros(len(q), dtype=np.int64)
    for k in range(4):
        byte = ((q[:, 2 * k] & 0xF) << 4) | (q[:, 2 * k + 1] & 0xF)
        want |= byte << (8 * k)                                                    # little-endian dword in memory order
    acc = np.zeros(len(q), dtype=np.int64)
    for j in range(8):
        acc = ((acc << 4) + (q[:, j] + 7)) & M32
    v = ((acc + 0x11111111) & M32) ^ 0x88888888
    swapped = ((v & 0xFF) << 24) | ((v & 0xFF00) << 8) | ((v >> 8) & 0xFF00) | (v >> 24)
    assert np.array_equal(swapped, want)


def test_prescan_maximum_from_the_signed_extremes():
    # max over the samples of |Clamp16(d)| (:112-118) == max(Clamp16(max d), -Clamp16(min d)), zero included as both start at 0
    rng = np.random.default_rng(10)
    d = rng.integers(-200_000, 200_000, (200_000, 30))
    d[::7] //= 50                                                                  # rows that never reach the clamp
    d[::11] = np.abs(d[::11])                                                      # rows without a negative distance
    ref = np.abs(np.clip(d, -32768, 32767)).max(axis=1)
    hi = np.maximum(d.max(axis=1), 0)
    lo = np.minimum(d.min(axis=1), 0)
    assert np.array_equal(np.maximum(np.clip(hi, -32768, 32767), -np.clip(lo, -32768, 32767)), ref)
