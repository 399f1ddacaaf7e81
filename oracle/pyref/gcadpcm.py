"""GC-ADPCM (Nintendo DSP-ADPCM) in plain Python, written from the reference's C#.

Sources: Codecs/GcAdpcm/GcAdpcmCoefficients.cs, GcAdpcmEncoder.cs, GcAdpcmDecoder.cs,
GcAdpcmMath.cs.  Test infrastructure (see oracle/pyref/__init__.py).
"""
import math

import numpy as np

from .csharp import SIGNED_NIBBLE, clamp4, clamp16, combine_nibbles, div_round_up, i32, tdiv, to_int

SAMPLES_PER_FRAME = 14
BYTES_PER_FRAME = 8
NIBBLES_PER_FRAME = 16


# ---------------------------------------------------------------- GcAdpcmMath.cs:11-47
def nibble_count_to_sample_count(nibbles):
    frames, extra = divmod(nibbles, NIBBLES_PER_FRAME)
    return SAMPLES_PER_FRAME * frames + (0 if extra < 2 else extra - 2)


def sample_count_to_nibble_count(samples):
    frames, extra = divmod(samples, SAMPLES_PER_FRAME)
    return NIBBLES_PER_FRAME * frames + (0 if extra == 0 else extra + 2)


def sample_count_to_byte_count(samples):
    n = sample_count_to_nibble_count(samples)
    return n // 2 + (n & 1)


def byte_count_to_sample_count(nbytes):
    return nibble_count_to_sample_count(nbytes * 2)


# ---------------------------------------------------------------- GcAdpcmCoefficients.cs
class _Lpc:
    """The per-frame analysis of CalculateCoefficients (:40-61): returns the record (r1, r2) or None."""

    @staticmethod
    def inner_product(buf):                                   # :112-120, buf = 28 shorts, frame at 14
        out = [0.0, 0.0, 0.0]
        for i in range(3):
            acc = 0.0
            for x in range(14):
                acc -= buf[14 + x - i] * buf[14 + x]
            out[i] = acc
        return out

    @staticmethod
    def outer_product(buf):                                   # :122-131 (rows/cols 1..2 of a 3x3)
        m = [[0.0] * 3 for _ in range(3)]
        for x in (1, 2):
            for y in (1, 2):
                acc = 0.0
                for z in range(14):
                    acc += buf[14 + z - x] * buf[14 + z - y]
                m[x][y] = acc
        return m

    @staticmethod
    def analyze_ranges(m, idx):                               # :133-208; True = reject
        recips = [0.0, 0.0, 0.0]
        for x in (1, 2):
            val = max(abs(m[x][1]), abs(m[x][2]))
            if val < 5e-324:
                return True
            recips[x] = 1.0 / val
        max_index = 0
        for i in (1, 2):
            for x in range(1, i):
                tmp = m[x][i]
                for y in range(1, x):
                    tmp -= m[x][y] * m[y][i]
                m[x][i] = tmp
            val = 0.0
            for x in range(i, 3):
                tmp = m[x][i]
                for y in range(1, i):
                    tmp -= m[x][y] * m[y][i]
                m[x][i] = tmp
                tmp = abs(tmp) * recips[x]
                if tmp >= val:
                    val = tmp
                    max_index = x
            if max_index != i:
                for y in (1, 2):
                    m[max_index][y], m[i][y] = m[i][y], m[max_index][y]
                recips[max_index] = recips[i]
            idx[i] = max_index
            if i != 2:
                tmp = 1.0 / m[i][i]
                for x in range(i + 1, 3):
                    m[x][i] *= tmp
        lo, hi = 1.0e10, 0.0
        for i in (1, 2):
            tmp = abs(m[i][i])
            if tmp < lo:
                lo = tmp
            if tmp > hi:
                hi = tmp
        return _div(lo, hi) < 1.0e-10

    @staticmethod
    def bidirectional_filter(m, idx, vec):                    # :210-237
        x = 0
        for i in (1, 2):
            index = idx[i]
            tmp = vec[index]
            vec[index] = vec[i]
            if x != 0:
                for y in range(x, i):
                    tmp -= vec[y] * m[i][y]
            elif tmp != 0.0:
                x = i
            vec[i] = tmp
        for i in (2, 1):
            tmp = vec[i]
            for y in range(i + 1, 3):
                tmp -= vec[y] * m[i][y]
            vec[i] = _div(tmp, m[i][i])
        vec[0] = 1.0

    @staticmethod
    def quadratic_merge(vec):                                 # :239-255; True = reject
        v2 = vec[2]
        tmp = 1.0 - (v2 * v2)
        if tmp == 0.0:
            return True
        v0 = _div(vec[0] - (v2 * v2), tmp)
        v1 = _div(vec[1] - (vec[1] * v2), tmp)
        vec[0], vec[1] = v0, v1
        return abs(v1) > 1.0


def _div(a, b):
    """IEEE double division (Python raises on /0; C# yields inf/NaN)"""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def _finish_record(v):                                        # :257-283 (both overloads)
    for z in (1, 2):
        if v[z] >= 1.0:
            v[z] = 0.9999999999
        elif v[z] <= -1.0:
            v[z] = -0.9999999999
    return [1.0, (v[2] * v[1]) + v[1], v[2]]


def _matrix_filter(rec):                                      # :285-305 -> dst
    m = [[0.0] * 3 for _ in range(3)]
    m[2][0] = 1.0
    m[2][1] = -rec[1]
    m[2][2] = -rec[2]
    for i in (2, 1):
        val = 1.0 - (m[i][i] * m[i][i])
        for y in range(1, i + 1):
            m[i - 1][y] = _div((m[i][i] * m[i][y]) + m[i][y], val)
    dst = [1.0, 0.0, 0.0]
    for i in (1, 2):
        acc = 0.0
        for y in range(1, i + 1):
            acc += m[i][y] * dst[i - y]
        dst[i] = acc
    return dst


def _merge_finish_record(src, dst):                           # :307-333 (dst updated in place)
    tmp = [0.0, 0.0, 0.0]
    val = src[0]
    dst[0] = 1.0
    for i in (1, 2):
        v2 = 0.0
        for y in range(1, i):
            v2 += dst[y] * src[i - y]
        dst[i] = _div(-(v2 + src[i]), val) if val > 0.0 else 0.0
        tmp[i] = dst[i]
        for y in range(1, i):
            dst[y] += dst[i] * dst[i - y]
        val *= 1.0 - (dst[i] * dst[i])
    dst[:] = _finish_record(tmp)


def _contrast_vectors(s1, rec):                               # :335-342
    val = _div(rec[2] * rec[1] + -rec[1], 1.0 - rec[2] * rec[2])
    val1 = (s1[0] * s1[0]) + (s1[1] * s1[1]) + (s1[2] * s1[2])
    val2 = (s1[0] * s1[1]) + (s1[1] * s1[2])
    val3 = s1[0] * s1[2]
    return val1 + (2.0 * val * val2) + (2.0 * (-rec[1] * val + -rec[2]) * val3)


def _filter_records(vec_best, exp, records):                  # :344-396
    for _ in range(2):
        counts = [0] * 8
        sums = [[0.0, 0.0, 0.0] for _ in range(8)]
        for rec in records:
            index, value = 0, 1.0e30
            for i in range(exp):
                t = _contrast_vectors(vec_best[i], rec)
                if t < value:
                    value, index = t, i
            counts[index] += 1
            d = _matrix_filter(rec)
            for i in range(3):
                sums[index][i] += d[i]
        for i in range(exp):
            if counts[i] > 0:
                for y in range(3):
                    sums[i][y] /= counts[i]
        for i in range(exp):
            _merge_finish_record(sums[i], vec_best[i])


def _round_to_short(d):                                       # :94-108
    if d > 0.0:
        return 32767 if d > 32767 else int(round(d))
    if d < -32768:
        return -32768
    if d != d:
        return 0                                              # (short)Math.Round(NaN): cvttsd2si -> 0x80000000 -> low 16 bits
    return int(round(d))


def calculate_coefficients(source):
    """GcAdpcmCoefficients.CalculateCoefficients (:9-110): 16 shorts for one channel."""
    source = [int(v) for v in source]
    n = len(source)
    hist = [0] * 28
    records = []
    for start in range(0, n, 14):
        frame = source[start:start + 14]
        hist[14:28] = frame + [0] * (14 - len(frame))
        vec = _Lpc.inner_product(hist)
        if abs(vec[0]) > 10.0:
            m = _Lpc.outer_product(hist)
            idx = [0, 0, 0]
            if not _Lpc.analyze_ranges(m, idx):
                _Lpc.bidirectional_filter(m, idx, vec)
                if not _Lpc.quadratic_merge(vec):
                    records.append(_finish_record(vec))
        hist[0:14] = hist[14:28]

    vec1 = [1.0, 0.0, 0.0]
    best = [[0.0, 0.0, 0.0] for _ in range(8)]
    for rec in records:
        best[0] = _matrix_filter(rec)
        vec1[1] += best[0][1]
        vec1[2] += best[0][2]
    for y in (1, 2):
        vec1[y] = _div(vec1[y], float(len(records)))          # 0/0 = NaN when no record survived
    _merge_finish_record(vec1, best[0])

    exp = 1
    for w in range(3):
        for i in range(exp):
            for y, v2 in enumerate((0.0, -1.0, 0.0)):
                best[exp + i][y] = (0.01 * v2) + best[i][y]
        exp = 1 << (w + 1)
        _filter_records(best, exp, records)

    coefs = []
    for z in range(8):
        coefs.append(_round_to_short(-best[z][1] * 2048.0))
        coefs.append(_round_to_short(-best[z][2] * 2048.0))
    return coefs


# ---------------------------------------------------------------- GcAdpcmEncoder.cs
_HALF = float(np.float32(0.4999999))


def _encode_coef(pcm_in, c0, c1):
    """DspEncodeCoef (:96-171) for one predictor; pcm_in = 16 ints (two history + 14)."""
    out = [0] * 16
    q = [0] * 14
    out[0], out[1] = pcm_in[0], pcm_in[1]
    max_distance = 0
    for s in range(14):
        predicted = tdiv(i32(pcm_in[s] * c1 + pcm_in[s + 1] * c0), 2048)
        distance = clamp16(i32(pcm_in[s + 2] - predicted))
        if abs(distance) > abs(max_distance):
            max_distance = distance
    scale_power = 0
    while scale_power <= 12 and (max_distance > 7 or max_distance < -8):
        max_distance = tdiv(max_distance, 2)
        scale_power += 1
    scale_power = -1 if scale_power <= 1 else scale_power - 2

    hazard = False
    while True:
        scale_power += 1
        pass_scale_power = scale_power
        scale = (1 << scale_power) * 2048
        fscale = np.float32(scale)
        total = 0.0
        max_overflow = 0
        for s in range(14):
            input_sample = pcm_in[s + 2] * 2048
            predicted = i32(out[s] * c1 + out[s + 1] * c0)
            distance = i32(input_sample - predicted)
            scaled = float(np.float32(distance) / fscale)      # (double)((float)distance / scale)
            unclamped = to_int(scaled + _HALF) if distance > 0 else to_int(scaled - _HALF)
            sample = clamp4(unclamped)
            if sample != unclamped:
                overflow = abs(unclamped - sample)
                if overflow > max_overflow:
                    max_overflow = overflow
            q[s] = sample
            corrected = i32(predicted + sample * scale)
            out[s + 2] = clamp16((corrected + 1024) >> 11)
            actual = float(pcm_in[s + 2] - out[s + 2])
            total += actual * actual
        x = max_overflow + 8
        while x > 256:
            scale_power += 1
            if scale_power >= 12:
                scale_power = 11
            x >>= 1
        if not (scale_power < 12 and max_overflow > 1):
            break
        if pass_scale_power >= 12:
            # The reference would repeat this identical pass forever (bump loop put scalePower back to 11).
            # Needs an int32 wrap in the predictor, i.e. coefficients CalculateCoefficients cannot produce.
            hazard = True
            break
    return out, q, scale_power, total, hazard


def encode_frame(pcm16, coefs):
    """DspEncodeFrame (:47-94): pcm16 = 16 ints, updated in place; returns the 8 frame bytes."""
    best, best_total = 0, 1.7976931348623157e308      # double.MaxValue
    results = []
    hazard = False
    for i in range(8):
        r = _encode_coef(pcm16, coefs[2 * i], coefs[2 * i + 1])
        results.append(r)
        hazard |= r[4]
        if r[3] < best_total:
            best_total, best = r[3], i
    out, q, scale, _, _ = results[best]
    for s in range(14):
        pcm16[s + 2] = out[s + 2]
    frame = [combine_nibbles(best, scale)]
    for i in range(7):
        frame.append(combine_nibbles(q[2 * i], q[2 * i + 1]))
    return bytes(frame), hazard


def encode(pcm, coefs, sample_count=None, hist1=0, hist2=0):
    """GcAdpcmEncoder.Encode (:14-45).  Returns (bytes, hazard flag)."""
    pcm = [int(v) for v in pcm]
    n = len(pcm) if sample_count is None or sample_count == -1 else sample_count
    adpcm = bytearray(sample_count_to_byte_count(n))
    buf = [0] * 16
    buf[0], buf[1] = hist2, hist1
    hazard = False
    for f in range(div_round_up(n, 14)):
        k = min(n - f * 14, 14)
        buf[2:2 + k] = pcm[f * 14:f * 14 + k]
        buf[2 + k:16] = [0] * (14 - k)
        frame, hz = encode_frame(buf, coefs)
        hazard |= hz
        nb = sample_count_to_byte_count(k)
        adpcm[f * 8:f * 8 + nb] = frame[:nb]
        buf[0], buf[1] = buf[14], buf[15]
    return bytes(adpcm), hazard


# ---------------------------------------------------------------- GcAdpcmDecoder.cs:10-54
def decode(adpcm, coefs, sample_count=None, hist1=0, hist2=0):
    n = byte_count_to_sample_count(len(adpcm)) if sample_count is None else sample_count
    pcm = []
    pos = 0
    for _ in range(div_round_up(n, 14)):
        ps = adpcm[pos]
        pos += 1
        scale = (1 << (ps & 0xF)) * 2048
        predictor = (ps >> 4) & 0xF
        c1, c2 = coefs[predictor * 2], coefs[predictor * 2 + 1]   # IndexError past predictor 7, as in C#
        for s in range(min(14, n - len(pcm))):
            if s % 2 == 0:
                nib = SIGNED_NIBBLE[(adpcm[pos] >> 4) & 0xF]
            else:
                nib = SIGNED_NIBBLE[adpcm[pos] & 0xF]
                pos += 1
            corrected = i32(i32(c1 * hist1 + c2 * hist2) + i32(scale * nib))
            sample = clamp16((corrected + 1024) >> 11)
            hist2, hist1 = hist1, sample
            pcm.append(sample)
    return pcm
