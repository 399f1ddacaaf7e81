"""CRI ADX in plain Python, written from the reference's C# (Codecs/CriAdx/CriAdxCodec.cs,
CriAdxParameters.cs, Formats/CriAdx/CriAdxHelpers.cs).  Test infrastructure
(see oracle/pyref/__init__.py)."""
import math
from dataclasses import dataclass

from .csharp import SIGNED_NIBBLE, clamp4, clamp16, combine_nibbles, div_round_up, i16, i32, tdiv, to_int

FIXED, LINEAR, EXPONENTIAL = 2, 3, 4                      # CriAdxType.cs

FIXED_COEFS = ((0, 0), (0x0F00, 0), (0x1CC0, i16(0xF300)), (0x1880, i16(0xF240)))   # CriAdxCodec.cs:186-191


@dataclass
class Params:                                             # CriAdxParameters.cs
    sample_rate: int = 48000
    highpass_frequency: int = 500
    frame_size: int = 18
    version: int = 4
    history: int = 0
    padding: int = 0
    type: int = LINEAR
    filter: int = 0


def calculate_coefficients(highpass_freq, sample_rate):   # CriAdxCodec.cs:173-184
    sqrt2 = math.sqrt(2)
    a = sqrt2 - math.cos(2.0 * math.pi * highpass_freq / sample_rate)
    b = sqrt2 - 1
    c = (a - math.sqrt((a + b) * (a - b))) / b
    return i16(to_int(c * 8192)), i16(to_int(c * c * -4096))


def sample_count_to_byte_count(sample_count, frame_size):  # CriAdxHelpers.cs:20-33
    frames, extra = divmod(sample_count, frame_size * 2 - 4)
    nibbles = frame_size * 2 * frames + (0 if extra == 0 else extra + 4)
    return nibbles // 2 + (nibbles & 1)


def _log2(value):                                         # Helpers.Log2 (floor) for value >= 1
    return value.bit_length() - 1


def _calculate_scale(max_distance, exponential):          # :149-165 -> (scale, gain, scale_to_write)
    scale = tdiv(max_distance - 1, 7) + 1
    if scale > 0x1000:
        scale = 0x1000
    to_write = scale - 1
    if exponential:
        power = 0 if to_write == 0 else _log2(to_write) + 1
        scale = 1 << power
        to_write = 12 - power
        max_distance = 8 * scale - 1
    gain = 0.0 if max_distance == 0 else 32767.0 / max_distance
    return scale, gain, to_write


def _short_to_nibble(sample):                             # :167-171
    sign = (sample > 0) - (sample < 0)
    return clamp4(tdiv(sample + 2340 * sign, 4681))


def encode_frame(pcm, coefs, samples_per_frame, adx_type, version):
    """EncodeFrame (:106-147); pcm (history + frame, ints) is updated in place.  Returns the frame bytes."""
    c0, c1 = coefs
    max_distance = 0
    for i in range(samples_per_frame):
        predicted = (pcm[i + 1] * c0 >> 12) + (pcm[i] * c1 >> 12)
        distance = abs(clamp16(pcm[i + 2] - predicted))
        if distance > max_distance:
            max_distance = distance
    scale, gain, to_write = _calculate_scale(max_distance, adx_type == EXPONENTIAL)
    nibbles = []
    for i in range(samples_per_frame):
        predicted = (pcm[i + 1] * c0 >> 12) + (pcm[i] * c1 >> 12)
        raw = pcm[i + 2] - predicted
        scaled = clamp16(to_int(raw * gain))
        q = _short_to_nibble(scaled)
        nibbles.append(q)
        decoded_distance = clamp16(scale * q)
        if version == 4:
            predicted = (pcm[i + 1] * c0 + pcm[i] * c1) >> 12
        pcm[i + 2] = clamp16(decoded_distance + predicted)
    out = bytearray([(to_write >> 8) & 0x1F, to_write & 0xFF])
    for i in range(samples_per_frame // 2):
        out.append(combine_nibbles(nibbles[2 * i], nibbles[2 * i + 1]))
    return out


def encode(pcm, p):
    """CriAdxCodec.Encode (:56-104).  Updates p.history like the reference; returns the bytes."""
    pcm = [int(v) for v in pcm]
    sample_count = len(pcm) + p.padding
    spf = (p.frame_size - 2) * 2
    frame_count = div_round_up(sample_count, spf)
    padding_remaining = p.padding
    coefs = FIXED_COEFS[p.filter] if p.type == FIXED else calculate_coefficients(500, p.sample_rate)
    buf = [0] * (spf + 2)
    out = bytearray(frame_count * p.frame_size)
    if p.version == 4 and p.padding == 0:
        buf[0] = buf[1] = pcm[0]
        p.history = pcm[0]
    for i in range(frame_count):
        to_copy = min(sample_count - i * spf, spf)
        start = 2
        if padding_remaining != 0:
            while padding_remaining > 0 and to_copy > 0:
                padding_remaining -= 1
                to_copy -= 1
                start += 1
            if to_copy == 0:
                continue
        src = max(i * spf - p.padding, 0)
        buf[start:start + to_copy] = pcm[src:src + to_copy]
        clear = spf - to_copy - start + 2
        buf[start + to_copy:start + to_copy + clear] = [0] * clear
        frame = encode_frame(buf, coefs, spf, p.type, p.version)
        if p.type == FIXED:
            frame[0] |= (p.filter << 5) & 0xFF
        out[i * p.frame_size:(i + 1) * p.frame_size] = frame
        buf[0], buf[1] = buf[spf], buf[spf + 1]
    return bytes(out)


def decode(adpcm, sample_count, p=None):
    """CriAdxCodec.Decode (:9-54)."""
    p = p or Params()
    spf = (p.frame_size - 2) * 2
    table = FIXED_COEFS if p.type == FIXED else (calculate_coefficients(p.highpass_frequency, p.sample_rate),)
    pcm = []
    hist1 = hist2 = p.history
    start = p.padding % spf if p.padding > 0 else 0
    pos = tdiv(p.padding, spf) * p.frame_size
    for _ in range(div_round_up(sample_count, spf)):
        filt = ((adpcm[pos] >> 4) & 0xF) >> 1
        scale = i16(((adpcm[pos] << 8) | adpcm[pos + 1]) & 0x1FFF)
        scale = i16(1 << (12 - scale)) if p.type == EXPONENTIAL else i16(scale + 1)
        pos += 2 + start // 2
        c0, c1 = table[filt]
        to_read = min(spf, sample_count - len(pcm))
        for s in range(start, to_read):
            if s % 2 == 0:
                nib = SIGNED_NIBBLE[(adpcm[pos] >> 4) & 0xF]
            else:
                nib = SIGNED_NIBBLE[adpcm[pos] & 0xF]
                pos += 1
            if p.version == 4:
                sample = scale * nib + ((hist1 * c0 + hist2 * c1) >> 12)
            else:
                sample = scale * nib + (hist1 * c0 >> 12) + (hist2 * c1 >> 12)
            sample = clamp16(sample)
            hist2, hist1 = hist1, sample
            pcm.append(sample)
        start = 0
    return pcm + [0] * (sample_count - len(pcm))      # `new short[sampleCount]`: what padding skipped stays 0
