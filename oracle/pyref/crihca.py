"""CRI HCA encoder and decoder in plain Python, written from the reference's C#.

Sources (under Codecs/CriHca/ unless noted): CriHcaEncoder.cs, CriHcaDecoder.cs, CriHcaPacking.cs,
CriHcaFrame.cs, CriHcaChannel.cs, CriHcaTables.cs, HcaInfo.cs, Utilities/Mdct.cs, Utilities/BitWriter.cs,
Utilities/BitReader.cs, Utilities/Crc16.cs, Formats/CriHca/CriHcaFormat.cs (the streaming loop).
Test infrastructure (see oracle/pyref/__init__.py).

The byte/short tables of CriHcaTables.cs live in a deflate-packed blob in the reference; they are DATA and
come from tests/golden/hca_tables.json (harvested from the blob by tests/golden/make_hca_fixtures.py).  The
generated double tables are computed here with libm, as CriHcaTables.cs:29-77 does with System.Math.
"""
import json
import math
import os
import struct

from .csharp import clamp, clamp16, div_round_up, i32, next_multiple, tdiv, to_int

SUBFRAMES = 8
SUB_BITS = 7
SUB = 1 << SUB_BITS                  # SamplesPerSubFrame
FRAME = SUBFRAMES * SUB              # SamplesPerFrame

DISCRETE, STEREO_PRIMARY, STEREO_SECONDARY = 0, 1, 2
QUALITY = {"NotSet": 0, "Highest": 1, "High": 2, "Middle": 3, "Low": 4, "Lowest": 5}


# ---------------------------------------------------------------- tables (CriHcaTables.cs)
class Tables:
    _loaded = None

    @classmethod
    def get(cls):
        if cls._loaded is None:
            cls._loaded = cls()
        return cls._loaded

    def __init__(self):
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "hca_tables.json")
        packed = json.load(open(path))["packed"]
        self.quantize_spectrum_bits = packed["QuantizeSpectrumBits"]
        self.quantize_spectrum_value = packed["QuantizeSpectrumValue"]
        self.quantized_spectrum_bits = packed["QuantizedSpectrumBits"]
        self.quantized_spectrum_max_bits = packed["QuantizedSpectrumMaxBits"]
        self.quantized_spectrum_value = packed["QuantizedSpectrumValue"]
        self.scale_to_resolution_curve = packed["ScaleToResolutionCurve"]
        self.ath_curve = packed["AthCurve"]
        # stored as float in the blob, widened to double by the unpacker
        self.mdct_window = [struct.unpack(">f", bytes.fromhex(h))[0] for h in packed["MdctWindow"]]
        self.default_channel_mapping = packed["DefaultChannelMapping"]
        self.valid_channel_mappings = packed["ValidChannelMappings"]

        base = math.pow(2, 53.0 / 128)

        def res_max(x):                                        # ResolutionMaxValueFunction
            return x if x < 8 else (1 << (x - 4)) - 1

        def dequant_scale(x):                                  # DequantizerScalingFunction
            return math.sqrt(128) * math.pow(base, x - 63)

        def inv_step(x):
            return res_max(x) + 0.5

        def step(x):
            return 0.0 if x == 0 else 1 / inv_step(x)

        def dead_zone(i):                                      # QuantizerDeadZoneFunction
            bits = struct.unpack("<q", struct.pack("<d", step(i) / 2))[0] - (res_max(i) + 1)
            return struct.unpack("<d", struct.pack("<q", bits))[0]

        self.dequantizer_scaling = [dequant_scale(i) for i in range(64)]
        self.quantizer_step_size = [step(i) for i in range(16)]
        self.quantizer_dead_zone = [dead_zone(i) for i in range(16)]
        self.quantizer_scaling = [1 / dequant_scale(i) for i in range(64)]
        self.quantizer_inverse_step_size = [inv_step(i) for i in range(16)]
        self.resolution_max_values = [res_max(i) for i in range(16)]
        self.intensity_ratio = [(28 - i * 2) / 14.0 for i in range(15)]
        self.intensity_ratio_bounds = [(27 - i * 2) / 14.0 for i in range(14)]
        self.scale_conversion = [math.pow(base, i - 64) if 1 < i < 127 else 0.0 for i in range(128)]


# ---------------------------------------------------------------- Utilities/Mdct.cs
def _bit_reverse(v, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


class Mdct:
    _sin, _cos, _shuffle = {}, {}, {}

    @classmethod
    def _tables(cls, bits):
        if bits not in cls._sin:
            size = 1 << bits
            cls._sin[bits] = [math.sin(math.pi * (4 * i + 1) / (4 * size)) for i in range(size)]
            cls._cos[bits] = [math.cos(math.pi * (4 * i + 1) / (4 * size)) for i in range(size)]
            cls._shuffle[bits] = [_bit_reverse(i ^ (i // 2), bits) for i in range(size)]
        return cls._sin[bits], cls._cos[bits], cls._shuffle[bits]

    def __init__(self, bits, window, scale=1.0):
        self.bits = bits
        self.size = 1 << bits
        self.scale = scale
        self.window = window
        for b in range(bits + 1):
            self._tables(b)
        self.mdct_previous = [0.0] * self.size
        self.imdct_previous = [0.0] * self.size

    def run_mdct(self, x):                                     # :63-92
        size, half, w, prev = self.size, self.size // 2, self.window, self.mdct_previous
        dct_in = [0.0] * size
        for i in range(half):
            a = w[half - i - 1] * -x[half + i]
            b = w[half + i] * x[half - i - 1]
            c = w[i] * prev[i]
            d = w[size - i - 1] * prev[size - i - 1]
            dct_in[i] = a - b
            dct_in[half + i] = c - d
        out = self.dct4(dct_in)
        self.mdct_previous = list(x)
        return out

    def run_imdct(self, x):                                    # :94-120
        size, half, w, prev = self.size, self.size // 2, self.window, self.imdct_previous
        d = self.dct4(x)
        out = [0.0] * size
        for i in range(half):
            out[i] = w[i] * d[i + half] + prev[i]
            out[i + half] = w[i + half] * -d[size - 1 - i] - prev[i + half]
            prev[i] = w[size - 1 - i] * -d[half - i - 1]
            prev[i + half] = w[half - i - 1] * d[i]
        return out

    def dct4(self, x):                                         # :127-181
        bits, size = self.bits, self.size
        sin, cos, shuffle = self._tables(bits)
        t = [0.0] * size
        last = size - 1
        for i in range(size // 2):
            a, b = x[2 * i], x[last - 2 * i]
            t[2 * i] = a * cos[i] + b * sin[i]
            t[2 * i + 1] = a * sin[i] - b * cos[i]
        stages = bits - 1
        for stage in range(stages):
            block_bits = stages - stage
            block = 1 << block_bits
            half_block = block >> 1
            sin, cos, _ = self._tables(block_bits - 1)
            for blk in range(1 << stage):
                for i in range(half_block):
                    f = (blk * block + i) * 2
                    k = f + block
                    a = t[f] - t[k]
                    b = t[f + 1] - t[k + 1]
                    t[f] += t[k]
                    t[f + 1] += t[k + 1]
                    t[k] = a * cos[i] + b * sin[i]
                    t[k + 1] = a * sin[i] - b * cos[i]
        return [t[shuffle[i]] * self.scale for i in range(size)]


# ---------------------------------------------------------------- Utilities/Crc16.cs, BitWriter.cs, BitReader.cs
def _crc_table(poly):
    table = []
    for i in range(256):
        v = i << 8
        for _ in range(8):
            v = ((v << 1) ^ poly) & 0xFFFF if v & 0x8000 else (v << 1) & 0xFFFF
        table.append(v)
    return table


_CRC = _crc_table(0x8005)


def crc16(data, size):
    crc = 0
    for i in range(size):
        crc = ((crc << 8) & 0xFFFF) ^ _CRC[(crc >> 8) ^ data[i]]
    return crc


class BitWriter:
    """Writes MSB-first.  The reference's fast paths OR into the first byte and ASSIGN the following ones;
    on a buffer written strictly front to back that equals plain bit placement, which is what this does --
    except for one thing kept literally: a write leaves the bytes after its last one untouched."""

    def __init__(self, buf):
        self.buf = buf
        self.length_bits = len(buf) * 8
        self.position = 0

    def write(self, value, bit_count):
        if bit_count > self.length_bits - self.position:
            raise ValueError("Not enough bits left in output buffer")
        remaining = self.length_bits - self.position
        byte_index, bit_index = divmod(self.position, 8)
        if bit_count <= 9 and remaining >= 16:
            width = 2
        elif bit_count <= 17 and remaining >= 24:
            width = 3
        elif bit_count <= 25 and remaining >= 32:
            width = 4
        else:
            width = 0
        if width:
            total = width * 8
            out = (((value << (total - bit_count)) & ((1 << total) - 1)) >> bit_index)
            b = out.to_bytes(width, "big")
            self.buf[byte_index] |= b[0]
            for k in range(1, width):
                self.buf[byte_index + k] = b[k]
        else:                                                  # WriteFallback
            bits = bit_count
            while bits > 0:
                if bit_index >= 8:
                    bit_index = 0
                    byte_index += 1
                shift = 8 - bit_index - bits
                shifted = value >> -shift if shift < 0 else value << shift
                n = min(bits, 8 - bit_index)
                mask = ((1 << n) - 1) << (8 - bit_index - n)
                self.buf[byte_index] = (self.buf[byte_index] & ~mask & 0xFF) | (shifted & mask)
                bit_index += n
                bits -= n
        self.position += bit_count

    def align(self, multiple):
        self.write(0, next_multiple(self.position, multiple) - self.position)


class BitReader:
    def __init__(self, buf):
        self.buf = buf
        self.length_bits = len(buf) * 8
        self.position = 0

    @property
    def remaining(self):
        return self.length_bits - self.position

    def peek(self, bit_count):                                 # PeekInt :56-94 (all paths give the same bits)
        rem = self.remaining
        if bit_count > rem:
            if self.position >= self.length_bits:
                return 0
            return self._bits(rem) << (bit_count - rem)
        return self._bits(bit_count)

    def _bits(self, n):
        v = 0
        for k in range(n):
            p = self.position + k
            v = (v << 1) | ((self.buf[p >> 3] >> (7 - (p & 7))) & 1)
        return v

    def read(self, bit_count):
        v = self.peek(bit_count)
        self.position += bit_count
        return v

    def read_offset_binary_positive(self, bit_count):          # ReadOffsetBinary(.., OffsetBias.Positive)
        return self.read(bit_count) - ((1 << (bit_count - 1)) - 1)


# ---------------------------------------------------------------- HcaInfo.cs, CriHcaFrame.cs, CriHcaChannel.cs
class HcaInfo:
    def __init__(self):
        self.channel_count = self.sample_rate = self.sample_count = self.frame_count = 0
        self.inserted_samples = self.appended_samples = self.header_size = self.frame_size = 0
        self.min_resolution = self.max_resolution = 0
        self.track_count = self.channel_config = 0
        self.total_band_count = self.base_band_count = self.stereo_band_count = 0
        self.hfr_band_count = self.bands_per_hfr_group = self.hfr_group_count = 0
        self.looping = False
        self.loop_start_frame = self.loop_end_frame = self.pre_loop_samples = self.post_loop_samples = 0
        self.use_ath_curve = False
        self.comment_length = 0

    @property
    def loop_start_sample(self):
        return self.loop_start_frame * 1024 + self.pre_loop_samples - self.inserted_samples

    def calculate_hfr_values(self):
        if self.bands_per_hfr_group <= 0:
            return
        self.hfr_band_count = self.total_band_count - self.base_band_count - self.stereo_band_count
        self.hfr_group_count = div_round_up(self.hfr_band_count, self.bands_per_hfr_group)


class Channel:
    def __init__(self, ctype, coded_count, tables):
        self.type = ctype
        self.coded = coded_count                               # CodedScaleFactorCount
        self.pcm_float = [[0.0] * SUB for _ in range(SUBFRAMES)]
        self.spectra = [[0.0] * SUB for _ in range(SUBFRAMES)]
        self.scaled_spectra = [[0.0] * SUBFRAMES for _ in range(SUB)]
        self.quantized_spectra = [[0] * SUB for _ in range(SUBFRAMES)]
        self.gain = [0.0] * SUB
        self.intensity = [0] * SUBFRAMES
        self.hfr_scales = [0] * 8
        self.hfr_group_average_spectra = [0.0] * 8
        self.mdct = Mdct(SUB_BITS, tables.mdct_window, math.sqrt(2.0 / SUB))
        self.scale_factors = [0] * SUB
        self.resolution = [0] * SUB
        self.header_length_bits = 0
        self.scale_factor_delta_bits = 0


def channel_types(hca):                                        # CriHcaFrame.GetChannelTypes :35-54
    per_track = hca.channel_count // hca.track_count
    if hca.stereo_band_count == 0 or per_track == 1:
        return [DISCRETE] * 8
    P, S, D = STEREO_PRIMARY, STEREO_SECONDARY, DISCRETE
    cfg = hca.channel_config
    if per_track == 2:
        return [P, S]
    if per_track == 3:
        return [P, S, D]
    if per_track == 4:
        return [P, S, D, D] if cfg != 0 else [P, S, P, S]
    if per_track == 5:
        return [P, S, D, D, D] if cfg > 2 else [P, S, D, P, S]
    if per_track == 6:
        return [P, S, D, D, P, S]
    if per_track == 7:
        return [P, S, D, D, P, S, D]
    if per_track == 8:
        return [P, S, D, D, P, S, P, S]
    return [D] * per_track


class Frame:
    def __init__(self, hca):
        t = Tables.get()
        self.hca = hca
        types = channel_types(hca)
        self.channels = [Channel(types[i],
                                 hca.base_band_count if types[i] == STEREO_SECONDARY
                                 else hca.base_band_count + hca.stereo_band_count, t)
                         for i in range(hca.channel_count)]
        self.ath_curve = self._scale_ath(hca.sample_rate, t) if hca.use_ath_curve else [0] * SUB
        self.acceptable_noise_level = 0
        self.evaluation_boundary = 0

    @staticmethod
    def _scale_ath(frequency, t):                              # :62-84
        ath = [0xFF] * SUB
        acc = 0
        for i in range(SUB):
            acc += frequency
            index = acc >> 13
            if index >= len(t.ath_curve):
                break
            ath[i] = t.ath_curve[index]
        return ath


def calculate_resolution(scale_factor, noise_level):           # CriHcaPacking.cs:58-67
    if scale_factor == 0:
        return 0
    pos = clamp(noise_level - tdiv(5 * scale_factor, 2) + 2, 0, 58)
    return Tables.get().scale_to_resolution_curve[pos]


# ---------------------------------------------------------------- CriHcaEncoder.cs
class Params:                                                  # CriHcaParameters.cs (+ CodecParameters.SampleCount)
    def __init__(self, channel_count, sample_rate, sample_count, quality="High", bitrate=0, limit_bitrate=False,
                 looping=False, loop_start=0, loop_end=0):
        self.channel_count, self.sample_rate, self.sample_count = channel_count, sample_rate, sample_count
        self.quality = QUALITY[quality] if isinstance(quality, str) else quality
        self.bitrate, self.limit_bitrate = bitrate, limit_bitrate
        self.looping, self.loop_start, self.loop_end = looping, loop_start, loop_end


class Encoder:
    def __init__(self, config):                                # Initialize :61-114
        if config.channel_count > 8:
            raise ValueError("HCA channel count must be 8 or below")
        self.t = Tables.get()
        self.cutoff_frequency = config.sample_rate // 2
        self.quality = config.quality
        self.post_samples = 128
        h = self.hca = HcaInfo()
        h.channel_count, h.track_count = config.channel_count, 1
        h.sample_count, h.sample_rate = config.sample_count, config.sample_rate
        h.min_resolution, h.max_resolution = 1, 15
        h.inserted_samples = SUB
        self.bitrate = self._calculate_bitrate(config.bitrate, config.limit_bitrate)
        self._calculate_band_counts(self.bitrate, self.cutoff_frequency)
        h.calculate_hfr_values()
        self._set_channel_configuration()
        input_sample_count = h.sample_count
        if config.looping:
            h.looping = True
            h.sample_count = min(config.loop_end, config.sample_count)
            h.inserted_samples += next_multiple(config.loop_start, FRAME) - config.loop_start
            self._calculate_loop_info(config.loop_start, config.loop_end)
            input_sample_count = min(next_multiple(h.sample_count, SUB), config.sample_count)
            input_sample_count += SUB * 2
            self.post_samples = input_sample_count - h.sample_count
        self._calculate_header_size()
        total = input_sample_count + h.inserted_samples
        h.frame_count = div_round_up(total, FRAME)
        h.appended_samples = h.frame_count * FRAME - h.inserted_samples - input_sample_count
        self.frame = Frame(h)
        self.channels = self.frame.channels
        self.pcm_buffer = [[0] * FRAME for _ in range(h.channel_count)]
        self.post_audio = [[0] * self.post_samples for _ in range(h.channel_count)]
        self.pending = []
        self.buffer_pre_samples = h.inserted_samples - 128
        self.buffer_position = 0
        self.samples_processed = 0
        self.frames_processed = 0

    # -- set-up helpers
    def _calculate_bitrate(self, bitrate, limit):              # :288-324
        h = self.hca
        pcm_bitrate = h.sample_rate * h.channel_count * 16
        max_bitrate = pcm_bitrate // 4
        min_bitrate = 0
        ratio = {1: 4, 2: 6, 3: 8,
                 4: 10 if h.channel_count == 1 else 12,
                 5: 12 if h.channel_count == 1 else 16}.get(self.quality, 6)
        bitrate = bitrate if bitrate != 0 else pcm_bitrate // ratio
        if limit:
            min_bitrate = min(42666 if h.channel_count == 1 else 32000 * h.channel_count, pcm_bitrate // 6)
        return clamp(bitrate, min_bitrate, max_bitrate)

    def _calculate_band_counts(self, bitrate, cutoff):         # :326-368
        h = self.hca
        h.frame_size = tdiv(tdiv(i32(bitrate * 1024), h.sample_rate), 8)
        pcm_bitrate = h.sample_rate * h.channel_count * 16
        if h.channel_count <= 1 or tdiv(pcm_bitrate, bitrate) <= 6:
            hfr_ratio, cutoff_ratio = 6, 12
        else:
            hfr_ratio, cutoff_ratio = 8, 16
        if bitrate < tdiv(pcm_bitrate, cutoff_ratio):
            cutoff = min(cutoff, tdiv(cutoff_ratio * bitrate, 32 * h.channel_count))
        total_bands = int(round(cutoff * 256.0 / h.sample_rate))
        hfr_start = int(min(total_bands, round((hfr_ratio * bitrate * 128.0) / pcm_bitrate)))
        stereo_start = hfr_start if hfr_ratio == 6 else (hfr_start + 1) // 2
        hfr_bands = total_bands - hfr_start
        per_group = div_round_up(hfr_bands, 8)
        groups = div_round_up(hfr_bands, per_group) if per_group > 0 else 0
        h.total_band_count = total_bands
        h.base_band_count = stereo_start
        h.stereo_band_count = hfr_start - stereo_start
        h.hfr_group_count = groups
        h.bands_per_hfr_group = per_group

    def _set_channel_configuration(self):                      # :370-381
        h = self.hca
        per_track = h.channel_count // h.track_count
        cfg = self.t.default_channel_mapping[per_track]
        if self.t.valid_channel_mappings[per_track - 1][cfg] != 1:
            raise ValueError("Channel mapping is not valid.")
        h.channel_config = cfg

    def _calculate_loop_info(self, loop_start, loop_end):      # :383-398
        h = self.hca
        loop_start += h.inserted_samples
        loop_end += h.inserted_samples
        h.loop_start_frame, h.pre_loop_samples = divmod(loop_start, FRAME)
        h.loop_end_frame = loop_end // FRAME
        h.post_loop_samples = FRAME - loop_end % FRAME
        if h.post_loop_samples == FRAME:
            h.loop_end_frame -= 1
            h.post_loop_samples = 0

    def _calculate_header_size(self):                          # :400-418
        h = self.hca
        h.header_size = next_multiple(96 + h.comment_length, 32)
        if h.looping:
            offset = h.header_size + h.frame_size * h.loop_start_frame
            padding_bytes = next_multiple(offset, 2048) - offset
            padding_frames = tdiv(padding_bytes, h.frame_size)
            h.inserted_samples += padding_frames * FRAME
            h.loop_start_frame += padding_frames
            h.loop_end_frame += padding_frames
            h.header_size += padding_bytes % h.frame_size

    # -- streaming shell :126-269
    @property
    def _buffer_remaining(self):
        return FRAME - self.buffer_position

    def encode(self, pcm):
        """One call of CriHcaEncoder.Encode with a [channels][1024] block; returns the frames it produced."""
        h = self.hca
        if self.frames_processed >= h.frame_count:
            raise RuntimeError("All audio frames have already been output by the encoder")
        out = []
        pos = 0
        if self.buffer_pre_samples > 0:
            while self.buffer_pre_samples > FRAME:             # EncodePreAudio
                self.buffer_position = FRAME
                self._output_frame(out)
                self.buffer_pre_samples -= FRAME
            for j in range(self.buffer_pre_samples):
                for c in range(len(pcm)):
                    self.pcm_buffer[c][j] = pcm[c][0]
            self.buffer_position = self.buffer_pre_samples
            self.buffer_pre_samples = 0
        if h.looping and h.loop_start_sample + self.post_samples >= self.samples_processed \
                and h.loop_start_sample < self.samples_processed + FRAME:
            start = max(h.loop_start_sample - self.samples_processed, 0)      # SaveLoopAudio
            loop_pos = max(self.samples_processed - h.loop_start_sample, 0)
            end = min(h.loop_start_sample - self.samples_processed + self.post_samples, FRAME)
            for c in range(len(pcm)):
                self.post_audio[c][loop_pos:loop_pos + (end - start)] = pcm[c][start:end]
        while FRAME - pos > 0 and h.sample_count > self.samples_processed:    # EncodeMainAudio
            n = min(self._buffer_remaining, FRAME - pos, h.sample_count - self.samples_processed)
            for c in range(len(pcm)):
                self.pcm_buffer[c][self.buffer_position:self.buffer_position + n] = pcm[c][pos:pos + n]
            self.buffer_position += n
            self.samples_processed += n
            pos += n
            self._output_frame(out)
        if h.sample_count == self.samples_processed:           # EncodePostAudio
            post_pos = 0
            while post_pos < self.post_samples:
                n = min(self._buffer_remaining, self.post_samples - post_pos)
                for c in range(len(pcm)):
                    self.pcm_buffer[c][self.buffer_position:self.buffer_position + n] = \
                        self.post_audio[c][post_pos:post_pos + n]
                self.buffer_position += n
                post_pos += n
                self._output_frame(out)
            while self.frames_processed < h.frame_count:
                for c in range(len(pcm)):
                    for k in range(self.buffer_position, FRAME):
                        self.pcm_buffer[c][k] = 0
                self.buffer_position = FRAME
                self._output_frame(out)
        return out

    def _output_frame(self, out):
        if self._buffer_remaining != 0:
            return
        out.append(self.encode_frame(self.pcm_buffer))
        self.buffer_position = 0
        self.frames_processed += 1

    # -- one frame :271-286
    def encode_frame(self, pcm):
        self._pcm_to_float(pcm)
        for ch in self.channels:
            for sf in range(SUBFRAMES):
                ch.spectra[sf] = ch.mdct.run_mdct(ch.pcm_float[sf])
        self._encode_intensity_stereo()
        self._calculate_scale_factors()
        self._scale_spectra()
        self._calculate_hfr_group_averages()
        self._calculate_hfr_scale()
        self._calculate_frame_header_length()
        self._calculate_noise_level()
        self._calculate_evaluation_boundary()
        self._calculate_frame_resolutions()
        self._quantize_spectra()
        return pack_frame(self.frame)

    def _pcm_to_float(self, pcm):                              # :845-858
        for c, ch in enumerate(self.channels):
            for sf in range(SUBFRAMES):
                base = sf * SUB
                ch.pcm_float[sf] = [pcm[c][base + i] * (1.0 / 32768.0) for i in range(SUB)]

    def _encode_intensity_stereo(self):                        # :711-764
        h = self.hca
        if h.stereo_band_count <= 0:
            return
        bounds = self.t.intensity_ratio_bounds
        for c, ch in enumerate(self.channels):
            if ch.type != STEREO_PRIMARY:
                continue
            for sf in range(SUBFRAMES):
                l, r = ch.spectra[sf], self.channels[c + 1].spectra[sf]
                energy_l = energy_r = energy_total = 0.0
                for b in range(h.base_band_count, h.total_band_count):
                    energy_l += abs(l[b])
                    energy_r += abs(r[b])
                    energy_total += abs(l[b] + r[b])
                energy_total *= 2
                energy_lr = energy_r + energy_l
                stored = _fdiv(2 * energy_l, energy_lr)
                ratio = clamp(_fdiv(energy_lr, energy_total), 0.5, math.sqrt(2) / 2)
                quantized = 1
                if energy_r > 0 or energy_l > 0:
                    while quantized < 13 and bounds[quantized] >= stored:
                        quantized += 1
                else:
                    quantized = 0
                    ratio = 1
                self.channels[c + 1].intensity[sf] = quantized
                for b in range(h.base_band_count, h.total_band_count):
                    l[b] = (l[b] + r[b]) * ratio
                    r[b] = 0.0

    def _find_scale_factor(self, value):                       # :691-709
        sf = self.t.dequantizer_scaling
        low, high = 0, 63
        while low < high:
            mid = (low + high) // 2
            if sf[mid] <= value:
                low = mid + 1
            else:
                high = mid
        return low

    def _calculate_scale_factors(self):                        # :673-689
        for ch in self.channels:
            for b in range(ch.coded):
                peak = 0.0
                for sf in range(SUBFRAMES):
                    peak = max(abs(ch.spectra[sf][b]), peak)
                ch.scale_factors[b] = self._find_scale_factor(peak)
            for b in range(ch.coded, SUB):
                ch.scale_factors[b] = 0

    def _scale_spectra(self):                                  # :651-671
        table = self.t.quantizer_scaling
        for ch in self.channels:
            for b in range(ch.coded):
                s = ch.scale_factors[b]
                for sf in range(SUBFRAMES):
                    ch.scaled_spectra[b][sf] = 0.0 if s == 0 else \
                        clamp(ch.spectra[sf][b] * table[s], -0.999999999999, 0.999999999999)

    def _calculate_hfr_group_averages(self):                   # :766-793
        h = self.hca
        if h.hfr_group_count == 0:
            return
        start = h.stereo_band_count + h.base_band_count
        for ch in self.channels:
            if ch.type == STEREO_SECONDARY:
                continue
            band = start
            for group in range(h.hfr_group_count):
                total, count, i = 0.0, 0, 0
                while i < h.bands_per_hfr_group and band < SUB:
                    for sf in range(SUBFRAMES):
                        total += abs(ch.spectra[sf][band])
                    count += SUBFRAMES
                    band += 1
                    i += 1
                ch.hfr_group_average_spectra[group] = _fdiv(total, float(count))

    def _calculate_hfr_scale(self):                            # :795-832
        h = self.hca
        if h.hfr_group_count == 0:
            return
        start = h.stereo_band_count + h.base_band_count
        hfr_bands = min(h.hfr_band_count, h.total_band_count - h.hfr_band_count)
        for ch in self.channels:
            if ch.type == STEREO_SECONDARY:
                continue
            band = 0
            for group in range(h.hfr_group_count):
                total, count, i = 0.0, 0, 0
                while i < h.bands_per_hfr_group and band < hfr_bands:
                    for sf in range(SUBFRAMES):
                        total += abs(ch.scaled_spectra[start - band - 1][sf])
                    count += SUBFRAMES
                    band += 1
                    i += 1
                average = _fdiv(total, float(count))
                if average > 0.0:
                    ch.hfr_group_average_spectra[group] *= min(1.0 / average, math.sqrt(2))
                ch.hfr_scales[group] = self._find_scale_factor(ch.hfr_group_average_spectra[group])

    def _calculate_frame_header_length(self):                  # :599-649
        for ch in self.channels:
            self._optimal_delta_length(ch)
            if ch.type == STEREO_SECONDARY:
                ch.header_length_bits += 32
            elif self.hca.hfr_group_count > 0:
                ch.header_length_bits += 6 * self.hca.hfr_group_count

    @staticmethod
    def _optimal_delta_length(ch):
        if all(ch.scale_factors[i] == 0 for i in range(ch.coded)):
            ch.header_length_bits, ch.scale_factor_delta_bits = 3, 0
            return
        best_bits, best_length = 6, 3 + 6 * ch.coded
        for delta_bits in range(1, 6):
            max_delta = (1 << (delta_bits - 1)) - 1
            length = 3 + 6
            for band in range(1, ch.coded):
                delta = ch.scale_factors[band] - ch.scale_factors[band - 1]
                length += delta_bits + 6 if abs(delta) > max_delta else delta_bits
            if length < best_length:
                best_length, best_bits = length, delta_bits
        ch.header_length_bits, ch.scale_factor_delta_bits = best_length, best_bits

    def _used_bits(self, noise_level, eval_boundary):          # CalculateUsedBits :554-597
        t = self.t
        length = 16 + 16 + 16
        for ch in self.channels:
            length += ch.header_length_bits
            for i in range(ch.coded):
                noise = noise_level - 1 if i < eval_boundary else noise_level
                res = calculate_resolution(ch.scale_factors[i], noise)
                if res >= 8:
                    bits = t.quantized_spectrum_max_bits[res] - 1
                    dead = t.quantizer_dead_zone[res]
                    for v in ch.scaled_spectra[i]:
                        length += bits
                        if abs(v) >= dead:
                            length += 1
                else:
                    inv = t.quantizer_inverse_step_size[res]
                    shift_up = inv + 1
                    shift_down = to_int(inv + 0.5 - 8)
                    row = t.quantize_spectrum_bits[res]
                    for v in ch.scaled_spectra[i]:
                        length += row[to_int(v * inv + shift_up) - shift_down]
        return length

    def _binary_search_level(self, available, low, high):      # :502-523
        top = high
        mid_value = 0
        while low != high:
            mid = (low + high) // 2
            mid_value = self._used_bits(mid, 0)
            if mid_value > available:
                low = mid + 1
            else:
                high = mid
        return -1 if low == top and mid_value > available else low

    def _binary_search_boundary(self, available, noise_level, low, high):     # :525-552
        top = high
        while abs(high - low) > 1:
            mid = (low + high) // 2
            mid_value = self._used_bits(noise_level, mid)
            if available < mid_value:
                high = mid - 1
            else:
                low = mid
        if low == high:
            return low if low < top else -1
        return low if self._used_bits(noise_level, high) > available else high

    def _calculate_noise_level(self):                          # :457-485
        h = self.hca
        highest_band = h.base_band_count + h.stereo_band_count - 1
        available = h.frame_size * 8
        level = self._binary_search_level(available, 0, 255)
        while level < 0:
            highest_band -= 2
            if highest_band < 0:
                raise ValueError("Bitrate is set too low.")
            for ch in self.channels:
                ch.scale_factors[highest_band + 1] = 0
                ch.scale_factors[highest_band + 2] = 0
            self._calculate_frame_header_length()
            level = self._binary_search_level(available, 0, 255)
        self.frame.acceptable_noise_level = level

    def _calculate_evaluation_boundary(self):                  # :487-500
        f = self.frame
        if f.acceptable_noise_level == 0:
            f.evaluation_boundary = 0
            return
        level = self._binary_search_boundary(self.hca.frame_size * 8, f.acceptable_noise_level, 0, 127)
        if level < 0:
            raise NotImplementedError()
        f.evaluation_boundary = level

    def _calculate_frame_resolutions(self):                    # :441-455
        f = self.frame
        for ch in self.channels:
            for i in range(ch.coded):
                noise = f.acceptable_noise_level - 1 if i < f.evaluation_boundary else f.acceptable_noise_level
                ch.resolution[i] = calculate_resolution(ch.scale_factors[i], noise)
            for i in range(ch.coded, SUB):
                ch.resolution[i] = 0

    def _quantize_spectra(self):                               # :420-439
        for ch in self.channels:
            for i in range(ch.coded):
                inv = self.t.quantizer_inverse_step_size[ch.resolution[i]]
                shift_up = inv + 1
                shift_down = to_int(inv + 0.5)
                for sf in range(SUBFRAMES):
                    ch.quantized_spectra[sf][i] = to_int(ch.scaled_spectra[i][sf] * inv + shift_up) - shift_down


def _fdiv(a, b):
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return math.nan
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def encode(pcm, config):
    """CriHcaFormat.EncodeFromPcm16 (Formats/CriHca/CriHcaFormat.cs:34-84): planar pcm [channels][samples]
    -> (HcaInfo, list of frames)."""
    enc = Encoder(config)
    frames = []
    n = len(pcm[0]) if pcm else 0
    block = [[0] * FRAME for _ in pcm]
    i = 0
    while len(frames) < enc.hca.frame_count:
        k = min(n - i * FRAME, FRAME)
        for c, ch in enumerate(pcm):
            if k > 0:
                block[c][0:k] = [int(v) for v in ch[FRAME * i:FRAME * i + k]]
            # the reference reuses one buffer: what a short copy does not overwrite stays (stale tail)
        got = enc.encode(block)
        if not got:
            raise RuntimeError("Encoder returned no audio. This should not happen.")
        frames.extend(got)
        i += 1
    return enc.hca, frames


# ---------------------------------------------------------------- CriHcaPacking.cs (write side)
def pack_frame(frame):                                         # :17-56
    t = Tables.get()
    h = frame.hca
    buf = bytearray(h.frame_size)
    w = BitWriter(buf)
    w.write(0xFFFF, 16)
    w.write(frame.acceptable_noise_level, 9)
    w.write(frame.evaluation_boundary, 7)
    for ch in frame.channels:
        _write_scale_factors(w, ch)
        if ch.type == STEREO_SECONDARY:
            for i in range(SUBFRAMES):
                w.write(ch.intensity[i], 4)
        elif h.hfr_group_count > 0:
            for i in range(h.hfr_group_count):
                w.write(ch.hfr_scales[i], 6)
    for sf in range(SUBFRAMES):
        for ch in frame.channels:
            for i in range(ch.coded):                          # WriteSpectra :238-261
                res = ch.resolution[i]
                q = ch.quantized_spectra[sf][i]
                if res == 0:
                    continue
                if res < 8:
                    w.write(t.quantize_spectrum_value[res][q + 8], t.quantize_spectrum_bits[res][q + 8])
                elif res < 16:
                    w.write(abs(q), t.quantized_spectrum_max_bits[res] - 1)
                    if q != 0:
                        w.write(0 if q > 0 else 1, 1)
    w.align(8)
    for i in range(w.position // 8, h.frame_size - 2):
        buf[i] = 0
    w.position = w.length_bits - 16                            # WriteChecksum :231-236
    w.write(crc16(buf, len(buf) - 2), 16)
    return bytes(buf)


def _write_scale_factors(w, ch):                               # :263-295
    bits = ch.scale_factor_delta_bits
    s = ch.scale_factors
    w.write(bits, 3)
    if bits == 0:
        return
    if bits == 6:
        for i in range(ch.coded):
            w.write(s[i], 6)
        return
    w.write(s[0], 6)
    max_delta = (1 << (bits - 1)) - 1
    escape = (1 << bits) - 1
    for i in range(1, ch.coded):
        delta = s[i] - s[i - 1]
        if abs(delta) > max_delta:
            w.write(escape, bits)
            w.write(s[i], 6)
        else:
            w.write(max_delta + delta, bits)


# ---------------------------------------------------------------- CriHcaPacking.cs (read side) + CriHcaDecoder.cs
def _unpack_frame(frame, r):                                   # UnpackFrame :10-15, header :69-108
    t = Tables.get()
    if r.read(16) != 0xFFFF:
        raise ValueError("Invalid frame header")
    frame.acceptable_noise_level = r.read(9)
    frame.evaluation_boundary = r.read(7)
    ath = frame.ath_curve
    for ch in frame.channels:
        if not _read_scale_factors(ch, r):
            return False
        for i in range(frame.evaluation_boundary):
            ch.resolution[i] = calculate_resolution(ch.scale_factors[i], ath[i] + frame.acceptable_noise_level - 1)
        for i in range(frame.evaluation_boundary, ch.coded):
            ch.resolution[i] = calculate_resolution(ch.scale_factors[i], ath[i] + frame.acceptable_noise_level)
        if ch.type == STEREO_SECONDARY:
            for i in range(SUBFRAMES):
                ch.intensity[i] = r.read(4)
        elif frame.hca.hfr_group_count > 0:
            for i in range(frame.hca.hfr_group_count):
                ch.hfr_scales[i] = r.read(6)
    for sf in range(SUBFRAMES):                                # ReadSpectralCoefficients :143-175
        for ch in frame.channels:
            for s in range(ch.coded):
                res = ch.resolution[s]
                bits = t.quantized_spectrum_max_bits[res]
                code = r.peek(bits)
                if res < 8:
                    bits = t.quantized_spectrum_bits[res][code]
                    ch.quantized_spectra[sf][s] = t.quantized_spectrum_value[res][code]
                else:
                    q = tdiv(code, 2) * (1 - (code % 2 * 2))
                    if q == 0:
                        bits -= 1
                    ch.quantized_spectra[sf][s] = q
                r.position += bits
            for s in range(ch.coded, 0x80):
                ch.spectra[sf][s] = 0.0
    rem = r.remaining                                          # UnpackingWasSuccessful :205-227
    empty = frame.acceptable_noise_level <= 0 and all(c.scale_factor_delta_bits <= 0 for c in frame.channels)
    return 16 <= rem <= 128 or empty or (frame.acceptable_noise_level == 0 and rem >= 16)


def _read_scale_factors(ch, r):                                # :110-127, DeltaDecode :177-203
    ch.scale_factor_delta_bits = r.read(3)
    bits = ch.scale_factor_delta_bits
    if bits == 0:
        for i in range(SUB):
            ch.scale_factors[i] = 0
        return True
    if bits >= 6:
        for i in range(ch.coded):
            ch.scale_factors[i] = r.read(6)
        return True
    ch.scale_factors[0] = r.read(6)
    max_delta = 1 << (bits - 1)
    for i in range(1, ch.coded):
        delta = r.read_offset_binary_positive(bits)
        if delta < max_delta:
            value = ch.scale_factors[i - 1] + delta
            if value < 0 or value > 63:
                return False
            ch.scale_factors[i] = value
        else:
            ch.scale_factors[i] = r.read(6)
    return True


def decode_frame(audio, frame):
    """CriHcaDecoder.DecodeFrame (:72-81): returns [channels][1024] shorts."""
    t = Tables.get()
    h = frame.hca
    _unpack_frame(frame, BitReader(audio))                     # the reference ignores the bool here too
    for ch in frame.channels:                                  # DequantizeFrame :83-100
        for i in range(ch.coded):
            ch.gain[i] = t.dequantizer_scaling[ch.scale_factors[i]] * t.quantizer_step_size[ch.resolution[i]]
    for sf in range(SUBFRAMES):
        for ch in frame.channels:
            for s in range(ch.coded):
                ch.spectra[sf][s] = ch.quantized_spectra[sf][s] * ch.gain[s]
    if h.hfr_group_count != 0:                                 # ReconstructHighFrequency :116-146
        total = min(h.total_band_count, 127)
        start = h.base_band_count + h.stereo_band_count
        hfr_bands = min(h.hfr_band_count, total - h.hfr_band_count)
        for ch in frame.channels:
            if ch.type == STEREO_SECONDARY:
                continue
            band = 0
            for group in range(h.hfr_group_count):
                i = 0
                while i < h.bands_per_hfr_group and band < hfr_bands:
                    hi, lo = start + band, start - band - 1
                    index = ch.hfr_scales[group] - ch.scale_factors[lo] + 64
                    for sf in range(SUBFRAMES):
                        ch.spectra[sf][hi] = t.scale_conversion[index] * ch.spectra[sf][lo]
                    band += 1
                    i += 1
    if h.stereo_band_count > 0:                                # ApplyIntensityStereo :148-166
        for c, ch in enumerate(frame.channels):
            if ch.type != STEREO_PRIMARY:
                continue
            for sf in range(SUBFRAMES):
                l, r = ch.spectra[sf], frame.channels[c + 1].spectra[sf]
                ratio_l = t.intensity_ratio[frame.channels[c + 1].intensity[sf]]
                ratio_r = ratio_l - 2.0
                for b in range(h.base_band_count, h.total_band_count):
                    r[b] = l[b] * ratio_r
                    l[b] *= ratio_l
    pcm = [[0] * FRAME for _ in frame.channels]
    for sf in range(SUBFRAMES):                                # RunImdct :168-177, PcmFloatToShort :179-192
        for c, ch in enumerate(frame.channels):
            samples = ch.mdct.run_imdct(ch.spectra[sf])
            for s in range(SUB):
                pcm[c][sf * SUB + s] = clamp16(to_int(samples[s] * 32768))
    return pcm


def decode(hca, frames):
    """CriHcaDecoder.Decode (:12-31) + CopyPcmToOutput (:33-47)."""
    out = [[0] * hca.sample_count for _ in range(hca.channel_count)]
    frame = Frame(hca)
    for i in range(hca.frame_count):
        block = decode_frame(frames[i], frame)
        current = i * FRAME - hca.inserted_samples
        remaining = min(hca.sample_count - current, hca.sample_count)
        src = clamp(0 - current, 0, FRAME)
        dst = max(current, 0)
        length = min(FRAME - src, remaining)
        if length <= 0:
            continue
        for c in range(hca.channel_count):
            out[c][dst:dst + length] = block[c][src:src + length]
    return out
