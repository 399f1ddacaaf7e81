"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see oracle/oracle.h).  The product package `vgaudio_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _declare(_LIB)
        _declare_hca(_LIB)
    return _LIB


class AdxParams(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("highpass_frequency", C.c_int), ("frame_size", C.c_int),
                ("version", C.c_int), ("history", C.c_int16), ("padding", C.c_int), ("type", C.c_int),
                ("filter", C.c_int)]


def _declare(L):
    i16p, u8p, i = C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.c_int
    for name in ("nibble_count_to_sample_count", "sample_count_to_nibble_count", "nibble_to_sample",
                 "sample_to_nibble", "sample_count_to_byte_count", "byte_count_to_sample_count"):
        f = getattr(L, "vgo_gc_" + name)
        f.argtypes, f.restype = [i], i
    L.vgo_gc_calculate_coefficients.argtypes = [i16p, i, i16p]
    L.vgo_gc_calculate_coefficients.restype = None
    L.vgo_gc_encode.argtypes = [i16p, i, i16p, i, C.c_int16, C.c_int16, u8p]
    L.vgo_gc_encode.restype = i
    L.vgo_gc_encode_frame.argtypes = [i16p, i, u8p, i16p]
    L.vgo_gc_encode_frame.restype = None
    L.vgo_gc_decode.argtypes = [u8p, i16p, i, C.c_int16, C.c_int16, i16p]
    L.vgo_gc_decode.restype = None
    L.vgo_gc_encode_batch.argtypes = [i16p, C.c_long, i, i, i16p, u8p, C.c_long, i]
    L.vgo_gc_encode_batch.restype = None
    L.vgo_gc_decode_batch.argtypes = [u8p, C.c_long, i16p, i, i, i16p, C.c_long, i]
    L.vgo_gc_decode_batch.restype = None
    L.vgo_gc_create_seek_table.argtypes = [i16p, i, i, i16p]
    L.vgo_gc_create_seek_table.restype = None
    L.vgo_gc_trip_histogram.argtypes = [C.POINTER(C.c_uint64)]
    L.vgo_gc_trip_histogram.restype = None
    L.vgo_gc_last_encode_hit_nontermination.argtypes = []
    L.vgo_gc_last_encode_hit_nontermination.restype = i
    ap = C.POINTER(AdxParams)
    L.vgo_adx_default_params.argtypes = [ap]
    L.vgo_adx_calculate_coefficients.argtypes = [i, i, i16p]
    L.vgo_adx_encoded_size.argtypes = [i, ap]
    L.vgo_adx_encoded_size.restype = i
    L.vgo_adx_encode.argtypes = [i16p, i, ap, u8p]
    L.vgo_adx_encode.restype = None
    L.vgo_adx_decode.argtypes = [u8p, i, ap, i16p]
    L.vgo_adx_decode.restype = None
    for name in ("nibble_count_to_sample_count", "sample_count_to_nibble_count", "sample_count_to_byte_count"):
        f = getattr(L, "vgo_adx_" + name)
        f.argtypes, f.restype = [i, i], i
    L.vgo_adx_encode_batch.argtypes = [i16p, C.c_long, i, i, ap, u8p, C.c_long, i16p, i]
    L.vgo_adx_encode_batch.restype = None
    L.vgo_adx_decode_batch.argtypes = [u8p, C.c_long, i, i, ap, i16p, C.c_long, i]
    L.vgo_adx_decode_batch.restype = None


class HcaInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "channel_count", "sample_rate", "sample_count", "frame_count", "inserted_samples", "appended_samples",
        "header_size", "frame_size", "min_resolution", "max_resolution", "track_count", "channel_config",
        "total_band_count", "base_band_count", "stereo_band_count", "hfr_band_count", "bands_per_hfr_group",
        "hfr_group_count", "looping", "loop_start_frame", "loop_end_frame", "pre_loop_samples", "post_loop_samples",
        "use_ath_curve", "comment_length")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class HcaParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("quality", "bitrate", "limit_bitrate", "channel_count", "sample_rate",
                                       "sample_count", "looping", "loop_start", "loop_end")]


def _declare_hca(L):
    i16p, u8p, i, dp = C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    hp, pp = C.POINTER(HcaInfo), C.POINTER(HcaParams)
    L.vgo_hca_encoder_init.argtypes = [pp, hp, ip, ip]
    L.vgo_hca_encode.argtypes = [i16p, C.c_long, pp, hp, u8p]
    L.vgo_hca_decode.argtypes = [hp, u8p, i16p, C.c_long]
    L.vgo_hca_encode_batch.argtypes = [i16p, C.c_long, C.c_long, i, pp, u8p, C.c_long, i]
    L.vgo_hca_decode_batch.argtypes = [hp, u8p, C.c_long, i, i16p, C.c_long, C.c_long, i]
    L.vgo_hca_table.argtypes = [C.c_char_p, dp, i]
    L.vgo_crc16.argtypes = [u8p, i]
    L.vgo_crc16.restype = C.c_uint16
    L.vgo_bitwriter_write.argtypes = [u8p, i, i, i, i]
    L.vgo_mdct_run.argtypes = [dp, i, dp, i]
    L.vgo_mdct_run.restype = None
    L.vgo_hca_debug_last_frame.argtypes = [i16p, C.c_long, pp, i, ip, ip, ip, ip, ip, dp]


def _i16(a):
    return a.ctypes.data_as(C.POINTER(C.c_int16))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


# ---------------- GC-ADPCM ----------------
def synth_generate(nch, n, first_channel=0, threads=1, out=None):
    """vgaudio_amd.synth.generate in C (oracle/synth_oracle.c): int16 [nch, n]"""
    L = lib()
    L.vgo_synth_generate.argtypes = [C.POINTER(C.c_int16), C.c_long, C.c_int, C.c_int, C.c_int, C.c_int]
    L.vgo_synth_generate.restype = None
    if out is None:
        out = np.empty((nch, n), dtype=np.int16)
    if nch == 0 or n == 0:
        return out
    assert out.dtype == np.int16 and out.shape[0] >= nch and out.shape[1] >= n and out.strides[1] == 2
    L.vgo_synth_generate(_i16(out), out.strides[0] // 2, nch, n, first_channel, threads)
    return out


def gc_sample_count_to_byte_count(n):
    return lib().vgo_gc_sample_count_to_byte_count(int(n))


def gc_sample_to_nibble(n):
    return lib().vgo_gc_sample_to_nibble(int(n))


def gc_sample_count_to_nibble_count(n):
    return lib().vgo_gc_sample_count_to_nibble_count(int(n))


def gc_calculate_coefficients(pcm):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    coefs = np.zeros(16, dtype=np.int16)
    lib().vgo_gc_calculate_coefficients(_i16(pcm), len(pcm), _i16(coefs))
    return coefs


def gc_encode(pcm, coefs, sample_count=-1, hist1=0, hist2=0):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    n = len(pcm) if sample_count == -1 else sample_count
    out = np.zeros(gc_sample_count_to_byte_count(n), dtype=np.uint8)
    rc = lib().vgo_gc_encode(_i16(pcm), len(pcm), _i16(coefs), sample_count, hist1, hist2, _u8(out))
    if rc != 0:
        raise ValueError("sample_count exceeds pcm length")
    return out


def gc_encode_frame(pcm16, coefs, sample_count=14):
    buf = np.ascontiguousarray(pcm16, dtype=np.int16).copy()
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    out = np.zeros(8, dtype=np.uint8)
    lib().vgo_gc_encode_frame(_i16(buf), sample_count, _u8(out), _i16(coefs))
    return out, buf


def gc_decode(adpcm, coefs, sample_count, hist1=0, hist2=0):
    adpcm = np.ascontiguousarray(adpcm, dtype=np.uint8)
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    out = np.zeros(sample_count, dtype=np.int16)
    lib().vgo_gc_decode(_u8(adpcm), _i16(coefs), sample_count, hist1, hist2, _i16(out))
    return out


def gc_last_encode_hit_nontermination():
    return bool(lib().vgo_gc_last_encode_hit_nontermination())


def gc_trip_histogram():
    h = (C.c_uint64 * 16)()
    lib().vgo_gc_trip_histogram(h)
    return np.array(list(h), dtype=np.uint64)


def gc_encode_batch(pcm2d, threads=1):
    """pcm2d: [nch, n] int16 -> (coefs [nch,16], adpcm [nch, bytes])"""
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    nch, n = pcm2d.shape
    nb = gc_sample_count_to_byte_count(n)
    coefs = np.zeros((nch, 16), dtype=np.int16)
    out = np.zeros((nch, nb), dtype=np.uint8)
    lib().vgo_gc_encode_batch(_i16(pcm2d), n, nch, n, _i16(coefs), _u8(out), nb, threads)
    return coefs, out


def gc_decode_batch(adpcm2d, coefs, sample_count, threads=1):
    adpcm2d = np.ascontiguousarray(adpcm2d, dtype=np.uint8)
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    nch, nb = adpcm2d.shape
    out = np.zeros((nch, sample_count), dtype=np.int16)
    lib().vgo_gc_decode_batch(_u8(adpcm2d), nb, _i16(coefs), nch, sample_count, _i16(out), sample_count, threads)
    return out


def gc_create_seek_table(pcm, samples_per_entry):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    entries = -(-len(pcm) // samples_per_entry)
    out = np.zeros(entries * 2, dtype=np.int16)
    lib().vgo_gc_create_seek_table(_i16(pcm), len(pcm), samples_per_entry, _i16(out))
    return out


class GcChannelParams(C.Structure):
    """vgo_gc_channel_params"""
    _fields_ = [(n, C.c_int) for n in ("sample_count", "looping", "loop_start", "loop_end", "loop_alignment_multiple",
                                       "samples_per_seek_table_entry")]


class GcChannelLayout(C.Structure):
    """vgo_gc_channel_layout"""
    _fields_ = [(n, C.c_int) for n in ("alignment_needed", "loop_start_aligned", "sample_count_aligned",
                                       "seek_table_entries")]


def gc_channel_params(sample_count, looping=False, loop_start=0, loop_end=0, alignment=0, samples_per_entry=0):
    if not looping:
        loop_start = loop_end = 0                  # GcAdpcmChannelBuilder.WithLoop(false) (:113-119)
    return GcChannelParams(sample_count, int(looping), loop_start, loop_end, alignment, samples_per_entry)


def gc_channel_layout(p):
    L = GcChannelLayout()
    lib().vgo_gc_channel_layout_for.argtypes = [C.POINTER(GcChannelParams), C.POINTER(GcChannelLayout)]
    lib().vgo_gc_channel_layout_for(C.byref(p), C.byref(L))
    return L


def gc_alignment(multiple, loop_start, loop_end, adpcm, coefs):
    """GcAdpcmAlignment ctor -> (rc, layout, adpcm_aligned, pcm_aligned)"""
    adpcm = np.ascontiguousarray(adpcm, dtype=np.uint8)
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    L = gc_channel_layout(GcChannelParams(loop_end, 1, loop_start, loop_end, multiple, 0))
    out = np.zeros(max(gc_sample_count_to_byte_count(L.sample_count_aligned), 1), dtype=np.uint8)
    pcm = np.zeros(max(L.sample_count_aligned, 1), dtype=np.int16)
    f = lib().vgo_gc_alignment
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(GcChannelLayout),
                  C.POINTER(C.c_uint8), C.POINTER(C.c_int16)]
    rc = f(multiple, loop_start, loop_end, _u8(adpcm), _i16(coefs), C.byref(L), _u8(out), _i16(pcm))
    return rc, L, out[:gc_sample_count_to_byte_count(L.sample_count_aligned)], pcm[:L.sample_count_aligned]


def gc_loop_context(adpcm, pcm, loop_start):
    adpcm = np.ascontiguousarray(adpcm, dtype=np.uint8)
    out = np.zeros(3, dtype=np.int16)
    f = lib().vgo_gc_loop_context
    f.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.c_int, C.POINTER(C.c_int16)]
    f.restype = None
    if pcm is None:
        f(_u8(adpcm), None, loop_start, _i16(out))
    else:
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        f(_u8(adpcm), _i16(pcm), loop_start, _i16(out))
    return out


def gc_build_channel(adpcm, coefs, p):
    """GcAdpcmChannel(builder) for a fresh channel -> (rc, layout, adpcm, pcm, seek_table, loop_context)"""
    adpcm = np.ascontiguousarray(adpcm, dtype=np.uint8)
    coefs = np.ascontiguousarray(coefs, dtype=np.int16)
    L = gc_channel_layout(p)
    a = np.zeros(max(gc_sample_count_to_byte_count(L.sample_count_aligned), 1), dtype=np.uint8)
    pcm = np.zeros(max(L.sample_count_aligned, 1), dtype=np.int16)
    seek = np.zeros(max(L.seek_table_entries * 2, 1), dtype=np.int16)
    ctx = np.zeros(3, dtype=np.int16)
    f = lib().vgo_gc_build_channel
    f.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(GcChannelParams), C.POINTER(GcChannelLayout),
                  C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.POINTER(C.c_int16)]
    rc = f(_u8(adpcm), _i16(coefs), C.byref(p), C.byref(L), _u8(a), _i16(pcm), _i16(seek), _i16(ctx))
    return (rc, L, a[:gc_sample_count_to_byte_count(L.sample_count_aligned)], pcm[:L.sample_count_aligned],
            seek[:L.seek_table_entries * 2], ctx)




# ---------------- DSP container ----------------
class DspParams(C.Structure):
    """vgo_dsp_params"""
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "sample_count", "looping", "loop_start", "loop_end",
                                       "samples_per_interleave", "loop_point_alignment", "trim_file")]


class DspLayout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sample_count", "loop_start", "loop_end", "start_addr", "end_addr", "cur_addr",
                                       "bytes_per_interleave", "frames_per_interleave", "audio_data_size", "file_size")]


class DspHeader(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sample_count", "nibble_count", "sample_rate", "looping", "format", "start_addr",
                                       "end_addr", "cur_addr", "channel_count", "frames_per_interleave")]


def dsp_params(sample_rate, sample_count, looping=False, loop_start=0, loop_end=0, samples_per_interleave=0x3800,
               loop_point_alignment=1, trim_file=True):
    return DspParams(sample_rate, sample_count, int(looping), loop_start, loop_end, samples_per_interleave,
                     loop_point_alignment, int(trim_file))


def dsp_layout(p, nch):
    L = DspLayout()
    f = lib().vgo_dsp_layout_for
    f.argtypes = [C.POINTER(DspParams), C.c_int, C.POINTER(DspLayout)]
    rc = f(C.byref(p), nch, C.byref(L))
    return rc, L


def interleave(inputs, size, output_size=-1):
    """InterleaveExtensions.Interleave(byte[][]) -> uint8 array of output_size * count bytes."""
    chans = [np.ascontiguousarray(a, dtype=np.uint8) for a in inputs]
    n = len(chans)
    osz = len(chans[0]) if output_size == -1 else output_size
    out = np.zeros(osz * n, dtype=np.uint8)
    ptrs = (C.POINTER(C.c_uint8) * n)(*[_u8(a) for a in chans])
    f = lib().vgo_interleave
    f.restype = None
    f.argtypes = [C.POINTER(C.POINTER(C.c_uint8)), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    f(ptrs, n, len(chans[0]), size, output_size, _u8(out))
    return out


def deinterleave(data, size, count, output_size=-1):
    """DeInterleave(byte[]) -> (rc, [count arrays of output_size bytes])"""
    data = np.ascontiguousarray(data, dtype=np.uint8)
    osz = len(data) // count if output_size == -1 else output_size
    outs = [np.zeros(max(osz, 1), dtype=np.uint8) for _ in range(count)]
    ptrs = (C.POINTER(C.c_uint8) * count)(*[_u8(a) for a in outs])
    f = lib().vgo_deinterleave
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint8))]
    rc = f(_u8(data), len(data), size, count, output_size, ptrs)
    return rc, [o[:osz] for o in outs]


def dsp_write(adpcm, coefs, p, gain=None, start_context=None, loop_context=None):
    """DspWriter -> (rc, file bytes).  adpcm: list of equally long uint8 arrays (GetAdpcmAudio per channel)."""
    nch = len(adpcm)
    chans = [np.ascontiguousarray(a, dtype=np.uint8) for a in adpcm]
    coefs = np.ascontiguousarray(coefs, dtype=np.int16).reshape(nch, 16)
    gain = np.zeros(nch, np.int16) if gain is None else np.ascontiguousarray(gain, dtype=np.int16)
    sc = np.zeros((nch, 3), np.int16) if start_context is None else np.ascontiguousarray(start_context, dtype=np.int16)
    lc = np.zeros((nch, 3), np.int16) if loop_context is None else np.ascontiguousarray(loop_context, dtype=np.int16)
    rc, L = dsp_layout(p, nch)
    if rc:
        return rc, None
    out = np.zeros(L.file_size, dtype=np.uint8)
    ptrs = (C.POINTER(C.c_uint8) * nch)(*[_u8(a) for a in chans])
    f = lib().vgo_dsp_write
    f.argtypes = [C.POINTER(C.POINTER(C.c_uint8)), C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                  C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.c_int, C.POINTER(DspParams), C.POINTER(C.c_uint8)]
    rc = f(ptrs, len(chans[0]), _i16(coefs), _i16(gain), _i16(sc), _i16(lc), nch, C.byref(p), _u8(out))
    return rc, out


def dsp_read(file_bytes):
    """DspReader -> (rc, header, coefs[nch,16], gain, start ctx, loop ctx, [adpcm per channel])"""
    data = np.ascontiguousarray(np.frombuffer(bytes(file_bytes), dtype=np.uint8))
    f = lib().vgo_dsp_read
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(DspHeader), C.POINTER(C.c_int16), C.POINTER(C.c_int16),
                  C.POINTER(C.c_int16), C.POINTER(C.c_int16), C.POINTER(C.POINTER(C.c_uint8))]
    h = DspHeader()
    rc = f(_u8(data), len(data), C.byref(h), None, None, None, None, None)
    if rc:
        return rc, h, None, None, None, None, None
    nch = h.channel_count
    coefs = np.zeros((nch, 16), np.int16); gain = np.zeros(nch, np.int16)
    sc = np.zeros((nch, 3), np.int16); lc = np.zeros((nch, 3), np.int16)
    nb = gc_sample_count_to_byte_count(h.sample_count)
    chans = [np.zeros(max(nb, 1), np.uint8) for _ in range(nch)]
    ptrs = (C.POINTER(C.c_uint8) * nch)(*[_u8(a) for a in chans])
    rc = f(_u8(data), len(data), C.byref(h), _i16(coefs), _i16(gain), _i16(sc), _i16(lc), ptrs)
    return rc, h, coefs, gain, sc, lc, [c[:nb] for c in chans]


# ---------------- ADX / HCA containers ----------------
ADXFILE_PARAM_FIELDS = ("sample_rate", "sample_count", "looping", "loop_start", "loop_end", "alignment_samples", "frame_size",
                        "version", "type", "highpass_frequency", "encryption_type", "trim_file")


class AdxFileParams(C.Structure):
    """vgo_adxfile_params"""
    _fields_ = [(n, C.c_int) for n in ADXFILE_PARAM_FIELDS]


class AdxFileLayout(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sample_count", "frame_count", "base_header_size", "alignment_bytes", "header_size",
                                       "audio_offset", "audio_size", "footer_offset", "footer_size", "loop_start_offset",
                                       "loop_end_offset", "file_size")]


class AdxFileHeader(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("header_size", "type", "frame_size", "bit_depth", "channel_count", "sample_rate",
                                       "sample_count", "highpass_frequency", "version", "revision", "inserted_samples",
                                       "loop_count", "looping", "loop_type", "loop_start_sample", "loop_start_byte",
                                       "loop_end_sample", "loop_end_byte")]


def adxfile_params(sample_rate, sample_count, looping=False, loop_start=0, loop_end=0, alignment_samples=0, frame_size=18,
                   version=4, type=3, highpass_frequency=500, encryption_type=0, trim_file=True):
    return AdxFileParams(sample_rate, sample_count, int(looping), loop_start, loop_end, alignment_samples, frame_size, version,
                         type, highpass_frequency, encryption_type, int(trim_file))


def adxfile_layout(p, nch):
    L = AdxFileLayout()
    f = lib().vgo_adxfile_layout_for
    f.argtypes = [C.POINTER(AdxFileParams), C.c_int, C.POINTER(AdxFileLayout)]
    return f(C.byref(p), nch, C.byref(L)), L


def adxfile_write(audio, history, p):
    """AdxWriter -> (rc, file bytes).  audio: equally long uint8 arrays (CriAdxChannel.Audio); history per channel."""
    nch = len(audio)
    chans = [np.ascontiguousarray(a, dtype=np.uint8) for a in audio]
    hist = np.ascontiguousarray(history, dtype=np.int16)
    rc, L = adxfile_layout(p, nch)
    if rc:
        return rc, None
    out = np.zeros(L.file_size, dtype=np.uint8)
    ptrs = (C.POINTER(C.c_uint8) * nch)(*[_u8(a) for a in chans])
    f = lib().vgo_adxfile_write
    f.argtypes = [C.POINTER(C.POINTER(C.c_uint8)), C.c_int, C.POINTER(C.c_int16), C.c_int, C.POINTER(AdxFileParams),
                  C.POINTER(C.c_uint8)]
    rc = f(ptrs, len(chans[0]), _i16(hist), nch, C.byref(p), _u8(out))
    return rc, out


def adxfile_read(file_bytes):
    """AdxReader -> (rc, header, history, [audio per channel])"""
    data = np.ascontiguousarray(np.frombuffer(bytes(file_bytes), dtype=np.uint8))
    f = lib().vgo_adxfile_read
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(AdxFileHeader), C.POINTER(C.c_int16),
                  C.POINTER(C.POINTER(C.c_uint8))]
    h = AdxFileHeader()
    rc = f(_u8(data), len(data), C.byref(h), None, None)
    if rc:
        return rc, h, None, None
    nch = h.channel_count
    spf = (h.frame_size - 2) * 2
    nb = h.frame_size * (-(-h.sample_count // spf))
    hist = np.zeros(nch, np.int16)
    chans = [np.zeros(max(nb, 1), np.uint8) for _ in range(nch)]
    ptrs = (C.POINTER(C.c_uint8) * nch)(*[_u8(a) for a in chans])
    rc = f(_u8(data), len(data), C.byref(h), _i16(hist), ptrs)
    return rc, h, hist, [c[:nb] for c in chans]


def hcafile_size(info):
    f = lib().vgo_hcafile_size
    f.argtypes = [C.POINTER(HcaInfo)]
    return f(C.byref(info))


def hcafile_write(info, frames, comment=None, volume=1.0, encryption_type=0, encrypted_ids=False):
    """HcaWriter -> (rc, file bytes).  frames: frame_count * frame_size bytes."""
    fr = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1)
    out = np.zeros(hcafile_size(info), dtype=np.uint8)
    f = lib().vgo_hcafile_write
    f.argtypes = [C.POINTER(HcaInfo), C.POINTER(C.c_uint8), C.c_char_p, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    rc = f(C.byref(info), _u8(fr), None if comment is None else comment.encode("utf-8"), volume, encryption_type,
           int(encrypted_ids), _u8(out))
    return rc, out


def hcafile_read(file_bytes):
    """HcaReader header -> (rc, info, volume, encryption_type, comment, version)"""
    data = np.ascontiguousarray(np.frombuffer(bytes(file_bytes), dtype=np.uint8))
    f = lib().vgo_hcafile_read
    f.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(HcaInfo), C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_char_p,
                  C.POINTER(C.c_int)]
    h = HcaInfo()
    vol, enc, ver = C.c_float(), C.c_int(), C.c_int()
    buf = C.create_string_buffer(max(len(data), 16))
    rc = f(_u8(data), len(data), C.byref(h), C.byref(vol), C.byref(enc), buf, C.byref(ver))
    return rc, h, vol.value, enc.value, buf.value.decode("utf-8", "replace"), ver.value


# ---------------- WAVE ----------------
class WaveInfo(C.Structure):
    _fields_ = ([(n, C.c_int) for n in ("channel_count", "sample_rate", "bits_per_sample", "sample_count", "sample_count_declared",
                                        "looping", "loop_start", "loop_end", "smpl_loop_count", "smpl_loop_start", "smpl_loop_end")]
                + [("data_offset", C.c_long), ("data_size", C.c_int), ("data_size_declared", C.c_int)])


class WaveParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "sample_count", "looping", "loop_start", "loop_end")]


def wave_parse(file_bytes):
    data = np.ascontiguousarray(np.frombuffer(bytes(file_bytes), dtype=np.uint8))
    w = WaveInfo()
    f = lib().vgo_wave_parse
    f.argtypes = [C.POINTER(C.c_uint8), C.c_long, C.POINTER(WaveInfo)]
    return f(_u8(data), len(data), C.byref(w)), w


def wave_read(file_bytes):
    """WaveReader -> (rc, info, [pcm per channel])"""
    data = np.ascontiguousarray(np.frombuffer(bytes(file_bytes), dtype=np.uint8))
    rc, w = wave_parse(file_bytes)
    if rc:
        return rc, w, None
    chans = [np.zeros(max(w.sample_count, 1), np.int16) for _ in range(w.channel_count)]
    ptrs = (C.POINTER(C.c_int16) * w.channel_count)(*[_i16(a) for a in chans])
    f = lib().vgo_wave_read_pcm16
    f.argtypes = [C.POINTER(C.c_uint8), C.c_long, C.POINTER(WaveInfo), C.POINTER(C.POINTER(C.c_int16))]
    rc = f(_u8(data), len(data), C.byref(w), ptrs)
    return rc, w, [c[:w.sample_count] for c in chans]


def wave_write(pcm, sample_rate, looping=False, loop_start=0, loop_end=0):
    """WaveWriter (16-bit) -> (rc, file bytes)"""
    chans = [np.ascontiguousarray(a, dtype=np.int16) for a in pcm]
    nch = len(chans)
    p = WaveParams(sample_rate, len(chans[0]), int(looping), loop_start, loop_end)
    fs = lib().vgo_wave_file_size
    fs.restype, fs.argtypes = C.c_long, [C.POINTER(WaveParams), C.c_int]
    out = np.zeros(fs(C.byref(p), nch), dtype=np.uint8)
    ptrs = (C.POINTER(C.c_int16) * nch)(*[_i16(a) for a in chans])
    f = lib().vgo_wave_write_pcm16
    f.argtypes = [C.POINTER(C.POINTER(C.c_int16)), C.c_int, C.POINTER(WaveParams), C.POINTER(C.c_uint8)]
    rc = f(ptrs, nch, C.byref(p), _u8(out))
    return rc, out


# ---------------- ADX / HCA encryption ----------------
class AdxKey(C.Structure):
    _fields_ = [("seed", C.c_int), ("mult", C.c_int), ("inc", C.c_int)]


def adx_key_from_code(code):
    k = AdxKey()
    f = lib().vgo_adx_key_from_code
    f.restype, f.argtypes = None, [C.c_uint64, C.POINTER(AdxKey)]
    f(code, C.byref(k))
    return k


def adx_key_from_string(s):
    k = AdxKey()
    f = lib().vgo_adx_key_from_string
    f.restype, f.argtypes = None, [C.c_char_p, C.POINTER(AdxKey)]
    f(s.encode("ascii"), C.byref(k))
    return k


def adx_key_code(k):
    f = lib().vgo_adx_key_code
    f.restype, f.argtypes = C.c_uint64, [C.POINTER(AdxKey)]
    return f(C.byref(k))


def adx_crypt(audio, key, encryption_type, frame_size=18):
    """CriAdxEncryption.EncryptDecrypt on copies -> list of uint8 arrays"""
    out = [np.array(a, dtype=np.uint8, copy=True) for a in audio]
    f = lib().vgo_adx_crypt_channel
    f.restype, f.argtypes = None, [C.POINTER(C.c_uint8), C.c_int, C.POINTER(AdxKey), C.c_int, C.c_int, C.c_int, C.c_int]
    for i, a in enumerate(out):
        f(_u8(a), len(a), C.byref(key), encryption_type, frame_size, i, len(out))
    return out


def adx_test_key(audio, key, encryption_type, frame_size=18):
    chans = [np.ascontiguousarray(a, dtype=np.uint8) for a in audio]
    ptrs = (C.POINTER(C.c_uint8) * len(chans))(*[_u8(a) for a in chans])
    f = lib().vgo_adx_test_key
    f.argtypes = [C.POINTER(C.POINTER(C.c_uint8)), C.c_int, C.c_int, C.POINTER(AdxKey), C.c_int, C.c_int]
    return f(ptrs, len(chans[0]), len(chans), C.byref(key), encryption_type, frame_size)


def hca_key_tables(key_type, key_code=0):
    dec, enc = np.zeros(256, np.uint8), np.zeros(256, np.uint8)
    f = lib().vgo_hca_key_tables
    f.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    return f(key_type, key_code, _u8(dec), _u8(enc)), dec, enc


def hca_crypt(frames, frame_size, table):
    out = np.array(frames, dtype=np.uint8, copy=True).reshape(-1)
    f = lib().vgo_hca_crypt
    f.restype, f.argtypes = None, [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    t = np.ascontiguousarray(table, dtype=np.uint8)
    f(_u8(out), len(out) // frame_size, frame_size, _u8(t))
    return out


def hca_find_key(info, frames, tables):
    """CriHcaEncryption.FindKey over `tables` ([nkeys, 256] decryption tables): index, -1 (none) or -3 (bad sync word)"""
    frames = np.ascontiguousarray(frames, dtype=np.uint8).reshape(-1)
    tables = np.ascontiguousarray(tables, dtype=np.uint8).reshape(-1, 256)
    f = lib().vgo_hca_find_key
    f.restype, f.argtypes = C.c_int, [C.POINTER(HcaInfo), C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint8), C.c_int]
    return f(C.byref(info), _u8(frames), len(frames) // info.frame_size, _u8(tables), tables.shape[0])


def adx_guess_default_candidates(encryption_type):
    m, n = np.zeros(0x2000, np.int32), np.zeros(0x2000, np.int32)
    nm, nn = C.c_int(), C.c_int()
    f = lib().vgo_adx_default_candidates
    f.restype, f.argtypes = C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
    assert f(encryption_type, m.ctypes.data, C.byref(nm), n.ctypes.data, C.byref(nn)) == 0
    return m[:nm.value].copy(), n[:nn.value].copy()


def adx_guess_keys(scales, start_frame, encryption_type, mults=None, incs=None, max_keys=4096):
    """GuessAdx's search for one file's scales -> sorted list of (seed, mult, inc); None when more than max_keys"""
    scales = np.ascontiguousarray(scales, dtype=np.uint16)
    out = (AdxKey * max_keys)()
    f = lib().vgo_adx_guess_keys
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(AdxKey), C.c_int]
    if mults is None or incs is None:
        n = f(scales.ctypes.data, len(scales), start_frame, encryption_type, None, 0, None, 0, out, max_keys)
    else:
        m = np.ascontiguousarray(mults, dtype=np.int32)
        i = np.ascontiguousarray(incs, dtype=np.int32)
        n = f(scales.ctypes.data, len(scales), start_frame, encryption_type, m.ctypes.data, len(m), i.ctypes.data, len(i), out, max_keys)
    if n < 0:
        return None
    return [(out[k].seed, out[k].mult, out[k].inc) for k in range(n)]


def hca_byte_position_counts(frames2d, frame_size, positions=30):
    frames2d = np.ascontiguousarray(frames2d, dtype=np.uint8)
    ns, pitch = frames2d.shape
    counts = np.zeros((positions, 256), dtype=np.uint32)
    f = lib().vgo_hca_byte_position_counts
    f.restype = None
    f.argtypes = [C.POINTER(C.c_uint8), C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f(_u8(frames2d), pitch, ns, pitch // frame_size, frame_size, positions, counts.ctypes.data)
    return counts


# ---------------- ADX ----------------
def adx_params(**kw):
    p = AdxParams()
    lib().vgo_adx_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def adx_calculate_coefficients(highpass, sample_rate):
    c = np.zeros(2, dtype=np.int16)
    lib().vgo_adx_calculate_coefficients(highpass, sample_rate, _i16(c))
    return c


def adx_encode(pcm, params):
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = np.zeros(lib().vgo_adx_encoded_size(len(pcm), C.byref(params)), dtype=np.uint8)
    lib().vgo_adx_encode(_i16(pcm), len(pcm), C.byref(params), _u8(out))
    return out


def adx_decode(adpcm, sample_count, params):
    adpcm = np.ascontiguousarray(adpcm, dtype=np.uint8)
    out = np.zeros(sample_count, dtype=np.int16)
    lib().vgo_adx_decode(_u8(adpcm), sample_count, C.byref(params), _i16(out))
    return out


def adx_encode_batch(pcm2d, params, threads=1):
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    nch, n = pcm2d.shape
    nb = lib().vgo_adx_encoded_size(n, C.byref(params))
    out = np.zeros((nch, nb), dtype=np.uint8)
    hist = np.zeros(nch, dtype=np.int16)
    lib().vgo_adx_encode_batch(_i16(pcm2d), n, nch, n, C.byref(params), _u8(out), nb, _i16(hist), threads)
    return out, hist


def adx_decode_batch(adpcm2d, sample_count, params, threads=1):
    adpcm2d = np.ascontiguousarray(adpcm2d, dtype=np.uint8)
    nch, nb = adpcm2d.shape
    out = np.zeros((nch, sample_count), dtype=np.int16)
    lib().vgo_adx_decode_batch(_u8(adpcm2d), nb, nch, sample_count, C.byref(params), _i16(out), sample_count, threads)
    return out


# ---------------- HCA ----------------
HCA_QUALITY = {"NotSet": 0, "Highest": 1, "High": 2, "Middle": 3, "Low": 4, "Lowest": 5}


def hca_params(channel_count, sample_count, sample_rate=48000, quality="High", bitrate=0, limit_bitrate=False,
               looping=False, loop_start=0, loop_end=0):
    return HcaParams(HCA_QUALITY[quality] if isinstance(quality, str) else quality, bitrate, int(limit_bitrate),
                     channel_count, sample_rate, sample_count, int(looping), loop_start, loop_end)


def hca_init(params):
    info = HcaInfo()
    post, pre = C.c_int(), C.c_int()
    rc = lib().vgo_hca_encoder_init(C.byref(params), C.byref(info), C.byref(post), C.byref(pre))
    return rc, info


def hca_encode(pcm2d, params):
    """pcm2d [nch, n] -> (rc, HcaInfo, frames [frame_count, frame_size] uint8)"""
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    rc, info = hca_init(params)
    if rc:
        return rc, info, None
    frames = np.zeros((info.frame_count, info.frame_size), dtype=np.uint8)
    rc = lib().vgo_hca_encode(_i16(pcm2d), pcm2d.shape[1], C.byref(params), C.byref(info), _u8(frames))
    return rc, info, frames


def hca_decode(info, frames):
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    out = np.zeros((info.channel_count, max(info.sample_count, 1)), dtype=np.int16)
    rc = lib().vgo_hca_decode(C.byref(info), _u8(frames), _i16(out), out.shape[1])
    return rc, out[:, :info.sample_count]


def hca_encode_batch(pcm3d, params, threads=1):
    """pcm3d [nstreams, nch, n] -> (rc, info, frames [nstreams, frame_count*frame_size])"""
    pcm3d = np.ascontiguousarray(pcm3d, dtype=np.int16)
    ns, nch, n = pcm3d.shape
    rc, info = hca_init(params)
    if rc:
        return rc, info, None
    fb = info.frame_count * info.frame_size
    frames = np.zeros((ns, fb), dtype=np.uint8)
    rc = lib().vgo_hca_encode_batch(_i16(pcm3d), nch * n, n, ns, C.byref(params), _u8(frames), fb, threads)
    return rc, info, frames


def hca_decode_batch(info, frames2d, threads=1):
    frames2d = np.ascontiguousarray(frames2d, dtype=np.uint8)
    ns = frames2d.shape[0]
    n = max(info.sample_count, 1)
    out = np.zeros((ns, info.channel_count, n), dtype=np.int16)
    rc = lib().vgo_hca_decode_batch(C.byref(info), _u8(frames2d), frames2d.shape[1], ns, _i16(out),
                                    info.channel_count * n, n, threads)
    return rc, out[:, :, :info.sample_count]


def hca_table(name):
    buf = np.zeros(256, dtype=np.float64)
    n = lib().vgo_hca_table(name.encode(), buf.ctypes.data_as(C.POINTER(C.c_double)), 256)
    return buf[:n].copy()


def crc16(data):
    data = np.ascontiguousarray(data, dtype=np.uint8)
    return int(lib().vgo_crc16(_u8(data), len(data)))


def mdct_run(blocks2d, inverse=False):
    x = np.ascontiguousarray(blocks2d, dtype=np.float64)
    out = np.zeros_like(x)
    dp = C.POINTER(C.c_double)
    lib().vgo_mdct_run(x.ctypes.data_as(dp), x.shape[0], out.ctypes.data_as(dp), int(inverse))
    return out


def hca_debug_last_frame(pcm2d, params, frames):
    pcm2d = np.ascontiguousarray(pcm2d, dtype=np.int16)
    nch = pcm2d.shape[0]
    nl, eb = C.c_int(), C.c_int()
    sf = np.zeros((nch, 128), np.int32)
    res = np.zeros((nch, 128), np.int32)
    q = np.zeros((nch, 8, 128), np.int32)
    sp = np.zeros((nch, 8, 128), np.float64)
    ip = C.POINTER(C.c_int)
    rc = lib().vgo_hca_debug_last_frame(_i16(pcm2d), pcm2d.shape[1], C.byref(params), frames, C.byref(nl), C.byref(eb),
                                        sf.ctypes.data_as(ip), res.ctypes.data_as(ip), q.ctypes.data_as(ip),
                                        sp.ctypes.data_as(C.POINTER(C.c_double)))
    return rc, nl.value, eb.value, sf, res, q, sp
