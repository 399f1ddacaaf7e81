"""ctypes binding of libvgaudio_hip.so (the C ABI in include/vgaudio_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
visible, calls fail loudly (VgaError / OSError).
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VGAUDIO_HIP_LIBRARY: load a different build of the same ABI (kernel experiments, tools/variants/)
SO_PATH = os.environ.get("VGAUDIO_HIP_LIBRARY") or os.path.join(_HERE, "libvgaudio_hip.so")

VGA_OK = 0
VGA_ERR_ARGUMENT = -1
VGA_ERR_OUT_OF_RANGE = -2
VGA_ERR_INVALID_DATA = -3
VGA_ERR_INVALID_OP = -4
VGA_ERR_DEVICE = -5


class VgaError(RuntimeError):
    """Base class; subclasses mirror the .NET exception the reference throws."""
    code = None


class ArgumentError(VgaError, ValueError):           # ArgumentException
    code = VGA_ERR_ARGUMENT


class ArgumentOutOfRangeError(ArgumentError):        # ArgumentOutOfRangeException
    code = VGA_ERR_OUT_OF_RANGE


class InvalidDataError(VgaError):                    # InvalidDataException
    code = VGA_ERR_INVALID_DATA


class InvalidOperationError(VgaError):               # InvalidOperationException
    code = VGA_ERR_INVALID_OP


class DeviceError(VgaError):                         # HIP failure / no device
    code = VGA_ERR_DEVICE


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)     # vga_progress_fn


class reporting:
    """`with reporting(progress, frames_per_unit):` around a *_batch call: IProgressReport.ReportAdd once per chunk of
    channels / streams the call finishes (vga_set_progress_callback), where the reference reports once per frame
    (GcAdpcmEncoder.cs:42, CriAdxCodec.cs:101, CriHcaFormat.cs:71,79).  progress None: nothing is installed."""

    def __init__(self, progress, per_unit):
        self.progress, self.per_unit, self.last, self.fn = progress, per_unit, 0, None

    def _report(self, _user, done, _total):
        self.progress.ReportAdd((done - self.last) * self.per_unit)
        self.last = done

    def __enter__(self):
        if self.progress is not None:
            self.fn = PROGRESS_FN(self._report)
            lib().vga_set_progress_callback(C.cast(self.fn, C.c_void_p), None)
        return self

    def __exit__(self, *exc):
        if self.fn is not None:
            lib().vga_set_progress_callback(None, None)
        return False


_EXC = {e.code: e for e in (ArgumentError, ArgumentOutOfRangeError, InvalidDataError, InvalidOperationError,
                            DeviceError)}

_lib = None

i16p = C.POINTER(C.c_int16)
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
i16pp = C.POINTER(i16p)
u8pp = C.POINTER(u8p)
vp = C.c_void_p
i64 = C.c_int64
ci = C.c_int


class ADPCMINFO(C.Structure):
    _pack_ = 1
    _fields_ = [("coef", C.c_int16 * 16), ("gain", C.c_uint16), ("pred_scale", C.c_uint16), ("yn1", C.c_int16),
                ("yn2", C.c_int16), ("loop_pred_scale", C.c_uint16), ("loop_yn1", C.c_int16), ("loop_yn2", C.c_int16)]


# name -> (restype, argtypes); every symbol include/vgaudio_hip.h declares
SIGNATURES = {
    "vga_last_error": (C.c_char_p, []),
    "vga_device_count": (ci, []),
    "vga_set_device": (ci, [ci]),
    "vga_version": (C.c_char_p, []),
    "vga_gcadpcm_nibble_count_to_sample_count": (ci, [ci]),
    "vga_gcadpcm_sample_count_to_nibble_count": (ci, [ci]),
    "vga_gcadpcm_nibble_to_sample": (ci, [ci]),
    "vga_gcadpcm_sample_to_nibble": (ci, [ci]),
    "vga_gcadpcm_sample_count_to_byte_count": (ci, [ci]),
    "vga_gcadpcm_byte_count_to_sample_count": (ci, [ci]),
    "vga_gcadpcm_encode_batch": (ci, [i16pp, ci, ci, C.c_int16, C.c_int16, i16p, u8pp]),
    "vga_gcadpcm_calculate_coefficients_batch": (ci, [i16pp, ci, ci, i16p]),
    "vga_gcadpcm_encode_with_coefs_batch": (ci, [i16pp, ci, ci, ci, i16p, i16p, i16p, u8pp]),
    "vga_gcadpcm_decode_batch": (ci, [u8pp, i16p, ci, ci, i16p, i16p, i16pp]),
    "vga_gcadpcm_encode_batch_v": (ci, [i16pp, C.POINTER(ci), ci, i16p, i16p, i16p, u8pp]),
    "vga_gcadpcm_calculate_coefficients_batch_v": (ci, [i16pp, C.POINTER(ci), ci, i16p]),
    "vga_gcadpcm_encode_with_coefs_batch_v": (ci, [i16pp, C.POINTER(ci), ci, i16p, i16p, i16p, u8pp]),
    "vga_gcadpcm_decode_batch_v": (ci, [u8pp, i16p, C.POINTER(ci), ci, i16p, i16p, i16pp]),
    "vga_gcadpcm_ragged_create": (ci, [C.POINTER(ci), ci, C.POINTER(vp)]),
    "vga_gcadpcm_ragged_destroy": (None, [vp]),
    "vga_gcadpcm_ragged_channels": (ci, [vp]),
    "vga_gcadpcm_ragged_pcm_samples": (i64, [vp]),
    "vga_gcadpcm_ragged_adpcm_bytes": (i64, [vp]),
    "vga_gcadpcm_ragged_coefs_workspace_bytes": (C.c_size_t, [vp]),
    "vga_gcadpcm_ragged_offsets": (ci, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "vga_gcadpcm_coefs_device_v": (ci, [vp, vp, vp, vp, C.c_size_t, vp]),
    "vga_gcadpcm_encode_device_v": (ci, [vp, vp, vp, vp, vp, vp, vp]),
    "vga_gcadpcm_decode_device_v": (ci, [vp, vp, vp, vp, vp, vp, vp, vp]),
    "encode": (None, [i16p, u8p, C.POINTER(ADPCMINFO), C.c_uint32]),
    "decode": (None, [u8p, i16p, C.POINTER(ADPCMINFO), C.c_uint32]),
    "correlateCoefs": (None, [i16p, C.c_uint32, i16p]),
    "encodeFrame": (None, [i16p, u8p, i16p, C.c_uint8]),
    "vga_gcadpcm_coefs_workspace_bytes": (C.c_size_t, [ci, ci]),
    "vga_gcadpcm_coefs_device": (ci, [vp, i64, ci, ci, vp, vp, C.c_size_t, vp]),
    "vga_gcadpcm_encode_device": (ci, [vp, i64, ci, ci, vp, vp, vp, vp, i64, vp]),
    "vga_gcadpcm_decode_device": (ci, [vp, i64, vp, ci, ci, vp, vp, vp, i64, vp, vp]),
    "vga_synth_pcm16_device": (ci, [vp, i64, ci, ci, ci, vp, vp]),
    "vga_adx_file_layout_for": (ci, [vp, ci, vp]),
    "vga_adx_write": (ci, [u8pp, ci, i16p, ci, vp, u8p]),
    "vga_adx_write_device": (ci, [vp, i64, ci, vp, ci, vp, vp, vp]),
    "vga_hca_file_size": (ci, [vp]),
    "vga_hca_file_header": (ci, [vp, C.c_char_p, C.c_float, ci, ci, u8p]),
    "vga_hca_write": (ci, [vp, u8p, C.c_char_p, C.c_float, ci, ci, u8p]),
    "vga_hca_write_device": (ci, [vp, vp, i64, ci, C.c_char_p, C.c_float, ci, ci, vp, i64, vp]),
    "vga_wave_parse": (ci, [u8p, i64, vp]),
    "vga_wave_read_pcm16": (ci, [u8p, i64, vp, i16pp]),
    "vga_wave_deinterleave_pcm16_device": (ci, [vp, ci, ci, vp, i64, vp]),
    "vga_wave_file_size": (i64, [vp, ci]),
    "vga_wave_write_pcm16": (ci, [i16pp, ci, vp, u8p]),
    "vga_wave_write_pcm16_device": (ci, [vp, i64, ci, vp, vp, vp]),
    "vga_adx_key_from_code": (ci, [C.c_uint64, vp]),
    "vga_adx_key_from_string": (ci, [C.c_char_p, vp]),
    "vga_adx_key_code": (C.c_uint64, [vp]),
    "vga_adx_crypt": (ci, [u8pp, ci, ci, vp, ci, ci]),
    "vga_adx_crypt_device": (ci, [vp, i64, ci, ci, vp, ci, ci, vp]),
    "vga_adx_find_key_device": (ci, [vp, i64, ci, ci, ci, ci, vp, ci, C.POINTER(ci), vp]),
    "vga_hca_key_tables": (ci, [ci, C.c_uint64, u8p, u8p]),
    "vga_hca_crypt": (ci, [u8p, ci, ci, u8p]),
    "vga_hca_find_key": (ci, [vp, u8p, ci, u8p, ci, vp]),
    "vga_hca_find_key_device": (ci, [vp, vp, ci, u8p, ci, vp, vp]),
    "vga_hca_byte_position_counts_device": (ci, [vp, i64, ci, ci, ci, ci, vp, vp]),
    "vga_adx_guess_default_candidates": (ci, [ci, vp, vp, vp, vp]),
    "vga_adx_guess_keys": (ci, [vp, ci, ci, ci, vp, ci, vp, ci, vp, ci, vp]),
    "vga_hca_crypt_device": (ci, [vp, i64, ci, ci, ci, u8p, vp]),
    "vga_release_cached_memory": (None, []),
    "vga_testing_force_open_seams_this_thread": (ci, [ci]),
    "vga_testing_host_pipeline_this_thread": (None, [ci, ci, ci, ci]),
    "vga_testing_gc_encoder_layout_this_thread": (ci, [ci]),
    "vga_testing_gc_coefs_variant_this_thread": (ci, [ci]),
    "vga_testing_gc_encoder_segments_this_thread": (ci, [ci]),
    "vga_testing_gc_encoder_persistent_this_thread": (ci, [ci]),
    "vga_testing_last_pipeline_stats": (ci, [vp, ci]),
    "vga_testing_host_pipeline_tail_this_thread": (None, [ci]),
    "vga_testing_buckets_order_this_thread": (None, [ci]),
    "vga_testing_host_transfer_this_thread": (None, [ci]),
    "vga_testing_host_compute_lanes_this_thread": (None, [ci]),
    "vga_testing_plan_buckets": (ci, [C.POINTER(ci), C.POINTER(ci), ci, ci, C.c_longlong, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci),
                                 C.POINTER(ci), ci]),
    "vga_testing_hca_device_info": (ci, [vp, vp, ci]),
    "vga_hca_stream_create": (ci, [vp, vp, C.POINTER(vp)]),
    "vga_hca_stream_encode": (ci, [vp, vp, u8p, C.POINTER(ci)]),
    "vga_hca_stream_pending_frame_count": (ci, [vp]),
    "vga_hca_stream_get_pending_frame": (ci, [vp, u8p]),
    "vga_hca_stream_frames_processed": (ci, [vp]),
    "vga_hca_stream_frame_size": (ci, [vp]),
    "vga_hca_stream_destroy": (None, [vp]),
    "vga_testing_gc_encode_stats": (ci, [C.POINTER(C.c_ulonglong), ci]),
    "vga_testing_gc_plan_pieces": (ci, [ci, ci, ci, C.c_longlong, ci, C.POINTER(ci)]),
    "vga_set_devices": (ci, [vp, ci]),
    "vga_set_progress_callback": (ci, [vp, vp]),
    "vga_get_devices": (ci, [vp, ci]),
    "vga_testing_hca_frames_per_group_this_thread": (ci, [ci]),
    "vga_dsp_layout_for": (ci, [vp, ci, vp]),
    "vga_dsp_write": (ci, [u8pp, ci, i16p, i16p, i16p, i16p, ci, vp, u8p]),
    "vga_dsp_write_device": (ci, [vp, i64, ci, vp, vp, vp, vp, ci, vp, vp, vp]),
    "vga_gcadpcm_channel_layout_for": (ci, [vp, vp]),
    "vga_gcadpcm_build_channels_batch": (ci, [u8pp, i16p, ci, vp, u8pp, i16pp, i16pp, i16p]),
    "vga_gcadpcm_build_channels_workspace_bytes": (C.c_size_t, [ci, vp]),
    "vga_gcadpcm_build_channels_device": (ci, [vp, i64, vp, ci, vp, vp, i64, vp, i64, vp, i64, vp, vp, C.c_size_t, vp]),
    "vga_adx_default_params": (None, [vp]),
    "vga_adx_calculate_coefficients": (ci, [ci, ci, i16p]),
    "vga_adx_nibble_count_to_sample_count": (ci, [ci, ci]),
    "vga_adx_sample_count_to_nibble_count": (ci, [ci, ci]),
    "vga_adx_sample_count_to_byte_count": (ci, [ci, ci]),
    "vga_adx_encoded_byte_count": (ci, [ci, vp]),
    "vga_adx_encode_batch": (ci, [i16pp, ci, ci, vp, u8pp, i16p]),
    "vga_adx_decode_batch": (ci, [u8pp, ci, ci, ci, vp, i16pp]),
    "vga_adx_encode_batch_v": (ci, [i16pp, C.POINTER(ci), ci, vp, u8pp, i16p]),
    "vga_adx_decode_batch_v": (ci, [u8pp, C.POINTER(ci), ci, C.POINTER(ci), vp, i16pp]),
    "vga_hca_encode_batch_v": (ci, [i16pp, ci, vp, vp, u8pp]),
    "vga_hca_decode_batch_v": (ci, [vp, u8pp, ci, i16pp]),
    "vga_adx_encode_device": (ci, [vp, i64, ci, ci, vp, vp, i64, vp, vp]),
    "vga_adx_decode_device": (ci, [vp, i64, ci, ci, ci, vp, vp, i64, vp, vp]),
    "vga_hca_encoder_initialize": (ci, [vp, vp]),
    "vga_hca_encode_batch": (ci, [i16pp, ci, vp, vp, u8pp]),
    "vga_hca_decode_batch": (ci, [vp, u8pp, ci, i16pp]),
    "vga_hca_decode_workspace_bytes": (C.c_size_t, [vp, ci]),
    "vga_hca_encode_device": (ci, [vp, i64, i64, ci, ci, vp, vp, i64, vp, vp]),
    "vga_hca_decode_device": (ci, [vp, vp, i64, ci, vp, i64, i64, vp, C.c_size_t, vp, vp]),
}


class AdxFileParamsC(C.Structure):
    """vga_adx_file_params"""
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "sample_count", "looping", "loop_start", "loop_end", "alignment_samples",
                                       "frame_size", "version", "type", "highpass_frequency", "encryption_type", "trim_file")]


class AdxFileLayoutC(C.Structure):
    """vga_adx_file_layout"""
    _fields_ = [(n, C.c_int) for n in ("sample_count", "frame_count", "base_header_size", "alignment_bytes", "header_size",
                                       "audio_offset", "audio_size", "footer_offset", "footer_size", "loop_start_offset",
                                       "loop_end_offset", "file_size")]


class WaveInfoC(C.Structure):
    """vga_wave_info"""
    _fields_ = ([(n, C.c_int) for n in ("channel_count", "sample_rate", "bits_per_sample", "sample_count", "sample_count_declared",
                                        "looping", "loop_start", "loop_end")]
                + [("data_offset", C.c_int64), ("data_size", C.c_int), ("data_size_declared", C.c_int)])


class WaveParamsC(C.Structure):
    """vga_wave_params"""
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "sample_count", "looping", "loop_start", "loop_end")]


class AdxKeyC(C.Structure):
    """vga_adx_key"""
    _fields_ = [("seed", C.c_int), ("mult", C.c_int), ("inc", C.c_int)]


class DspParamsC(C.Structure):
    """vga_dsp_params"""
    _fields_ = [(n, C.c_int) for n in ("sample_rate", "sample_count", "looping", "loop_start", "loop_end",
                                       "samples_per_interleave", "loop_point_alignment", "trim_file")]


class DspLayoutC(C.Structure):
    """vga_dsp_layout"""
    _fields_ = [(n, C.c_int) for n in ("sample_count", "loop_start", "loop_end", "start_addr", "end_addr", "cur_addr",
                                       "bytes_per_interleave", "frames_per_interleave", "audio_data_size", "file_size")]


class GcChannelParamsC(C.Structure):
    """vga_gcadpcm_channel_params (include/vgaudio_hip.h)"""
    _fields_ = [(n, C.c_int) for n in ("sample_count", "looping", "loop_start", "loop_end", "loop_alignment_multiple",
                                       "samples_per_seek_table_entry")]


class GcChannelLayoutC(C.Structure):
    """vga_gcadpcm_channel_layout"""
    _fields_ = [(n, C.c_int) for n in ("alignment_needed", "loop_start_aligned", "sample_count_aligned",
                                       "seek_table_entries")]


class HcaInfoC(C.Structure):
    """vga_hca_info (include/vgaudio_hip.h) == HcaInfo."""
    _fields_ = [(n, C.c_int) for n in (
        "channel_count", "sample_rate", "sample_count", "frame_count", "inserted_samples", "appended_samples",
        "header_size", "frame_size", "min_resolution", "max_resolution", "track_count", "channel_config",
        "total_band_count", "base_band_count", "stereo_band_count", "hfr_band_count", "bands_per_hfr_group",
        "hfr_group_count", "looping", "loop_start_frame", "loop_end_frame", "pre_loop_samples", "post_loop_samples",
        "use_ath_curve", "comment_length")]


class HcaParamsC(C.Structure):
    """vga_hca_params == CriHcaParameters."""
    _fields_ = [(n, C.c_int) for n in ("quality", "bitrate", "limit_bitrate", "channel_count", "sample_rate",
                                       "sample_count", "looping", "loop_start", "loop_end")]


class AdxParams(C.Structure):
    """vga_adx_params (include/vgaudio_hip.h) == CriAdxParameters."""
    _fields_ = [("sample_rate", C.c_int), ("highpass_frequency", C.c_int), ("frame_size", C.c_int),
                ("version", C.c_int), ("history", C.c_int16), ("padding", C.c_int), ("type", C.c_int),
                ("filter", C.c_int)]


def _preload_torch_hip_runtime():
    """One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 (SONAME
    libamdhip64.so.7, same as /opt/rocm's).  If this library were loaded first it would bind
    /opt/rocm's copy, torch would then load its own, and the second HSA runtime in the process
    reports "no ROCm-capable device".  Pre-loading torch's copy (when torch is installed) makes
    both resolve to the same runtime regardless of import order.  Without torch (C#/C++ hosts)
    the system ROCm runtime is used."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load the shared library (built in-tree by `python -m vgaudio_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise OSError(f"{SO_PATH} not found: run `python -m vgaudio_amd.build` (hipcc, gfx950). "
                          "There is no CPU fallback.")
        _preload_torch_hip_runtime()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(rc):
    if rc == VGA_OK:
        return
    msg = lib().vga_last_error().decode("utf-8", "replace")
    raise _EXC.get(rc, VgaError)(msg or f"vgaudio_hip error {rc}")
