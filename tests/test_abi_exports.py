"""The C-ABI library loads on a CPU-only box and exports every symbol that
include/*.h declare (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    inc = os.path.join(ROOT, "include")
    text = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"\btypedef\b[^;{]*;", "", text)          # function-pointer types are not symbols
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(names))


def test_header_symbols_exported():
    from vgaudio_amd import _lib
    names = _declared_symbols()
    assert "vga_gcadpcm_encode_batch" in names and "encodeFrame" in names
    L = ctypes.CDLL(_lib.SO_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"not exported: {missing}"
    # and the ctypes table covers the header
    assert not [n for n in names if n not in _lib.SIGNATURES], "ctypes SIGNATURES missing entries"


def _declared_arg_counts():
    inc = os.path.join(ROOT, "include")
    text = "".join(open(os.path.join(inc, f)).read() for f in sorted(os.listdir(inc)) if f.endswith(".h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//.*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"\btypedef\b[^;{]*;", "", text)
    out = {}
    for name, args in re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(([^;{()]*)\)\s*;", text):
        args = args.strip()
        out[name] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_ctypes_signatures_have_the_headers_argument_counts():
    """ctypes tolerates extra or missing arguments of cdecl functions silently: hold the table to the header"""
    from vgaudio_amd import _lib
    counts = _declared_arg_counts()
    wrong = {n: (len(_lib.SIGNATURES[n][1]), c) for n, c in counts.items()
             if n in _lib.SIGNATURES and len(_lib.SIGNATURES[n][1]) != c}
    assert not wrong, f"(ctypes, header) argument counts differ: {wrong}"


def test_progress_callback_is_per_thread_host_state():
    """vga_set_progress_callback without a GPU: installing and removing a callback is host state only"""
    from vgaudio_amd import _lib
    L = _lib.lib()
    calls = []
    fn = _lib.PROGRESS_FN(lambda user, done, total: calls.append((done, total)))
    import ctypes as C
    assert L.vga_set_progress_callback(C.cast(fn, C.c_void_p), None) == 0
    assert L.vga_set_progress_callback(None, None) == 0
    assert calls == []


def test_device_list_is_host_state_and_validated():
    """vga_set_devices without a GPU: an empty list is always fine, a device that does not exist is refused loudly"""
    import ctypes as C
    from vgaudio_amd import _lib
    L = _lib.lib()
    assert L.vga_set_devices(None, 0) == 0 and L.vga_get_devices(None, 0) == 0
    bad = (C.c_int * 2)(0, 99)
    assert L.vga_set_devices(bad, 2) != 0
    assert L.vga_get_devices(None, 0) == 0


def test_test_hooks_are_not_in_the_drop_in_header():
    text = open(os.path.join(ROOT, "include", "vgaudio_hip.h")).read()
    assert "debug" not in text.lower() and "testing" not in text.lower()


def test_host_side_size_math_needs_no_gpu():
    from vgaudio_amd.gcadpcm import GcAdpcmMath
    # Tests/Formats/GcAdpcm/GcAdpcmHelpersTests.cs:8-79
    assert GcAdpcmMath.NibbleToSample(100010) == 87508
    assert GcAdpcmMath.SampleToNibble(87508) == 100010
    assert GcAdpcmMath.NibbleCountToSampleCount(100000) == 87500
    assert GcAdpcmMath.SampleCountToNibbleCount(87500) == 100000
    assert [GcAdpcmMath.SampleCountToByteCount(n) for n in (0, 1, 2, 3, 13, 14, 15, 87500)] == \
        [0, 2, 2, 3, 8, 8, 10, 50000]


def test_fails_loudly_without_device():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import vgaudio_amd
    from vgaudio_amd.gcadpcm import GcAdpcmCoefficients
    with pytest.raises(vgaudio_amd.DeviceError):
        GcAdpcmCoefficients.CalculateCoefficients(np.zeros(28, dtype=np.int16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vgaudio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in src and "liboracle" not in src and "oracle/" not in src, f


def test_hca_stream_encode_mirror_rejects_short_arguments_before_the_library_sees_them():
    """CriHcaEncoder.Encode (vgaudio_amd/crihca.py): the C side reads ChannelCount row pointers and writes FrameSize bytes;
    the reference would throw IndexOutOfRange -- the mirror raises ArgumentError, on a box without a GPU too (the checks
    come before the stream is opened)."""
    import numpy as np
    from vgaudio_amd import _lib
    from vgaudio_amd.crihca import CriHcaEncoder, CriHcaParameters
    p = CriHcaParameters()
    p.ChannelCount, p.SampleRate, p.SampleCount = 2, 48000, 4096
    enc = CriHcaEncoder.InitializeNew(p)
    block = [np.zeros(1024, dtype=np.int16) for _ in range(2)]
    good = np.zeros(enc.FrameSize, dtype=np.uint8)
    for pcm, out in ((block[:1], good),                                             # a row short
                     (block, np.zeros(enc.FrameSize - 1, dtype=np.uint8)),           # a byte short
                     (block, np.zeros(enc.FrameSize, dtype=np.int16)),               # wrong element type
                     (block, np.zeros(2 * enc.FrameSize, dtype=np.uint8)[::2]),      # not contiguous
                     ([block[0], block[1][:1000]], good)):                            # a short row
        with pytest.raises(_lib.ArgumentError):
            enc.Encode(pcm, out)
