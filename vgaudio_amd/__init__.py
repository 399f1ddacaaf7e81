"""vgaudio_amd -- MI355X-native batch audio-codec engine behind VGAudio's
IAudioFormat seam (GC-ADPCM, CRI ADX, CRI HCA).  The compute path is the HIP
library libvgaudio_hip.so (C ABI: include/vgaudio_hip.h); this package is the
thin host-side mirror of the reference's codec/format classes used by the
tests and the benchmark.  No CPU fallback exists.
"""
from . import _lib  # noqa: F401
from ._lib import (ArgumentError, ArgumentOutOfRangeError, DeviceError, InvalidDataError,  # noqa: F401
                   InvalidOperationError, VgaError)
