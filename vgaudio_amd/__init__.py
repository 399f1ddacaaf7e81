"""vgaudio_amd -- MI355X-native batch audio-codec engine behind VGAudio's
IAudioFormat seam (GC-ADPCM, CRI ADX, CRI HCA).  The compute path is the HIP
library libvgaudio_hip.so (C ABI: include/vgaudio_hip.h); this package is the
thin host-side mirror of the reference's codec/format classes used by the
tests and the benchmark.  No CPU fallback exists.
"""
import os as _os

# The host pipeline wants more hardware queues than the HIP runtime's default of 4 (see ask_for_hardware_queues in
# csrc/capi_gcadpcm.hip); the runtime reads the variable when it initialises, i.e. at the first torch.cuda / HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import _lib  # noqa: E402,F401
from ._lib import (ArgumentError, ArgumentOutOfRangeError, DeviceError, InvalidDataError,  # noqa: F401
                   InvalidOperationError, VgaError)
