"""DSP container writer for GC-ADPCM -- the host-side mirror of
VGAudio/Containers/Dsp/DspWriter.cs and DspConfiguration.cs.  The file image is assembled on the GPU
(vga_dsp_write); there is no CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, i16p, u8p
from .gcadpcm import GcAdpcmFormat, _i16, _ptr_array


class DspConfiguration:
    """Containers/Dsp/DspConfiguration.cs + Containers/Configuration.cs (TrimFile)."""

    def __init__(self, SamplesPerInterleave=0x3800, LoopPointAlignment=1, TrimFile=True, RecalculateLoopContext=True):
        self.SamplesPerInterleave = SamplesPerInterleave
        self.LoopPointAlignment = LoopPointAlignment
        self.TrimFile = TrimFile
        self.RecalculateLoopContext = RecalculateLoopContext

    @property
    def SamplesPerInterleave(self):
        return self._samples_per_interleave

    @SamplesPerInterleave.setter
    def SamplesPerInterleave(self, value):           # DspConfiguration.cs:31-45
        if value < 1:
            raise _lib.ArgumentOutOfRangeError("Number of samples per interleave must be positive")
        if value % 14 != 0:
            raise _lib.ArgumentOutOfRangeError("Number of samples per interleave must be divisible by 14")
        self._samples_per_interleave = int(value)


class DspWriter:
    """AudioWriter<DspWriter, DspConfiguration> (Containers/AudioWriter.cs:11-44): GetFile(format, configuration)."""

    def __init__(self, configuration=None):
        self.Configuration = configuration or DspConfiguration()

    def _params(self, fmt):
        c = self.Configuration
        return _lib.DspParamsC(fmt.SampleRate, fmt.SampleCount, int(fmt.Looping), fmt.LoopStart, fmt.LoopEnd,
                               c.SamplesPerInterleave, c.LoopPointAlignment, int(bool(c.TrimFile)))

    def Layout(self, fmt):
        """The header geometry DspWriter.cs:18-36,99-100 derives (sizes only)."""
        L = _lib.DspLayoutC()
        p = self._params(fmt)
        check(_lib.lib().vga_dsp_layout_for(C.byref(p), fmt.ChannelCount, C.byref(L)))
        return L

    def GetFile(self, audio, configuration=None):
        if configuration is not None:
            self.Configuration = configuration
        if not isinstance(audio, GcAdpcmFormat):
            raise _lib.ArgumentError("DspWriter takes a GcAdpcmFormat (encode PCM with GcAdpcmFormat.EncodeFromPcm16 first)")
        fmt = audio
        nch = fmt.ChannelCount
        L = self.Layout(fmt)
        p = self._params(fmt)
        src = [np.ascontiguousarray(ch.GetAdpcmAudio(), dtype=np.uint8) for ch in fmt.Channels]
        if any(len(a) != len(src[0]) for a in src):
            raise _lib.ArgumentOutOfRangeError("Inputs must be of equal length")                 # Interleave.cs:49-50
        coefs = np.ascontiguousarray(np.stack([ch.Coefs for ch in fmt.Channels]), dtype=np.int16).reshape(nch, 16)
        gain = np.array([getattr(ch, "Gain", 0) for ch in fmt.Channels], dtype=np.int16)
        start = np.array([[ch.StartContext.PredScale, ch.StartContext.Hist1, ch.StartContext.Hist2] for ch in fmt.Channels],
                         dtype=np.int16)
        loop = np.array([[ch.LoopContext.PredScale, ch.LoopContext.Hist1, ch.LoopContext.Hist2] for ch in fmt.Channels],
                        dtype=np.int16)
        out = np.zeros(L.file_size, dtype=np.uint8)
        check(_lib.lib().vga_dsp_write(_ptr_array(u8p, src), len(src[0]), _i16(coefs), _i16(gain), _i16(start), _i16(loop),
                                       nch, C.byref(p), out.ctypes.data_as(u8p)))
        return out.tobytes()
