"""ADX container writer -- host-side mirror of VGAudio/Containers/Adx/AdxWriter.cs and
AdxConfiguration.cs.  The image is assembled on the GPU (vga_adx_write), encryption included (vga_adx_crypt); there is no CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, u8p
from .criadx import CriAdxEncryption, CriAdxFormat, CriAdxParameters, CriAdxType
from .gcadpcm import Pcm16Format, _i16, _ptr_array


class AdxConfiguration:
    """Containers/Adx/AdxConfiguration.cs:5-13 + Configuration.TrimFile."""

    def __init__(self, Version=4, EncryptionType=0, EncryptionKey=None, FrameSize=18, Filter=2, Type=CriAdxType.Linear,
                 TrimFile=True, Progress=None):
        self.Version, self.EncryptionType, self.EncryptionKey = Version, EncryptionType, EncryptionKey
        self.FrameSize, self.Filter, self.Type, self.TrimFile, self.Progress = FrameSize, Filter, Type, TrimFile, Progress


class AdxWriter:
    """AudioWriter<AdxWriter, AdxConfiguration>: GetFile(audio, configuration)."""

    def __init__(self, configuration=None):
        self.Configuration = configuration or AdxConfiguration()

    def _setup(self, audio):                                 # SetupWriter (:39-56)
        cfg = self.Configuration
        if isinstance(audio, Pcm16Format):                   # AudioData.GetFormat<CriAdxFormat>(encodingConfig)
            audio = CriAdxFormat().EncodeFromPcm16(audio, CriAdxParameters(
                Progress=cfg.Progress, Version=cfg.Version, FrameSize=cfg.FrameSize, Filter=cfg.Filter, Type=cfg.Type))
        if not isinstance(audio, CriAdxFormat):
            raise _lib.ArgumentError("AdxWriter takes a CriAdxFormat or a Pcm16Format")
        return audio

    def _params(self, fmt):
        cfg = self.Configuration
        return _lib.AdxFileParamsC(fmt.SampleRate, fmt.SampleCount, int(fmt.Looping), fmt.LoopStart, fmt.LoopEnd,
                                   fmt.AlignmentSamples, fmt.FrameSize, fmt.Version, fmt.Type, fmt.HighpassFrequency,
                                   cfg.EncryptionType, int(bool(cfg.TrimFile)))

    def Layout(self, fmt):
        L = _lib.AdxFileLayoutC()
        p = self._params(fmt)
        check(_lib.lib().vga_adx_file_layout_for(C.byref(p), fmt.ChannelCount, C.byref(L)))
        return L

    def GetFile(self, audio, configuration=None):
        if configuration is not None:
            self.Configuration = configuration
        fmt = self._setup(audio)
        L = self.Layout(fmt)
        p = self._params(fmt)
        src = [np.ascontiguousarray(ch.Audio, dtype=np.uint8) for ch in fmt.Channels]
        cfg = self.Configuration
        if cfg.EncryptionKey is not None:                    # WriteData (:123-128): encrypt copies, not the format's audio
            src = [a.copy() for a in src]
            CriAdxEncryption.EncryptDecrypt(src, cfg.EncryptionKey, cfg.EncryptionType, fmt.FrameSize)
        if any(len(a) != len(src[0]) for a in src):
            raise _lib.ArgumentOutOfRangeError("Inputs must be of equal length")            # Interleave.cs:49-50
        hist = np.array([ch.History for ch in fmt.Channels], dtype=np.int16)
        out = np.zeros(L.file_size, dtype=np.uint8)
        check(_lib.lib().vga_adx_write(_ptr_array(u8p, src), len(src[0]), _i16(hist), fmt.ChannelCount, C.byref(p),
                                       out.ctypes.data_as(u8p)))
        return out.tobytes()
