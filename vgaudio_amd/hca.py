"""HCA container writer -- host-side mirror of VGAudio/Containers/Hca/HcaWriter.cs and
HcaConfiguration.cs over vga_hca_write / vga_hca_file_header."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, u8p
from .crihca import CriHcaEncryption, CriHcaFormat, CriHcaParameters, CriHcaQuality
from .gcadpcm import Pcm16Format


class HcaConfiguration:
    """Containers/Hca/HcaConfiguration.cs:5-11."""

    def __init__(self, EncryptionKey=None, Quality=CriHcaQuality.NotSet, Bitrate=0, LimitBitrate=False, Progress=None):
        self.EncryptionKey, self.Quality, self.Bitrate, self.LimitBitrate, self.Progress = (
            EncryptionKey, Quality, Bitrate, LimitBitrate, Progress)


class HcaWriter:
    """AudioWriter<HcaWriter, HcaConfiguration>: GetFile(audio, configuration)."""

    def __init__(self, configuration=None):
        self.Configuration = configuration or HcaConfiguration()

    def _setup(self, audio):                                 # SetupWriter (:23-46)
        cfg = self.Configuration
        if isinstance(audio, Pcm16Format):
            enc = CriHcaParameters(Progress=cfg.Progress, Bitrate=cfg.Bitrate, LimitBitrate=cfg.LimitBitrate)
            if cfg.Quality != CriHcaQuality.NotSet:
                enc.Quality = cfg.Quality
            audio = CriHcaFormat().EncodeFromPcm16(audio, enc)
        if not isinstance(audio, CriHcaFormat):
            raise _lib.ArgumentError("HcaWriter takes a CriHcaFormat or a Pcm16Format")
        if cfg.EncryptionKey is not None:                    # :39-45 -- like the reference, this encrypts the format's frames in place
            audio.AudioData = np.ascontiguousarray(audio.AudioData, dtype=np.uint8)
            CriHcaEncryption.Crypt(audio.Hca, audio.AudioData, cfg.EncryptionKey, False)
            audio.Hca.EncryptionType = cfg.EncryptionKey.KeyType
        return audio

    @staticmethod
    def _comment(hca):
        return None if hca.Comment is None else hca.Comment.encode("utf-8")

    def GetHeader(self, fmt):
        hca = fmt.Hca
        out = np.zeros(hca.HeaderSize, dtype=np.uint8)
        check(_lib.lib().vga_hca_file_header(C.byref(hca.c), self._comment(hca), float(hca.Volume), int(hca.EncryptionType),
                                             int(self.Configuration.EncryptionKey is not None), out.ctypes.data_as(u8p)))
        return out.tobytes()

    def GetFile(self, audio, configuration=None):
        if configuration is not None:
            self.Configuration = configuration
        fmt = self._setup(audio)
        hca = fmt.Hca
        size = _lib.lib().vga_hca_file_size(C.byref(hca.c))
        if size < 0:
            check(size)
        frames = np.ascontiguousarray(fmt.AudioData, dtype=np.uint8).reshape(-1)
        if len(frames) != hca.FrameCount * hca.FrameSize:
            raise _lib.ArgumentError("AudioData does not hold FrameCount frames of FrameSize bytes")
        out = np.zeros(size, dtype=np.uint8)
        check(_lib.lib().vga_hca_write(C.byref(hca.c), frames.ctypes.data_as(u8p), self._comment(hca), float(hca.Volume),
                                       int(hca.EncryptionType), int(self.Configuration.EncryptionKey is not None),
                                       out.ctypes.data_as(u8p)))
        return out.tobytes()
