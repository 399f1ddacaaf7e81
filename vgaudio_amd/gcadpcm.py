"""Host-side mirror of the reference's GC-ADPCM codec and format classes, routed
through the C ABI (include/vgaudio_hip.h) into the HIP kernels.

Mirrors (reference paths relative to /root/reference/src/VGAudio/):
  GcAdpcmMath            Codecs/GcAdpcm/GcAdpcmMath.cs:7-47
  GcAdpcmParameters      Codecs/GcAdpcm/GcAdpcmParameters.cs:3-7 (+ CodecParameters.cs:5-6)
  GcAdpcmCoefficients    Codecs/GcAdpcm/GcAdpcmCoefficients.cs:9
  GcAdpcmEncoder         Codecs/GcAdpcm/GcAdpcmEncoder.cs:14,48
  GcAdpcmDecoder         Codecs/GcAdpcm/GcAdpcmDecoder.cs:10
  Pcm16Format            Formats/Pcm16/Pcm16Format.cs:14
  GcAdpcmChannel/Format  Formats/GcAdpcm/GcAdpcmFormat.cs:42-74,129-135

Same names, argument meaning and error behaviour (exceptions map 1:1, see
_lib.py).  Every method is one batched call into the GPU library; nothing here
computes codec results on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, i16p, u8p


def _i16(a):
    return a.ctypes.data_as(i16p)


def _u8(a):
    return a.ctypes.data_as(u8p)


def _ptr_array(ptr_type, arrays):
    return (ptr_type * len(arrays))(*[a.ctypes.data_as(ptr_type) for a in arrays])


class GcAdpcmMath:
    BytesPerFrame = 8
    SamplesPerFrame = 14
    NibblesPerFrame = 16

    @staticmethod
    def NibbleCountToSampleCount(n):
        return _lib.lib().vga_gcadpcm_nibble_count_to_sample_count(int(n))

    @staticmethod
    def SampleCountToNibbleCount(n):
        return _lib.lib().vga_gcadpcm_sample_count_to_nibble_count(int(n))

    @staticmethod
    def NibbleToSample(n):
        return _lib.lib().vga_gcadpcm_nibble_to_sample(int(n))

    @staticmethod
    def SampleToNibble(n):
        return _lib.lib().vga_gcadpcm_sample_to_nibble(int(n))

    @staticmethod
    def SampleCountToByteCount(n):
        return _lib.lib().vga_gcadpcm_sample_count_to_byte_count(int(n))

    @staticmethod
    def ByteCountToSampleCount(n):
        return _lib.lib().vga_gcadpcm_byte_count_to_sample_count(int(n))


class GcAdpcmParameters:
    def __init__(self, SampleCount=-1, History1=0, History2=0, Progress=None):
        self.SampleCount = SampleCount
        self.History1 = History1
        self.History2 = History2
        self.Progress = Progress


def _as_channels(pcm, dtype):
    """Accept one 1-D array or a list / 2-D array of equal-length channels."""
    if isinstance(pcm, np.ndarray) and pcm.ndim == 1:
        return [np.ascontiguousarray(pcm, dtype=dtype)], True
    return [np.ascontiguousarray(p, dtype=dtype) for p in pcm], False


class GcAdpcmCoefficients:
    @staticmethod
    def CalculateCoefficients(source):
        """short[] -> short[16]; or a batch of equal-length channels -> [nch,16]."""
        chans, single = _as_channels(source, np.int16)
        nch = len(chans)
        n = len(chans[0]) if nch else 0
        if any(len(c) != n for c in chans):
            raise _lib.ArgumentError("channels of one batch must have equal length")
        coefs = np.zeros((nch, 16), dtype=np.int16)
        check(_lib.lib().vga_gcadpcm_calculate_coefficients_batch(_ptr_array(i16p, chans), nch, n, _i16(coefs)))
        return coefs[0] if single else coefs


class GcAdpcmEncoder:
    @staticmethod
    def Encode(pcm, coefs, config=None):
        """byte[] Encode(short[] pcm, short[] coefs, GcAdpcmParameters config = null); batched when
        pcm is a list / 2-D array (coefs then [nch,16]; History may be per-channel arrays)."""
        config = config or GcAdpcmParameters()
        chans, single = _as_channels(pcm, np.int16)
        nch = len(chans)
        n = len(chans[0]) if nch else 0
        if any(len(c) != n for c in chans):
            raise _lib.ArgumentError("channels of one batch must have equal length")
        sample_count = n if config.SampleCount == -1 else config.SampleCount
        coefs = np.ascontiguousarray(coefs, dtype=np.int16).reshape(nch, 16)
        h1 = np.ascontiguousarray(np.broadcast_to(np.asarray(config.History1, dtype=np.int16), (nch,)))
        h2 = np.ascontiguousarray(np.broadcast_to(np.asarray(config.History2, dtype=np.int16), (nch,)))
        nbytes = GcAdpcmMath.SampleCountToByteCount(max(sample_count, 0))
        outs = [np.zeros(nbytes, dtype=np.uint8) for _ in range(nch)]
        check(_lib.lib().vga_gcadpcm_encode_with_coefs_batch(
            _ptr_array(i16p, chans), nch, n, config.SampleCount, _i16(coefs), _i16(h1), _i16(h2),
            _ptr_array(u8p, outs)))
        if config.Progress is not None:
            config.Progress.ReportAdd(-(-sample_count // 14) * nch)   # one report per batch (boundary)
        return outs[0] if single else outs

    @staticmethod
    def DspEncodeFrame(pcmInOut, sampleCount, adpcmOut, coefsIn):
        """In-place single frame (GcAdpcmEncoder.cs:48-94) via the dsptool-compatible export."""
        buf = np.ascontiguousarray(pcmInOut, dtype=np.int16)
        out = np.zeros(8, dtype=np.uint8)
        co = np.ascontiguousarray(coefsIn, dtype=np.int16)
        _lib.lib().encodeFrame(_i16(buf), _u8(out), _i16(co), 1)
        pcmInOut[:] = buf
        adpcmOut[:8] = out


class GcAdpcmDecoder:
    @staticmethod
    def Decode(adpcm, coefficients, config=None):
        chans, single = _as_channels(adpcm, np.uint8)
        nch = len(chans)
        nb = len(chans[0]) if nch else 0
        if config is None:
            config = GcAdpcmParameters(SampleCount=GcAdpcmMath.ByteCountToSampleCount(nb))
        sample_count = config.SampleCount
        need = GcAdpcmMath.SampleCountToByteCount(max(sample_count, 0))
        if any(len(c) < need for c in chans):
            raise _lib.ArgumentError("adpcm shorter than SampleCount requires")   # IndexOutOfRange in C#
        coefs = np.ascontiguousarray(coefficients, dtype=np.int16).reshape(nch, 16)
        h1 = np.ascontiguousarray(np.broadcast_to(np.asarray(config.History1, dtype=np.int16), (nch,)))
        h2 = np.ascontiguousarray(np.broadcast_to(np.asarray(config.History2, dtype=np.int16), (nch,)))
        outs = [np.zeros(max(sample_count, 0), dtype=np.int16) for _ in range(nch)]
        check(_lib.lib().vga_gcadpcm_decode_batch(_ptr_array(u8p, chans), _i16(coefs), nch, sample_count,
                                                   _i16(h1), _i16(h2), _ptr_array(i16p, outs)))
        return outs[0] if single else outs

    @staticmethod
    def GetPredictorScale(adpcm, sample):
        return int(adpcm[sample // 14 * 8])


class Pcm16Format:
    """Planar PCM carrier: Channels is short[ChannelCount][SampleCount] (Pcm16Format.cs:14)."""

    def __init__(self, channels=None, sampleRate=48000):
        self.Channels = [np.ascontiguousarray(c, dtype=np.int16) for c in (channels if channels is not None else [])]
        self.SampleRate = sampleRate
        n = {len(c) for c in self.Channels}
        if len(n) > 1:
            raise _lib.ArgumentError("All channels must have the same sample count")
        self.SampleCount = n.pop() if n else 0
        self.Looping, self.LoopStart, self.LoopEnd = False, 0, 0

    @property
    def ChannelCount(self):
        return len(self.Channels)

    def WithLoop(self, loop, loopStart=None, loopEnd=None):
        """AudioFormatBaseBuilder.WithLoop (Formats/AudioFormatBaseBuilder.cs:23-58), applied in place."""
        if not loop:
            self.Looping, self.LoopStart, self.LoopEnd = False, 0, 0
            return self
        if loopStart is None and loopEnd is None:
            loopStart, loopEnd = 0, self.SampleCount
        if loopStart < 0 or loopStart > self.SampleCount:
            raise _lib.ArgumentOutOfRangeError("Loop points must be less than the number of samples and non-negative.")
        if loopEnd < 0 or loopEnd > self.SampleCount:
            raise _lib.ArgumentOutOfRangeError("Loop points must be less than the number of samples and non-negative.")
        if loopEnd < loopStart:
            raise _lib.ArgumentOutOfRangeError("The loop end must be greater than the loop start")
        self.Looping, self.LoopStart, self.LoopEnd = True, loopStart, loopEnd
        return self


class GcAdpcmChannel:
    def __init__(self, adpcm, coefs, sampleCount):
        self.Adpcm, self.Coefs, self.SampleCount = adpcm, coefs, sampleCount

    def GetAdpcmAudio(self):
        return self.Adpcm


class GcAdpcmFormat:
    """IAudioFormat for GC-ADPCM; EncodeFromPcm16/ToPcm16 are each ONE batched GPU call
    (the reference's Parallel.For over channels, GcAdpcmFormat.cs:65 / :45)."""

    def __init__(self, channels=None, sampleRate=48000):
        self.Channels = list(channels) if channels is not None else []
        self.SampleRate = sampleRate

    @property
    def ChannelCount(self):
        return len(self.Channels)

    @property
    def SampleCount(self):
        return self.Channels[0].SampleCount if self.Channels else 0

    def EncodeFromPcm16(self, pcm16, config=None):
        nch, n = pcm16.ChannelCount, pcm16.SampleCount
        if config is not None and config.Progress is not None:
            config.Progress.SetTotal(-(-n // 14) * nch)
        if config is not None and config.SampleCount != -1:
            # GcAdpcmEncoder.cs:17 honours the override; coefficients still come from the whole channel
            coefs = GcAdpcmCoefficients.CalculateCoefficients(pcm16.Channels) if nch else np.zeros((0, 16), np.int16)
            adpcm = GcAdpcmEncoder.Encode(pcm16.Channels, coefs, config) if nch else []
        else:
            coefs = np.zeros((nch, 16), dtype=np.int16)
            nbytes = GcAdpcmMath.SampleCountToByteCount(n)
            adpcm = [np.zeros(nbytes, dtype=np.uint8) for _ in range(nch)]
            h1 = config.History1 if config else 0
            h2 = config.History2 if config else 0
            check(_lib.lib().vga_gcadpcm_encode_batch(_ptr_array(i16p, pcm16.Channels), nch, n, h1, h2, _i16(coefs),
                                                       _ptr_array(u8p, adpcm)))
            if config is not None and config.Progress is not None:
                config.Progress.ReportAdd(-(-n // 14) * nch)
        chans = [GcAdpcmChannel(adpcm[i], coefs[i].copy(), n) for i in range(nch)]
        return GcAdpcmFormat(chans, pcm16.SampleRate)

    def ToPcm16(self):
        if not self.Channels:
            return Pcm16Format([], self.SampleRate)
        n = self.SampleCount
        pcm = GcAdpcmDecoder.Decode([c.Adpcm for c in self.Channels], np.stack([c.Coefs for c in self.Channels]),
                                    GcAdpcmParameters(SampleCount=n))
        return Pcm16Format(pcm, self.SampleRate)

    def BuildSeekTable(self, entryCount, samplesPerEntry, bigEndian=True):
        """GcAdpcmFormat.BuildSeekTable (GcAdpcmFormat.cs:99-113) on top of ToPcm16():
        per channel {hist1, hist2} at every samplesPerEntry (GcAdpcmSeekTable.cs:25-38), interleaved by 2."""
        pcm = self.ToPcm16().Channels
        tables = []
        for p in pcm:
            entries = -(-len(p) // samplesPerEntry)
            t = np.zeros(entries * 2, dtype=np.int16)
            for i in range(1, entries):
                t[2 * i] = p[i * samplesPerEntry - 1]
                t[2 * i + 1] = p[i * samplesPerEntry - 2]
            tables.append(t)
        entries = len(tables[0]) // 2
        inter = np.zeros(entries * 2 * len(tables), dtype=np.int16)
        for i in range(entries):
            for c, t in enumerate(tables):
                inter[(i * len(tables) + c) * 2:(i * len(tables) + c) * 2 + 2] = t[2 * i:2 * i + 2]
        out = np.zeros(entryCount * 2 * len(tables), dtype=np.int16)
        m = min(len(out), len(inter))
        out[:m] = inter[:m]
        return out.astype(">i2" if bigEndian else "<i2").tobytes()
