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
        with _lib.reporting(config.Progress, -(-sample_count // 14)):   # frames per channel (GcAdpcmEncoder.cs:42)
            check(_lib.lib().vga_gcadpcm_encode_with_coefs_batch(
                _ptr_array(i16p, chans), nch, n, config.SampleCount, _i16(coefs), _i16(h1), _i16(h2),
                _ptr_array(u8p, outs)))
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


class GcAdpcmContext:
    """PredScale / Hist1 / Hist2 (Codecs/GcAdpcm/GcAdpcmContext.cs)."""

    def __init__(self, predScale=0, hist1=0, hist2=0):
        self.PredScale, self.Hist1, self.Hist2 = int(predScale), int(hist1), int(hist2)


class GcAdpcmChannel:
    """Formats/GcAdpcm/GcAdpcmChannel.cs.  A channel made by the 3-argument constructor carries only the
    encoded audio; build_channels() (the batched GcAdpcmChannelBuilder.Build) adds what the reference derives:
    aligned audio, decoded PCM, loop context and seek table."""

    def __init__(self, adpcm, coefs, sampleCount):
        self.Adpcm, self.Coefs, self.UnalignedSampleCount = adpcm, coefs, sampleCount
        self.AlignmentNeeded = False
        self._aligned_adpcm = None
        self._aligned_count = sampleCount
        self._pcm = None
        self._seek = None
        self.SamplesPerSeekTableEntry = 0
        self.LoopContext = GcAdpcmContext()
        self.LoopContextStart = 0
        self.StartContext = GcAdpcmContext(adpcm[0] if len(adpcm) else 0, 0, 0)
        self.Gain = 0

    @property
    def SampleCount(self):                       # GcAdpcmChannel.cs:11
        return self._aligned_count if self.AlignmentNeeded else self.UnalignedSampleCount

    def GetAdpcmAudio(self):                     # :63
        return self._aligned_adpcm if self.AlignmentNeeded else self.Adpcm

    def GetPcmAudio(self):                       # :57-60
        if self._pcm is None:
            self._pcm = GcAdpcmDecoder.Decode(self.GetAdpcmAudio(), self.Coefs, GcAdpcmParameters(SampleCount=self.SampleCount))
        return self._pcm

    def GetSeekTable(self):                      # :62
        return self._seek if self._seek is not None else np.zeros(0, dtype=np.int16)


def build_channels(channels, looping=False, loopStart=0, loopEnd=0, alignmentMultiple=0, samplesPerSeekTableEntry=0,
                   keepPcm=False, loopContext=True):
    """GcAdpcmChannelBuilder.Build for a batch of freshly encoded channels that share one loop
    (GcAdpcmFormat.cs:27-40) -- ONE GPU call (vga_gcadpcm_build_channels_batch).  Returns new channels.
    loopContext=False is the bare GcAdpcmAlignment constructor (no GetLoopContext)."""
    if not channels:
        return []
    n = channels[0].UnalignedSampleCount
    if any(c.UnalignedSampleCount != n for c in channels):
        raise _lib.ArgumentError("channels of one build must share the sample count")
    if not looping:
        loopStart = loopEnd = 0                  # GcAdpcmChannelBuilder.WithLoop(false) (:113-119)
    p = _lib.GcChannelParamsC(n, int(bool(looping)), loopStart, loopEnd, alignmentMultiple, samplesPerSeekTableEntry)
    L = _lib.GcChannelLayoutC()
    check(_lib.lib().vga_gcadpcm_channel_layout_for(C.byref(p), C.byref(L)))
    nch = len(channels)
    want_ctx = loopContext and L.loop_start_aligned != 0
    if not (L.alignment_needed or want_ctx or L.seek_table_entries or keepPcm):
        out = [GcAdpcmChannel(c.Adpcm, c.Coefs, n) for c in channels]           # nothing to derive
        for o in out:
            o.SamplesPerSeekTableEntry = samplesPerSeekTableEntry
        return out
    src = [np.ascontiguousarray(c.Adpcm, dtype=np.uint8) for c in channels]
    coefs = np.ascontiguousarray(np.stack([c.Coefs for c in channels]), dtype=np.int16)
    nbytes = GcAdpcmMath.SampleCountToByteCount(L.sample_count_aligned)
    aligned = [np.zeros(nbytes, dtype=np.uint8) for _ in range(nch)] if L.alignment_needed else None
    pcm = [np.zeros(L.sample_count_aligned, dtype=np.int16) for _ in range(nch)]
    seek = [np.zeros(L.seek_table_entries * 2, dtype=np.int16) for _ in range(nch)] if L.seek_table_entries else None
    ctx = np.zeros((nch, 3), dtype=np.int16)
    check(_lib.lib().vga_gcadpcm_build_channels_batch(
        _ptr_array(u8p, src), _i16(coefs), nch, C.byref(p), _ptr_array(u8p, aligned) if aligned else None,
        _ptr_array(i16p, pcm), _ptr_array(i16p, seek) if seek else None, _i16(ctx) if loopContext else None))
    out = []
    for i, c in enumerate(channels):
        o = GcAdpcmChannel(c.Adpcm, c.Coefs, n)
        o.AlignmentNeeded = bool(L.alignment_needed)
        o._aligned_adpcm = aligned[i] if aligned else None
        o._aligned_count = L.sample_count_aligned
        o._pcm = pcm[i]
        o._seek = seek[i] if seek else None
        o.SamplesPerSeekTableEntry = samplesPerSeekTableEntry
        o.LoopContext = GcAdpcmContext(*ctx[i].tolist())
        o.LoopContextStart = L.loop_start_aligned
        out.append(o)
    return out


class GcAdpcmFormat:
    """IAudioFormat for GC-ADPCM; EncodeFromPcm16/ToPcm16 and the channel build of the constructor are each
    ONE batched GPU call (the reference's Parallel.For over channels, GcAdpcmFormat.cs:65 / :45 / :32)."""

    def __init__(self, channels=None, sampleRate=48000, looping=False, loopStart=0, loopEnd=0, alignmentMultiple=0,
                 samplesPerSeekTableEntry=0):
        self.SampleRate = sampleRate
        self.Looping = bool(looping)
        self.UnalignedLoopStart = loopStart if looping else 0
        self.UnalignedLoopEnd = loopEnd if looping else 0
        self.AlignmentMultiple = alignmentMultiple
        self.SamplesPerSeekTableEntry = samplesPerSeekTableEntry
        chans = list(channels) if channels is not None else []
        # GcAdpcmFormat(GcAdpcmFormatBuilder) rebuilds every channel with the format's loop (:27-40)
        self.Channels = build_channels(chans, self.Looping, self.UnalignedLoopStart, self.UnalignedLoopEnd, alignmentMultiple,
                                       samplesPerSeekTableEntry)

    @property
    def ChannelCount(self):
        return len(self.Channels)

    @property
    def UnalignedSampleCount(self):
        return self.Channels[0].UnalignedSampleCount if self.Channels else 0

    @property
    def _alignment_samples(self):                # GcAdpcmFormat.cs:19
        m = self.AlignmentMultiple
        s = self.UnalignedLoopStart
        return (s + m - s % m if m > 0 and s % m else s) - s

    @property
    def LoopStart(self):                         # :20
        return self.UnalignedLoopStart + self._alignment_samples

    @property
    def LoopEnd(self):                           # :21
        return self.UnalignedLoopEnd + self._alignment_samples

    @property
    def SampleCount(self):                       # :22
        return self.UnalignedSampleCount if self._alignment_samples == 0 else self.LoopEnd

    def _clone(self, **kw):
        a = dict(sampleRate=self.SampleRate, looping=self.Looping, loopStart=self.UnalignedLoopStart,
                 loopEnd=self.UnalignedLoopEnd, alignmentMultiple=self.AlignmentMultiple,
                 samplesPerSeekTableEntry=self.SamplesPerSeekTableEntry)
        a.update(kw)
        base = [GcAdpcmChannel(c.Adpcm, c.Coefs, c.UnalignedSampleCount) for c in self.Channels]
        return GcAdpcmFormat(base, **a)

    def WithLoop(self, loop, loopStart=None, loopEnd=None):          # AudioFormatBase.WithLoop (:61-63)
        if not loop:
            return self._clone(looping=False, loopStart=0, loopEnd=0)
        n = self.UnalignedSampleCount
        if loopStart is None and loopEnd is None:
            loopStart, loopEnd = 0, n
        if loopStart < 0 or loopStart > n or loopEnd < 0 or loopEnd > n:
            raise _lib.ArgumentOutOfRangeError("Loop points must be less than the number of samples and non-negative.")
        if loopEnd < loopStart:
            raise _lib.ArgumentOutOfRangeError("The loop end must be greater than the loop start")
        return self._clone(looping=True, loopStart=loopStart, loopEnd=loopEnd)

    def WithAlignment(self, loopStartAlignment):                     # GcAdpcmFormat.cs:124-126
        return self._clone(alignmentMultiple=loopStartAlignment)

    def WithSamplesPerSeekTableEntry(self, samplesPerEntry):
        """What the container writers do per channel (GetCloneBuilder().WithSamplesPerSeekTableEntry(n).Build(),
        e.g. the reference's own test helper, Tests/Formats/GcAdpcmFormatTests.cs:149-156), batched."""
        return self._clone(samplesPerSeekTableEntry=samplesPerEntry)

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
            with _lib.reporting(config.Progress if config is not None else None, -(-n // 14)):
                check(_lib.lib().vga_gcadpcm_encode_batch(_ptr_array(i16p, pcm16.Channels), nch, n, h1, h2, _i16(coefs),
                                                           _ptr_array(u8p, adpcm)))
        chans = [GcAdpcmChannel(adpcm[i], coefs[i].copy(), n) for i in range(nch)]
        # new GcAdpcmFormatBuilder(channels, rate).WithLoop(pcm16.Looping, LoopStart, LoopEnd).Build() (:70-73)
        return GcAdpcmFormat(chans, pcm16.SampleRate, pcm16.Looping, pcm16.LoopStart, pcm16.LoopEnd)

    def ToPcm16(self):                           # GcAdpcmFormat.cs:42-54
        if not self.Channels:
            return Pcm16Format([], self.SampleRate)
        if all(c._pcm is not None for c in self.Channels):
            pcm = [c._pcm for c in self.Channels]                    # GetPcmAudio(): already decoded by the build
        else:
            n = self.Channels[0].SampleCount
            pcm = GcAdpcmDecoder.Decode([c.GetAdpcmAudio() for c in self.Channels],
                                        np.stack([c.Coefs for c in self.Channels]), GcAdpcmParameters(SampleCount=n))
        out = Pcm16Format(pcm, self.SampleRate)
        out.Looping, out.LoopStart, out.LoopEnd = self.Looping, self.LoopStart, self.LoopEnd
        return out

    def BuildSeekTable(self, entryCount, bigEndian=True):
        """GcAdpcmFormat.BuildSeekTable (GcAdpcmFormat.cs:99-113): the channels' seek tables
        (GcAdpcmSeekTable.cs:25-38, built on the device) interleaved by 2, resized to entryCount."""
        tables = [c.GetSeekTable() for c in self.Channels]
        nch = len(tables)
        entries = max((len(t) // 2 for t in tables), default=0)
        inter = np.zeros(entries * 2 * nch, dtype=np.int16)
        for c, t in enumerate(tables):                               # ArrayExtensions.Interleave(2)
            v = inter.reshape(entries, nch, 2)
            v[:len(t) // 2, c, :] = t.reshape(-1, 2)
        out = np.zeros(entryCount * 2 * nch, dtype=np.int16)         # Array.Resize
        m = min(len(out), len(inter))
        out[:m] = inter[:m]
        return out.astype(">i2" if bigEndian else "<i2").tobytes()


def encode_files(pcm16_list, configs=None):
    """The reference's batch conversion (VGAudio.Cli/Batch.cs:24-25: Parallel.ForEach over files, each file
    GcAdpcmFormat.EncodeFromPcm16) as ONE ragged GPU call (vga_gcadpcm_encode_batch_v): `pcm16_list` holds one
    Pcm16Format per file, of any length and channel count; returns one GcAdpcmFormat per file, each what
    GcAdpcmFormat().EncodeFromPcm16(file) returns.  configs: None or one GcAdpcmParameters (or None) per file
    (History1 / History2; a SampleCount override takes the per-file path)."""
    files = list(pcm16_list)
    configs = list(configs) if configs is not None else [None] * len(files)
    if any(c is not None and c.SampleCount != -1 for c in configs):
        return [GcAdpcmFormat().EncodeFromPcm16(f, c) for f, c in zip(files, configs)]
    chans, counts, h1, h2 = [], [], [], []
    for f, c in zip(files, configs):
        for ch in f.Channels:
            chans.append(ch)
            counts.append(f.SampleCount)
            h1.append(c.History1 if c else 0)
            h2.append(c.History2 if c else 0)
    nch = len(chans)
    counts = np.array(counts, dtype=np.int32)
    coefs = np.zeros((nch, 16), dtype=np.int16)
    adpcm = [np.zeros(GcAdpcmMath.SampleCountToByteCount(int(n)), dtype=np.uint8) for n in counts]
    h1 = np.array(h1, dtype=np.int16)
    h2 = np.array(h2, dtype=np.int16)
    if nch:
        check(_lib.lib().vga_gcadpcm_encode_batch_v(_ptr_array(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                                     _i16(h1), _i16(h2), _i16(coefs), _ptr_array(u8p, adpcm)))
    out, at = [], 0
    for f in files:
        k = f.ChannelCount
        cs = [GcAdpcmChannel(adpcm[at + i], coefs[at + i].copy(), f.SampleCount) for i in range(k)]
        out.append(GcAdpcmFormat(cs, f.SampleRate, f.Looping, f.LoopStart, f.LoopEnd))
        at += k
    return out


def decode_files(formats):
    """ToPcm16 (GcAdpcmFormat.cs:42-54) of many files in ONE ragged GPU call (vga_gcadpcm_decode_batch_v)."""
    formats = list(formats)
    chans, coefs, counts = [], [], []
    for f in formats:
        for c in f.Channels:
            chans.append(np.ascontiguousarray(c.GetAdpcmAudio(), dtype=np.uint8))
            coefs.append(c.Coefs)
            counts.append(c.SampleCount)
    nch = len(chans)
    counts = np.array(counts, dtype=np.int32)
    pcm = [np.zeros(int(n), dtype=np.int16) for n in counts]
    if nch:
        co = np.ascontiguousarray(np.stack(coefs), dtype=np.int16)
        check(_lib.lib().vga_gcadpcm_decode_batch_v(_ptr_array(u8p, chans), _i16(co), counts.ctypes.data_as(C.POINTER(C.c_int)), nch,
                                                     None, None, _ptr_array(i16p, pcm)))
    out, at = [], 0
    for f in formats:
        k = f.ChannelCount
        p = Pcm16Format(pcm[at:at + k], f.SampleRate)
        p.Looping, p.LoopStart, p.LoopEnd = f.Looping, f.LoopStart, f.LoopEnd
        out.append(p)
        at += k
    return out
