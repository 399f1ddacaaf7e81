"""Host-side mirror of the reference's CRI ADX codec and format classes, routed through the C
ABI into the HIP kernels.  Mirrors (paths relative to /root/reference/src/VGAudio/):
  CriAdxType         Codecs/CriAdx/CriAdxType.cs
  CriAdxParameters   Codecs/CriAdx/CriAdxParameters.cs:5-12
  CriAdxCodec        Codecs/CriAdx/CriAdxCodec.cs:9 (Decode), :56 (Encode), :173 (CalculateCoefficients)
  CriAdxHelpers      Formats/CriAdx/CriAdxHelpers.cs:7-31
  CriAdxChannel/Format  Formats/CriAdx/CriAdxFormat.cs:34-88
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, i16p, u8p
from .gcadpcm import Pcm16Format, _as_channels, _i16, _ptr_array


class CriAdxType:
    Fixed = 2
    Linear = 3
    Exponential = 4


class CriAdxParameters:
    def __init__(self, SampleRate=48000, HighpassFrequency=500, FrameSize=18, Version=4, History=0, Padding=0,
                 Type=CriAdxType.Linear, Filter=0, Progress=None):
        self.SampleRate, self.HighpassFrequency, self.FrameSize = SampleRate, HighpassFrequency, FrameSize
        self.Version, self.History, self.Padding, self.Type, self.Filter = Version, History, Padding, Type, Filter
        self.Progress = Progress

    def _c(self):
        return _lib.AdxParams(self.SampleRate, self.HighpassFrequency, self.FrameSize, self.Version,
                              int(self.History), self.Padding, self.Type, self.Filter)


class CriAdxHelpers:
    @staticmethod
    def NibbleCountToSampleCount(n, frameSize):
        return _lib.lib().vga_adx_nibble_count_to_sample_count(int(n), frameSize)

    @staticmethod
    def SampleCountToNibbleCount(n, frameSize):
        return _lib.lib().vga_adx_sample_count_to_nibble_count(int(n), frameSize)

    @staticmethod
    def SampleCountToByteCount(n, frameSize):
        return _lib.lib().vga_adx_sample_count_to_byte_count(int(n), frameSize)


class CriAdxCodec:
    @staticmethod
    def CalculateCoefficients(highpassFreq, sampleRate):
        c = np.zeros(2, dtype=np.int16)
        check(_lib.lib().vga_adx_calculate_coefficients(highpassFreq, sampleRate, _i16(c)))
        return c

    @staticmethod
    def Encode(pcm, config):
        """byte[] Encode(short[] pcm, CriAdxParameters config); sets config.History like the reference
        (CriAdxCodec.cs:73).  A list / 2-D array encodes a batch (History then becomes an array)."""
        chans, single = _as_channels(pcm, np.int16)
        nch = len(chans)
        n = len(chans[0]) if nch else 0
        if any(len(c) != n for c in chans):
            raise _lib.ArgumentError("channels of one batch must have equal length")
        cp = config._c()
        nbytes = _lib.lib().vga_adx_encoded_byte_count(n, C.byref(cp))
        if nbytes < 0:
            check(nbytes)
        outs = [np.zeros(nbytes, dtype=np.uint8) for _ in range(nch)]
        hist = np.zeros(max(nch, 1), dtype=np.int16)
        with _lib.reporting(config.Progress, nbytes // config.FrameSize):      # frames per channel (CriAdxCodec.cs:101)
            check(_lib.lib().vga_adx_encode_batch(_ptr_array(i16p, chans), nch, n, C.byref(cp), _ptr_array(u8p, outs),
                                                   _i16(hist)))
        config.History = int(hist[0]) if single else hist[:nch].copy()
        return outs[0] if single else outs

    @staticmethod
    def Decode(adpcm, sampleCount, config=None):
        config = config or CriAdxParameters()
        chans, single = _as_channels(adpcm, np.uint8)
        nch = len(chans)
        nb = len(chans[0]) if nch else 0
        if any(len(c) != nb for c in chans):
            raise _lib.ArgumentError("channels of one batch must have equal length")
        cp = config._c()
        outs = [np.zeros(max(sampleCount, 0), dtype=np.int16) for _ in range(nch)]
        check(_lib.lib().vga_adx_decode_batch(_ptr_array(u8p, chans), nb, nch, sampleCount, C.byref(cp),
                                               _ptr_array(i16p, outs)))
        return outs[0] if single else outs


class CriAdxChannel:
    def __init__(self, audio, history=0, version=4):
        self.Audio, self.History, self.Version = audio, history, version


def _get_next_multiple(value, multiple):
    # Utilities/Helpers.cs GetNextMultiple
    if multiple <= 0:
        return value
    if value % multiple == 0:
        return value
    return value + multiple - value % multiple


class CriAdxFormat:
    """IAudioFormat for CRI ADX; EncodeFromPcm16 / ToPcm16 are one batched GPU call each."""

    def __init__(self, channels=None, sampleCount=0, sampleRate=48000, frameSize=18, highpassFrequency=500,
                 alignmentSamples=0, type_=CriAdxType.Linear, version=4, looping=False, loopStart=0, loopEnd=0):
        self.Channels = list(channels) if channels is not None else []
        self.UnalignedSampleCount = sampleCount
        self.SampleRate, self.FrameSize, self.HighpassFrequency = sampleRate, frameSize, highpassFrequency
        self.AlignmentSamples, self.Type, self.Version = alignmentSamples, type_, version
        self.Looping = bool(looping)                     # AudioFormatBaseBuilder.WithLoop(false) zeroes the points
        self.UnalignedLoopStart, self.UnalignedLoopEnd = (loopStart, loopEnd) if looping else (0, 0)

    @property
    def ChannelCount(self):
        return len(self.Channels)

    @property
    def SampleCount(self):
        return self.UnalignedSampleCount + self.AlignmentSamples

    @property
    def LoopStart(self):                             # CriAdxFormat.cs:18
        return self.UnalignedLoopStart + self.AlignmentSamples

    @property
    def LoopEnd(self):                               # :19
        return self.UnalignedLoopEnd + self.AlignmentSamples

    def EncodeFromPcm16(self, pcm16, config=None):
        config = config or CriAdxParameters()
        spf = (config.FrameSize - 2) * 2
        multiple = spf * 2 if pcm16.ChannelCount == 1 else spf
        loop_start = pcm16.LoopStart if pcm16.Looping else 0
        alignment = _get_next_multiple(loop_start, multiple) - loop_start                  # CriAdxFormat.cs:59-62
        if config.Progress is not None:
            frames = -(-CriAdxHelpers.SampleCountToByteCount(pcm16.SampleCount, config.FrameSize) // config.FrameSize)
            config.Progress.SetTotal(frames * pcm16.ChannelCount)
        ch_cfg = CriAdxParameters(SampleRate=pcm16.SampleRate, FrameSize=config.FrameSize, Padding=alignment,
                                  Filter=config.Filter, Type=config.Type, Version=config.Version,
                                  Progress=config.Progress)
        if pcm16.ChannelCount:
            audio = CriAdxCodec.Encode(pcm16.Channels, ch_cfg)
            hist = np.atleast_1d(ch_cfg.History)
        else:
            audio, hist = [], []
        chans = [CriAdxChannel(audio[i], int(hist[i]), ch_cfg.Version) for i in range(pcm16.ChannelCount)]
        return CriAdxFormat(chans, pcm16.SampleCount, pcm16.SampleRate, config.FrameSize, 500, alignment, config.Type,
                            config.Version, pcm16.Looping, pcm16.LoopStart, pcm16.LoopEnd)

    def ToPcm16(self):
        if not self.Channels:
            return Pcm16Format([], self.SampleRate)
        opts = CriAdxParameters(SampleRate=self.SampleRate, FrameSize=self.FrameSize, Padding=self.AlignmentSamples,
                                HighpassFrequency=self.HighpassFrequency, Type=self.Type, Version=self.Version)
        pcm = CriAdxCodec.Decode([c.Audio for c in self.Channels], self.UnalignedSampleCount, opts)   # :39-48
        out = Pcm16Format(pcm, self.SampleRate)
        out.Looping, out.LoopStart, out.LoopEnd = self.Looping, self.UnalignedLoopStart, self.UnalignedLoopEnd
        return out


def encode_files(pcm16_list, configs=None):
    """CriAdxFormat.EncodeFromPcm16 (CriAdxFormat.cs:57-88) of many files -- any lengths, channel counts, sample rates and
    parameters -- in ONE ragged GPU call (vga_adx_encode_batch_v): the reference's Batch.cs runs a worker per file
    (VGAudio.Cli/Batch.cs:24-25).  Returns one CriAdxFormat per file, each what CriAdxFormat().EncodeFromPcm16(file, config)
    returns."""
    files = list(pcm16_list)
    configs = [c or CriAdxParameters() for c in (configs if configs is not None else [None] * len(files))]
    chans, lens, per, align = [], [], [], []
    for f, cfg in zip(files, configs):
        spf = (cfg.FrameSize - 2) * 2
        multiple = spf * 2 if f.ChannelCount == 1 else spf
        loop_start = f.LoopStart if f.Looping else 0
        a = _get_next_multiple(loop_start, multiple) - loop_start                              # CriAdxFormat.cs:59-62
        align.append(a)
        for ch in f.Channels:
            chans.append(ch)
            lens.append(f.SampleCount)
            per.append(CriAdxParameters(SampleRate=f.SampleRate, FrameSize=cfg.FrameSize, Padding=a, Filter=cfg.Filter,
                                        Type=cfg.Type, Version=cfg.Version)._c())
    nch = len(chans)
    L = _lib.lib()
    params = (_lib.AdxParams * max(nch, 1))(*per)
    counts = np.array(lens, dtype=np.int32)
    outs = []
    for c in range(nch):
        nb = L.vga_adx_encoded_byte_count(int(counts[c]), C.byref(params[c]))
        if nb < 0:
            check(nb)
        outs.append(np.zeros(nb, dtype=np.uint8))
    hist = np.zeros(max(nch, 1), dtype=np.int16)
    if nch:
        check(L.vga_adx_encode_batch_v(_ptr_array(i16p, chans), counts.ctypes.data_as(C.POINTER(C.c_int)), nch, params,
                                       _ptr_array(u8p, outs), _i16(hist)))
    result, at = [], 0
    for f, cfg, a in zip(files, configs, align):
        k = f.ChannelCount
        cs = [CriAdxChannel(outs[at + i], int(hist[at + i]), cfg.Version) for i in range(k)]
        result.append(CriAdxFormat(cs, f.SampleCount, f.SampleRate, cfg.FrameSize, 500, a, cfg.Type, cfg.Version, f.Looping,
                                   f.LoopStart, f.LoopEnd))
        at += k
    return result


class CriAdxKey:
    """Codecs/CriAdx/CriAdxKey.cs:10-56: CriAdxKey(seed, mult, inc) | CriAdxKey(keyCode) | CriAdxKey(keyString)."""

    def __init__(self, *args):
        k = _lib.AdxKeyC()
        self.KeyString = None
        if len(args) == 3:
            k.seed, k.mult, k.inc = (int(a) for a in args)
        elif len(args) == 1 and isinstance(args[0], str):
            check(_lib.lib().vga_adx_key_from_string(args[0].encode("ascii"), C.byref(k)))
            self.KeyString = args[0] or None
        elif len(args) == 1:
            check(_lib.lib().vga_adx_key_from_code(int(args[0]), C.byref(k)))
        else:
            raise _lib.ArgumentError("CriAdxKey(seed, mult, inc) | CriAdxKey(keyCode) | CriAdxKey(keyString)")
        self.c = k

    Seed = property(lambda self: self.c.seed)
    Mult = property(lambda self: self.c.mult)
    Inc = property(lambda self: self.c.inc)

    @property
    def KeyCode(self):
        return int(_lib.lib().vga_adx_key_code(C.byref(self.c)))


class CriAdxEncryption:
    """Codecs/CriAdx/CriAdxEncryption.cs: EncryptDecrypt (:8-14) and FindKey (:43-57, over the caller's candidates)."""

    @staticmethod
    def EncryptDecrypt(adpcm, key, encryptionType, frameSize):
        """In place on the caller's uint8 arrays, like the reference."""
        for a in adpcm:
            if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags.c_contiguous and a.flags.writeable):
                raise _lib.ArgumentError("EncryptDecrypt works in place on contiguous, writable uint8 arrays")
        if adpcm:
            check(_lib.lib().vga_adx_crypt(_ptr_array(u8p, adpcm), len(adpcm[0]), len(adpcm), C.byref(key.c), encryptionType, frameSize))

    @staticmethod
    def FindKey(adpcm, encryptionType, frameSize, keys):
        import torch
        if not keys:
            return None
        n = len(adpcm[0]) if adpcm else 0
        host = np.zeros((max(len(adpcm), 1), max(n, 1)), dtype=np.uint8)
        for i, a in enumerate(adpcm):
            host[i, :n] = a
        d = torch.from_numpy(host).cuda()
        arr = (_lib.AdxKeyC * len(keys))(*[k.c for k in keys])
        idx = C.c_int(-1)
        check(_lib.lib().vga_adx_find_key_device(d.data_ptr(), host.shape[1], n, len(adpcm), encryptionType, frameSize, arr, len(keys),
                                                 C.byref(idx), torch.cuda.current_stream().cuda_stream))
        return keys[idx.value] if idx.value >= 0 else None


class AdxFile:
    """VGAudio.Tools/CrackAdx/GuessAdx.cs:272-303: what the key search needs of one file -- the big-endian 16-bit frame
    headers in stream order and the frame that holds the first non-zero byte."""

    def __init__(self, audio, frameSize):
        audio = np.ascontiguousarray(audio, dtype=np.uint8).reshape(-1)
        self.FrameSize = frameSize
        self.FrameCount = len(audio) // frameSize
        a = audio[:self.FrameCount * frameSize].reshape(self.FrameCount, frameSize)
        self.Scales = ((a[:, 0].astype(np.uint16) << 8) | a[:, 1]).astype(np.uint16)
        nz = np.flatnonzero(audio)
        self.StartFrame = int(nz[0]) // frameSize if len(nz) else 0


class GuessAdx:
    """The search of the reference's `crackadx` tool (GuessAdx.Run / TryScale / FindStartingKey / KeyIsValid,
    GuessAdx.cs:118-218) for one file: every (scale index, multiplier, increment) on the GPU in one call."""

    @staticmethod
    def DefaultCandidates(encryptionType):
        m, n = np.zeros(0x2000, np.int32), np.zeros(0x2000, np.int32)
        nm, nn = C.c_int(), C.c_int()
        check(_lib.lib().vga_adx_guess_default_candidates(encryptionType, m.ctypes.data, C.byref(nm), n.ctypes.data, C.byref(nn)))
        return m[:nm.value].copy(), n[:nn.value].copy()

    @staticmethod
    def Run(adxFile, encryptionType, mults=None, incs=None, maxKeys=4096):
        scales = np.ascontiguousarray(adxFile.Scales, dtype=np.uint16)
        out = (_lib.AdxKeyC * maxKeys)()
        n = C.c_int(0)
        if mults is None or incs is None:
            mp = ip = None
            nm = ni = 0
        else:
            m = np.ascontiguousarray(mults, dtype=np.int32)
            i = np.ascontiguousarray(incs, dtype=np.int32)
            mp, ip, nm, ni = m.ctypes.data, i.ctypes.data, len(m), len(i)
        check(_lib.lib().vga_adx_guess_keys(scales.ctypes.data, len(scales), adxFile.StartFrame, encryptionType, mp, nm, ip, ni,
                                            out, maxKeys, C.byref(n)))
        return [CriAdxKey(out[k].seed, out[k].mult, out[k].inc) for k in range(n.value)]
