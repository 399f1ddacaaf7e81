"""Host-side mirror of the reference's CRI HCA classes, routed through the C ABI into the HIP kernels.
Mirrors (paths relative to /root/reference/src/VGAudio/):
  CriHcaQuality      Codecs/CriHca/CriHcaQuality.cs
  CriHcaParameters   Codecs/CriHca/CriHcaParameters.cs:3-15
  HcaInfo            Codecs/CriHca/HcaInfo.cs:5-59
  CriHcaEncoder      Codecs/CriHca/CriHcaEncoder.cs:49 (InitializeNew) -- parameters only; frames are encoded in batches
  CriHcaDecoder      Codecs/CriHca/CriHcaDecoder.cs:11 (Decode)
  CriHcaFormat       Formats/CriHca/CriHcaFormat.cs:26-84
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, i16p, u8p
from .gcadpcm import Pcm16Format, _ptr_array


class CriHcaQuality:
    NotSet, Highest, High, Middle, Low, Lowest = range(6)


class CriHcaParameters:
    def __init__(self, Quality=CriHcaQuality.High, Bitrate=0, LimitBitrate=False, ChannelCount=0, SampleRate=0,
                 SampleCount=-1, Looping=False, LoopStart=0, LoopEnd=0, Progress=None):
        self.Quality, self.Bitrate, self.LimitBitrate = Quality, Bitrate, LimitBitrate
        self.ChannelCount, self.SampleRate, self.SampleCount = ChannelCount, SampleRate, SampleCount
        self.Looping, self.LoopStart, self.LoopEnd, self.Progress = Looping, LoopStart, LoopEnd, Progress

    def _c(self):
        return _lib.HcaParamsC(self.Quality, self.Bitrate, int(self.LimitBitrate), self.ChannelCount, self.SampleRate,
                               self.SampleCount, int(self.Looping), self.LoopStart, self.LoopEnd)


class HcaInfo:
    """Thin attribute view over vga_hca_info with the reference's property names."""
    _MAP = dict(ChannelCount="channel_count", SampleRate="sample_rate", SampleCount="sample_count",
                FrameCount="frame_count", InsertedSamples="inserted_samples", AppendedSamples="appended_samples",
                HeaderSize="header_size", FrameSize="frame_size", MinResolution="min_resolution",
                MaxResolution="max_resolution", TrackCount="track_count", ChannelConfig="channel_config",
                TotalBandCount="total_band_count", BaseBandCount="base_band_count",
                StereoBandCount="stereo_band_count", HfrBandCount="hfr_band_count",
                BandsPerHfrGroup="bands_per_hfr_group", HfrGroupCount="hfr_group_count", Looping="looping",
                LoopStartFrame="loop_start_frame", LoopEndFrame="loop_end_frame", PreLoopSamples="pre_loop_samples",
                PostLoopSamples="post_loop_samples", UseAthCurve="use_ath_curve", CommentLength="comment_length")

    _HOST = ("Comment", "Volume", "EncryptionType")          # HcaInfo.cs:43-47: container-only fields

    def __init__(self, c=None):
        object.__setattr__(self, "c", c if c is not None else _lib.HcaInfoC())
        object.__setattr__(self, "Comment", None)
        object.__setattr__(self, "Volume", 1.0)
        object.__setattr__(self, "EncryptionType", 0)

    def __getattr__(self, name):
        return getattr(self.c, HcaInfo._MAP[name])

    def __setattr__(self, name, value):
        if name in HcaInfo._HOST:
            object.__setattr__(self, name, value)
        else:
            setattr(self.c, HcaInfo._MAP[name], int(value))

    @property
    def LoopStartSample(self):
        return self.LoopStartFrame * 1024 + self.PreLoopSamples - self.InsertedSamples

    @property
    def LoopEndSample(self):
        return (self.LoopEndFrame + 1) * 1024 - self.PostLoopSamples - self.InsertedSamples


class CriHcaEncoder:
    """CriHcaEncoder (Codecs/CriHca/CriHcaEncoder.cs): InitializeNew derives the stream's parameters; Encode / GetPendingFrame /
    PendingFrameCount / FramesProcessed are the reference's streaming shell (:126-163) over vga_hca_stream_* -- the frames
    themselves come from the same kernel the batched CriHcaFormat.EncodeFromPcm16 uses."""

    def __init__(self, hca, config=None):
        self.Hca = hca
        self._config = config
        self._stream = None

    @property
    def FrameSize(self):
        return self.Hca.FrameSize

    @staticmethod
    def InitializeNew(config):
        info = _lib.HcaInfoC()
        cp = config._c()
        check(_lib.lib().vga_hca_encoder_initialize(C.byref(cp), C.byref(info)))
        return CriHcaEncoder(HcaInfo(info), config)

    def _open(self):
        if self._stream is None:
            if self._config is None:
                raise _lib.InvalidOperationError("encoder was not created by InitializeNew")
            cp = self._config._c()
            h = C.c_void_p()
            check(_lib.lib().vga_hca_stream_create(C.byref(cp), None, C.byref(h)))
            self._stream = h
        return self._stream

    def Encode(self, pcm, hcaOut):
        """pcm: [ChannelCount][1024] (or longer rows); hcaOut: a writable uint8 array of FrameSize bytes.  Returns the number
        of frames output: the first in hcaOut, the others through GetPendingFrame."""
        rows = [np.ascontiguousarray(np.asarray(r, dtype=np.int16)[:1024]) for r in pcm]
        # the library reads ChannelCount row pointers and writes FrameSize bytes: what the reference would answer with an
        # IndexOutOfRangeException must not become an out-of-bounds access behind the C ABI
        if len(rows) != self.Hca.ChannelCount:
            raise _lib.ArgumentError("Encode takes [ChannelCount][1024] samples: %d rows for %d channels" % (len(rows), self.Hca.ChannelCount))
        if any(len(r) < 1024 for r in rows):
            raise _lib.ArgumentError("Encode takes [ChannelCount][1024] samples")
        if not (isinstance(hcaOut, np.ndarray) and hcaOut.dtype == np.uint8 and hcaOut.ndim == 1 and hcaOut.flags.c_contiguous
                and hcaOut.flags.writeable and hcaOut.size >= self.FrameSize):
            raise _lib.ArgumentError("hcaOut must be a writable, contiguous uint8 array of at least FrameSize (%d) bytes" % self.FrameSize)
        st = self._open()
        ptrs = (_lib.i16p * len(rows))(*[r.ctypes.data_as(_lib.i16p) for r in rows])
        n = C.c_int(0)
        check(_lib.lib().vga_hca_stream_encode(st, ptrs, hcaOut.ctypes.data_as(_lib.u8p), C.byref(n)))
        return n.value

    @property
    def PendingFrameCount(self):
        return _lib.lib().vga_hca_stream_pending_frame_count(self._stream) if self._stream else 0

    @property
    def FramesProcessed(self):
        return _lib.lib().vga_hca_stream_frames_processed(self._stream) if self._stream else 0

    def GetPendingFrame(self):
        if not self._stream:
            raise _lib.InvalidOperationError("There are no pending frames")
        out = np.zeros(self.FrameSize, dtype=np.uint8)
        check(_lib.lib().vga_hca_stream_get_pending_frame(self._stream, out.ctypes.data_as(_lib.u8p)))
        return out

    def close(self):
        if self._stream:
            _lib.lib().vga_hca_stream_destroy(self._stream)
            self._stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class CriHcaDecoder:
    @staticmethod
    def Decode(hca, audio, config=None):
        """short[][] Decode(HcaInfo, byte[][] audio); `audio` is [FrameCount][FrameSize] for one stream, or a
        list of such arrays for a batch of streams with the same HcaInfo."""
        single = isinstance(audio, np.ndarray) and audio.ndim == 2
        streams = [audio] if single else list(audio)
        flat = [np.ascontiguousarray(a, dtype=np.uint8).reshape(-1) for a in streams]
        need = hca.FrameCount * hca.FrameSize
        if any(len(f) < need for f in flat):
            raise _lib.ArgumentError("audio shorter than FrameCount * FrameSize")
        nch, n = hca.ChannelCount, max(hca.SampleCount, 0)
        outs = [np.zeros(n, dtype=np.int16) for _ in range(len(streams) * nch)]
        progress = config.Progress if config is not None else None
        if progress is not None:
            progress.SetTotal(hca.FrameCount * len(streams))
        with _lib.reporting(progress, hca.FrameCount):                 # frames per stream (CriHcaFormat.cs:71,79)
            check(_lib.lib().vga_hca_decode_batch(C.byref(hca.c), _ptr_array(u8p, flat), len(streams),
                                                   _ptr_array(i16p, outs)))
        per = [outs[s * nch:(s + 1) * nch] for s in range(len(streams))]
        return per[0] if single else per


class CriHcaFormat:
    def __init__(self, audioData=None, hca=None):
        self.AudioData = audioData     # [FrameCount][FrameSize] uint8
        self.Hca = hca

    @property
    def ChannelCount(self):
        return self.Hca.ChannelCount if self.Hca else 0

    def EncodeFromPcm16(self, pcm16, config=None):
        return CriHcaFormat.EncodeBatchFromPcm16([pcm16], config)[0]

    @staticmethod
    def EncodeBatchFromPcm16(pcm16_list, config=None):
        """One GPU call for a batch of equally shaped streams (the reference's batch mode runs one serial
        CriHcaFormat.EncodeFromPcm16 per file under Parallel.ForEach, Cli/Batch.cs:24-25)."""
        config = config or CriHcaParameters()
        first = pcm16_list[0]
        for p in pcm16_list:
            if (p.ChannelCount, p.SampleCount, p.SampleRate) != (first.ChannelCount, first.SampleCount, first.SampleRate):
                raise _lib.ArgumentError("streams of one batch must share channel count, length and sample rate")
        config.ChannelCount, config.SampleRate, config.SampleCount = first.ChannelCount, first.SampleRate, first.SampleCount
        config.Looping, config.LoopStart, config.LoopEnd = first.Looping, first.LoopStart, first.LoopEnd
        enc = CriHcaEncoder.InitializeNew(config)                      # CriHcaFormat.cs:44
        hca = enc.Hca
        if config.Progress is not None:
            config.Progress.SetTotal(hca.FrameCount * len(pcm16_list))
        chans = [c for p in pcm16_list for c in p.Channels]
        outs = [np.zeros(hca.FrameCount * hca.FrameSize, dtype=np.uint8) for _ in pcm16_list]
        info = _lib.HcaInfoC()
        cp = config._c()
        with _lib.reporting(config.Progress, hca.FrameCount):
            check(_lib.lib().vga_hca_encode_batch(_ptr_array(i16p, chans), len(pcm16_list), C.byref(cp), C.byref(info),
                                                   _ptr_array(u8p, outs)))
        return [CriHcaFormat(o.reshape(hca.FrameCount, hca.FrameSize), HcaInfo(info)) for o in outs]

    def ToPcm16(self, config=None):
        pcm = CriHcaDecoder.Decode(self.Hca, self.AudioData, config)
        return Pcm16Format(pcm, self.Hca.SampleRate)


def encode_files(pcm16_list, configs=None):
    """CriHcaFormat.EncodeFromPcm16 (CriHcaFormat.cs:34-84) of many files of any shape in ONE ragged GPU call
    (vga_hca_encode_batch_v); the reference encodes one file per worker (VGAudio.Cli/Batch.cs:24-25).  Returns one
    CriHcaFormat per file."""
    files = list(pcm16_list)
    configs = [c or CriHcaParameters() for c in (configs if configs is not None else [None] * len(files))]
    ns = len(files)
    cps = (_lib.HcaParamsC * max(ns, 1))()
    for s_, (f, cfg) in enumerate(zip(files, configs)):
        cfg.ChannelCount, cfg.SampleRate, cfg.SampleCount = f.ChannelCount, f.SampleRate, f.SampleCount     # :36-41
        cfg.Looping, cfg.LoopStart, cfg.LoopEnd = f.Looping, f.LoopStart, f.LoopEnd
        cps[s_] = cfg._c()
    infos = (_lib.HcaInfoC * max(ns, 1))()
    L = _lib.lib()
    for s_ in range(ns):
        check(L.vga_hca_encoder_initialize(C.byref(cps[s_]), C.byref(infos[s_])))                             # sizes the outputs
    outs = [np.zeros(infos[s_].frame_count * infos[s_].frame_size, dtype=np.uint8) for s_ in range(ns)]
    chans = [c for f in files for c in f.Channels]
    if ns:
        check(L.vga_hca_encode_batch_v(_ptr_array(i16p, chans), ns, cps, infos, _ptr_array(u8p, outs)))
    result = []
    for s_ in range(ns):
        info = _lib.HcaInfoC.from_buffer_copy(infos[s_])
        result.append(CriHcaFormat(outs[s_].reshape(info.frame_count, info.frame_size), HcaInfo(info)))
    return result


class CriHcaKey:
    """Codecs/CriHca/CriHcaKey.cs:8-39: CriHcaKey(keyCode) (KeyType 56) | CriHcaKey.Type0 / Type1."""
    Type0, Type1 = "Type0", "Type1"

    def __init__(self, key):
        self.DecryptionTable = np.zeros(256, dtype=np.uint8)
        self.EncryptionTable = np.zeros(256, dtype=np.uint8)
        if key == CriHcaKey.Type0:
            self.KeyType, self.KeyCode = 0, 0
        elif key == CriHcaKey.Type1:
            self.KeyType, self.KeyCode = 1, 0
        else:
            self.KeyType, self.KeyCode = 56, int(key)
        check(_lib.lib().vga_hca_key_tables(self.KeyType, self.KeyCode, self.DecryptionTable.ctypes.data_as(u8p),
                                            self.EncryptionTable.ctypes.data_as(u8p)))


class CriHcaEncryption:
    """Codecs/CriHca/CriHcaEncryption.cs:12-33."""

    @staticmethod
    def Crypt(hca, audio, key, doDecrypt):
        """In place on `audio` ([FrameCount][FrameSize] uint8, contiguous)."""
        if not (isinstance(audio, np.ndarray) and audio.dtype == np.uint8 and audio.flags.c_contiguous and audio.flags.writeable):
            raise _lib.ArgumentError("Crypt works in place on a contiguous, writable uint8 array")
        table = key.DecryptionTable if doDecrypt else key.EncryptionTable
        check(_lib.lib().vga_hca_crypt(audio.ctypes.data_as(u8p), hca.FrameCount, hca.FrameSize, table.ctypes.data_as(u8p)))

    @staticmethod
    def FindKey(hca, audio, keys):
        """CriHcaKey FindKey(HcaInfo hca, byte[][] audio) (CriHcaEncryption.cs:34-46) over the caller's candidate keys
        (the reference's own list, CriHcaEncryptionKeys.cs, is data and stays with the caller): the first key under
        which the first ten non-empty frames unpack, or None."""
        if not keys:
            return None
        frames = np.ascontiguousarray(audio, dtype=np.uint8).reshape(-1)
        tables = np.ascontiguousarray(np.stack([k.DecryptionTable for k in keys]), dtype=np.uint8)
        idx = C.c_int(-1)
        check(_lib.lib().vga_hca_find_key(C.byref(hca.c), frames.ctypes.data_as(u8p), len(frames) // hca.FrameSize,
                                          tables.ctypes.data_as(u8p), len(keys), C.byref(idx)))
        return keys[idx.value] if idx.value >= 0 else None
