"""WAVE reader/writer for 16-bit PCM -- host-side mirror of
VGAudio/Containers/Wave/WaveReader.cs and WaveWriter.cs.  The RIFF header is parsed on the host
(vga_wave_parse); the interleaved <-> planar transposes run on the GPU.  There is no CPU path."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, i16p, u8p
from .gcadpcm import Pcm16Format, _ptr_array


class WaveReader:
    """AudioReader<WaveReader, WaveStructure, WaveConfiguration>: ReadFormat(file bytes) -> Pcm16Format."""

    @staticmethod
    def ReadMetadata(file):
        data = np.frombuffer(bytes(file), dtype=np.uint8)
        info = _lib.WaveInfoC()
        check(_lib.lib().vga_wave_parse(data.ctypes.data_as(u8p), len(data), C.byref(info)))
        return info

    @staticmethod
    def ReadFormat(file):
        data = np.frombuffer(bytes(file), dtype=np.uint8)
        info = _lib.WaveInfoC()
        check(_lib.lib().vga_wave_parse(data.ctypes.data_as(u8p), len(data), C.byref(info)))
        chans = [np.zeros(info.sample_count, dtype=np.int16) for _ in range(info.channel_count)]
        check(_lib.lib().vga_wave_read_pcm16(data.ctypes.data_as(u8p), len(data), C.byref(info), _ptr_array(i16p, chans)))
        return Pcm16Format(chans, info.sample_rate).WithLoop(bool(info.looping), info.loop_start, info.loop_end)


class WaveWriter:
    """AudioWriter<WaveWriter, WaveConfiguration> with Codec = Pcm16Bit: GetFile(Pcm16Format)."""

    @staticmethod
    def GetFile(audio):
        if not isinstance(audio, Pcm16Format):
            raise _lib.ArgumentError("WaveWriter takes a Pcm16Format (decode with ToPcm16 first)")
        p = _lib.WaveParamsC(audio.SampleRate, audio.SampleCount, int(audio.Looping), audio.LoopStart, audio.LoopEnd)
        size = _lib.lib().vga_wave_file_size(C.byref(p), audio.ChannelCount)
        if size < 0:
            check(int(size))
        out = np.zeros(size, dtype=np.uint8)
        check(_lib.lib().vga_wave_write_pcm16(_ptr_array(i16p, audio.Channels), audio.ChannelCount, C.byref(p),
                                              out.ctypes.data_as(u8p)))
        return out.tobytes()
