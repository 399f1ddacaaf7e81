"""Second, independent restatement (plain Python, written from the C#) of the two container writers whose oracle the
reference's tests do not pin: VGAudio/Containers/Adx/AdxWriter.cs:16-139 (+ Formats/CriAdx/CriAdxHelpers.cs,
Utilities/Interleave.cs:43-78 for one-frame interleave) and VGAudio/Containers/Hca/HcaWriter.cs:20-178.
Test infrastructure only (tests/test_pyref_crosscheck.py)."""
import struct

from . import crihca as _hca
from . import crypt as _crypt


def _next_multiple(value, multiple):                              # Helpers.GetNextMultiple
    if multiple <= 0:
        return value
    if value % multiple == 0:
        return value
    return value + multiple - value % multiple


def _div_up(a, b):
    return (a + b - 1) // b


def adx_sample_count_to_byte_count(sample_count, frame_size):     # CriAdxHelpers.cs:19-32
    nibbles_per_frame = frame_size * 2
    samples_per_frame = nibbles_per_frame - 4
    frames = sample_count // samples_per_frame
    extra_samples = sample_count % samples_per_frame
    extra_nibbles = 0 if extra_samples == 0 else extra_samples + 4
    nibbles = nibbles_per_frame * frames + extra_nibbles
    return (nibbles + 1) // 2                                     # DivideBy2RoundUp


def adx_write(audio, history, sample_rate, sample_count, looping=False, loop_start=0, loop_end=0, alignment_samples=0,
              frame_size=18, version=4, adx_type=3, highpass_frequency=500, encryption_type=0, trim_file=True, key=None):
    """audio: per-channel bytes (CriAdxChannel.Audio, whole frames); history: per-channel int16 (CriAdxChannel.History)."""
    nch = len(audio)
    spf = (frame_size - 2) * 2
    count = loop_end + spf * 3 if (trim_file and looping) else sample_count      # :20-21
    frame_count = _div_up(count, spf)
    base_header = (60 if version == 4 else 52) if looping else (36 if version == 4 else 32)
    alignment_bytes = 0
    if looping:                                                   # CalculateAlignmentBytes :57-68
        start_loop_offset = adx_sample_count_to_byte_count(loop_start, frame_size) * nch + base_header + 4
        alignment_bytes = _next_multiple(start_loop_offset, 0x800) - start_loop_offset
        if version == 3:
            alignment_bytes += alignment_samples // spf * 0x800
    header_size = base_header + alignment_bytes
    audio_offset = header_size + 4
    audio_size = frame_size * frame_count * nch
    footer_offset = audio_offset + audio_size
    footer_size = _next_multiple(footer_offset + frame_size, 0x800) - footer_offset if looping else frame_size
    loop_start_offset = audio_offset + adx_sample_count_to_byte_count(loop_start, frame_size) * nch
    loop_end_offset = audio_offset + _next_multiple(adx_sample_count_to_byte_count(loop_end, frame_size), frame_size) * nch
    out = bytearray(audio_offset + audio_size + footer_size)

    h = bytearray()
    h += struct.pack(">Hh", 0x8000, _i16(header_size))            # WriteHeader :83-118
    h += struct.pack(">BBBB", adx_type & 0xff, frame_size & 0xff, 4, nch & 0xff)
    h += struct.pack(">ii", sample_rate, count)
    h += struct.pack(">h", highpass_frequency if adx_type != 2 else 0)      # CriAdxType.Fixed = 2
    h += struct.pack(">BB", version & 0xff, encryption_type & 0xff)
    if version == 4:
        h += struct.pack(">i", 0)
        for c in range(nch):
            h += struct.pack(">hh", history[c], history[c])
        if nch == 1:
            h += struct.pack(">i", 0)
    h += struct.pack(">hh", _i16(alignment_samples), 1 if looping else 0)
    h += struct.pack(">iiiii", 1 if looping else 0, loop_start, loop_start_offset, loop_end, loop_end_offset)
    out[:len(h)] = h
    out[header_size - 2:header_size - 2 + 6] = b"(c)CRI"

    chans = [bytearray(a) for a in audio]                         # WriteData :120-132
    if key is not None:
        for i, a in enumerate(chans):
            _crypt.adx_crypt_channel(a, key, encryption_type, frame_size, i, nch)
    # Interleave(stream, FrameSize, FrameCount * FrameSize), Utilities/Interleave.cs:43-78, with the stream position it
    # leaves behind: only min(input blocks, output blocks) blocks are written, so when the trimmed length needs more
    # frames than the channels hold (TrimFile with a loop end near the end of the audio) the FOOTER THAT FOLLOWS lands right
    # after the last copied frame, inside the zero-filled audio area, not at FooterOffset (the stream was sized before).
    pos = audio_offset
    input_size = len(chans[0])
    output_size = frame_count * frame_size
    in_blocks, out_blocks = _div_up(input_size, frame_size), _div_up(output_size, frame_size)
    last_in = input_size - (in_blocks - 1) * frame_size
    last_out = output_size - (out_blocks - 1) * frame_size
    for b in range(min(in_blocks, out_blocks)):
        cur_in = last_in if b == in_blocks - 1 else frame_size
        cur_out = last_out if b == out_blocks - 1 else frame_size
        n = min(cur_in, cur_out)
        for a in chans:
            out[pos:pos + n] = a[frame_size * b:frame_size * b + n]
            pos += n
            if n < cur_out:
                pos += cur_out - n
    footer_offset = pos                                           # WriteFooter writes at the stream's position
    out[footer_offset:footer_offset + 4] = struct.pack(">Hh", 0x8001, _i16(footer_size - 4))
    return bytes(out)


def _i16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def hca_write(hca, frames, comment=None, volume=1.0, key_type=None):
    """hca: crihca.HcaInfo; frames: list of frame bytes (already encrypted when key_type is not None, as SetupWriter
    :38-43 does before writing); key_type: the CriHcaKey.KeyType written into the ciph chunk, chunk ids get their top bits."""
    encrypted = key_type is not None
    out = bytearray(hca.header_size + hca.frame_size * hca.frame_count)
    w = bytearray()

    def chunk_id(s):                                              # WriteChunkId :151-164
        b = bytearray(s.encode("utf-8"))
        if encrypted:
            for i in range(len(b)):
                if b[i] != 0:
                    b[i] |= 0x80
        return bytes(b)

    w += chunk_id("HCA\0") + struct.pack(">hh", 0x0200, _i16(hca.header_size))
    w += chunk_id("fmt\0") + struct.pack(">BBh", hca.channel_count & 0xff, (hca.sample_rate >> 16) & 0xff, _i16(hca.sample_rate))
    w += struct.pack(">iHH", hca.frame_count, hca.inserted_samples & 0xFFFF, hca.appended_samples & 0xFFFF)
    w += chunk_id("comp") + struct.pack(">hBBBBBBBBh", _i16(hca.frame_size), hca.min_resolution & 0xff, hca.max_resolution & 0xff,
                                        hca.track_count & 0xff, hca.channel_config & 0xff, hca.total_band_count & 0xff,
                                        hca.base_band_count & 0xff, hca.stereo_band_count & 0xff, hca.bands_per_hfr_group & 0xff, 0)
    if hca.looping:
        w += chunk_id("loop") + struct.pack(">iihh", hca.loop_start_frame, hca.loop_end_frame, _i16(hca.pre_loop_samples),
                                            _i16(hca.post_loop_samples))
    w += chunk_id("ciph") + struct.pack(">h", _i16(key_type if encrypted else 0))
    if struct.pack(">f", volume) != struct.pack(">f", 1.0):       # volume != 1 (float)
        w += chunk_id("rva\0") + struct.pack(">f", volume)
    if comment is None or comment.strip() == "":                  # string.IsNullOrWhiteSpace
        w += chunk_id("pad")
    else:
        w += chunk_id("comm\0") + comment.encode("utf-8") + b"\0"
    out[:len(w)] = w
    crc = _hca.crc16(out, hca.header_size - 2)                    # over the zero-padded header
    out[hca.header_size - 2:hca.header_size] = struct.pack(">H", crc)
    pos = hca.header_size
    for i in range(hca.frame_count):
        out[pos:pos + hca.frame_size] = frames[i][:hca.frame_size]
        pos += hca.frame_size
    return bytes(out)
