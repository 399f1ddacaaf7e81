/* wave_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of the reference's WAVE reader and
 * writer for 16-bit PCM (the step before the codec path).
 *
 *   VGAudio/Containers/Wave/WaveReader.cs:13-100, Utilities/Riff/RiffParser.cs:36-78, RiffChunk.cs, RiffSubchunk.cs,
 *   WaveFmtChunk.cs, WaveFormatExtensible.cs, WaveDataChunk.cs, WaveSmplChunk.cs,
 *   Utilities/Interleave.cs:188-207 (InterleavedByteToShort), :168-186 (ShortToInterleavedByte),
 *   VGAudio/Containers/Wave/WaveWriter.cs:12-165
 *
 * Pinned by the reference's own WavePcm16BuildAndParseEqual / WavePcm16LoopedBuildAndParseEqual
 * (VGAudio.Tests/Containers/WaveTests.cs:9-43) and hand-derived header bytes (tests/test_oracle_wave.py).
 * 8-bit PCM is parsed (header) but not converted: Pcm8 is out of scope. */
#include "oracle.h"

#include <string.h>

static int rd_i16(const uint8_t *p) { return (int16_t)(p[0] | (p[1] << 8)); }
static int rd_u16(const uint8_t *p) { return p[0] | (p[1] << 8); }
static int rd_i32(const uint8_t *p) { return (int)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }

/* MediaSubtypes.MediaSubtypePcm as Guid.ToByteArray() lays it out */
static const uint8_t kSubtypePcm[16] = {0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71};

/* Error codes: -3 = InvalidDataException (the reference's message in the comment), -6 = the reader ran off the
 * end of the file (EndOfStreamException / ArgumentOutOfRangeException out of BinaryReader). */
int vgo_wave_parse(const uint8_t *file, long file_len, vgo_wave_info *w)
{
    memset(w, 0, sizeof *w);
#define NEED(n) do { if (pos + (n) > file_len) return -6; } while (0)
    long pos = 0;
    NEED(12);
    if (memcmp(file, "RIFF", 4) != 0) return -3;                        /* "Not a valid RIFF file" (RiffChunk.cs:21-24) */
    long riff_size = rd_i32(file + 4);
    int is_wave = memcmp(file + 8, "WAVE", 4) == 0;
    pos = 12;
    long end_offset = 8 + riff_size;                                    /* RiffParser.cs:44-45 */
    int have_fmt = 0, have_data = 0, have_smpl = 0, have_ext = 0, ext_is_pcm = 0;
    int format_tag = 0, block_align = 0;
    while (pos + 8 < end_offset) {                                      /* :48 */
        NEED(8);
        const uint8_t *id = file + pos;
        long size = rd_i32(file + pos + 4);
        long start = pos + 8;
        if (size < 0) return -6;                                        /* ReadBytes(negative) throws */
        pos += 8;
        if (!memcmp(id, "fmt ", 4)) {                                   /* WaveFmtChunk.cs:16-34 */
            NEED(16);
            have_fmt = 1; have_ext = 0;
            format_tag = rd_u16(file + pos);
            w->channel_count = rd_i16(file + pos + 2);
            w->sample_rate = rd_i32(file + pos + 4);
            block_align = rd_i16(file + pos + 12);
            w->bits_per_sample = rd_i16(file + pos + 14);
            pos += 16;
            if (format_tag == 0xFFFE) {                                 /* WaveFormatExtensible.cs:20-27 */
                NEED(24);
                long ext_start = pos + 2;
                int ext_size = rd_i16(file + pos);
                have_ext = 1;
                ext_is_pcm = memcmp(file + pos + 8, kSubtypePcm, 16) == 0;
                pos += 24;
                long ext_end = ext_start + ext_size;
                if (ext_end > pos) pos = ext_end < file_len ? ext_end : file_len;   /* Ext.Extra = ReadBytes(remaining) */
            }
        } else if (!memcmp(id, "data", 4)) {                            /* WaveDataChunk.cs:9-15 */
            have_data = 1;
            w->data_offset = pos;
            w->data_size_declared = (int)size;
            long avail = file_len - pos;
            w->data_size = (int)(size < avail ? size : avail);           /* ReadBytes returns what is there */
            pos += w->data_size;
        } else if (!memcmp(id, "smpl", 4)) {                            /* WaveSmplChunk.cs:19-45 */
            NEED(36);
            have_smpl = 1;
            int loops = rd_i32(file + pos + 28);
            pos += 36;
            if (loops < 0) return -6;                                   /* new SampleLoop[negative] throws */
            w->smpl_loop_count = loops;
            for (int i = 0; i < loops; i++) {
                NEED(24);
                if (i == 0) { w->smpl_loop_start = rd_i32(file + pos + 8); w->smpl_loop_end = rd_i32(file + pos + 12); }
                pos += 24;
            }
        } else if (!memcmp(id, "fact", 4)) {                            /* WaveFactChunk.cs */
            NEED(4);
            pos += 4;
        }
        long end = start + size;                                        /* RiffParser.cs:72-76 */
        if (end > pos) pos = end < file_len ? end : file_len;           /* Extra = ReadBytes(remaining) */
        pos = end + (end & 1);
    }
#undef NEED
    /* ValidateWaveFile (WaveReader.cs:70-98), in its order */
    if (!is_wave) return -3;                                            /* "Not a valid WAVE file" */
    if (!have_fmt) return -3;                                           /* "File must have a valid fmt chunk" */
    if (!have_data) return -3;                                          /* "File must have a valid data chunk" */
    int bytes_per_sample = (w->bits_per_sample + 7) / 8;
    if (format_tag != 1 && format_tag != 0xFFFE) return -3;             /* "Must contain PCM data..." */
    if (w->bits_per_sample != 16 && w->bits_per_sample != 8) return -3; /* "Must have 8 or 16 bits per sample" */
    if (w->channel_count == 0) return -3;                               /* "Channel count must not be zero" */
    if (block_align != bytes_per_sample * w->channel_count) return -3;  /* "File has invalid block alignment" */
    if (have_ext && !ext_is_pcm) return -3;                             /* "... unsupported SubFormat" */
    if (w->channel_count < 0) return -6;                                /* new short[negative][] */
    /* ReadFile (:26-38) */
    w->sample_count_declared = w->data_size_declared / bytes_per_sample / w->channel_count;
    w->sample_count = w->data_size / bytes_per_sample / w->channel_count;   /* what InterleavedByteToShort yields (:190) */
    if (have_smpl && w->smpl_loop_count > 0) {
        w->loop_start = w->smpl_loop_start;
        w->loop_end = w->smpl_loop_end;
        w->looping = w->loop_end > w->loop_start;
    }
    /* ToAudioStream -> Pcm16FormatBuilder.WithLoop (AudioFormatBaseBuilder.cs:23-50) */
    if (w->looping) {
        if (w->loop_start < 0 || w->loop_start > w->sample_count) return -2;
        if (w->loop_end < 0 || w->loop_end > w->sample_count) return -2;
    } else {
        w->loop_start = w->loop_end = 0;
    }
    return 0;
}

/* InterleavedByteToShort (Interleave.cs:188-207): pcm_out[o]: info->sample_count shorts */
int vgo_wave_read_pcm16(const uint8_t *file, long file_len, const vgo_wave_info *w, int16_t *const *pcm_out)
{
    if (w->bits_per_sample != 16) return -1;
    const uint8_t *in = file + w->data_offset;
    for (int i = 0; i < w->sample_count; i++)
        for (int o = 0; o < w->channel_count; o++) {
            long off = ((long)i * w->channel_count + o) * 2;
            pcm_out[o][i] = (int16_t)(in[off] | (in[off + 1] << 8));
        }
    return 0;
}

/* ------------------------------------------------------------------ writer (WaveWriter.cs, 16-bit) */
static int channel_mask(int n)                                          /* :147-164 */
{
    switch (n) {
    case 4: return 0x0033;
    case 5: return 0x0133;
    case 6: return 0x0633;
    case 7: return 0x01f3;
    case 8: return 0x06f3;
    default: return (int)((1u << (n & 31)) - 1);                        /* C# shifts use the low 5 bits of the count */
    }
}

long vgo_wave_file_size(const vgo_wave_params *p, int nch)              /* :25-30 */
{
    long fmt = nch > 2 ? 40 : 16;
    long data = (long)nch * p->sample_count * 2;
    long riff = 4 + 8 + fmt + 8 + data + (p->looping ? 8 + 0x3c : 0);
    return 8 + riff;
}

static void wr16(uint8_t **c, int v) { (*c)[0] = (uint8_t)v; (*c)[1] = (uint8_t)(v >> 8); *c += 2; }
static void wr32(uint8_t **c, int v) { wr16(c, v); wr16(c, v >> 16); }

int vgo_wave_write_pcm16(const int16_t *const *pcm, int nch, const vgo_wave_params *p, uint8_t *file_out)
{
    long size = vgo_wave_file_size(p, nch);
    if (size > 0x7FFFFFFF || nch < 1) return -1;
    memset(file_out, 0, (size_t)size);
    uint8_t *c = file_out;
    memcpy(c, "RIFF", 4); c += 4;                                       /* :68-73 */
    wr32(&c, (int)(size - 8));
    memcpy(c, "WAVE", 4); c += 4;
    memcpy(c, "fmt ", 4); c += 4;                                       /* :75-95; positions stay even throughout */
    wr32(&c, nch > 2 ? 40 : 16);
    wr16(&c, nch > 2 ? 0xFFFE : 1);
    wr16(&c, nch);
    wr32(&c, p->sample_rate);
    wr32(&c, p->sample_rate * 2 * nch);
    wr16(&c, 2 * nch);
    wr16(&c, 16);
    if (nch > 2) {
        wr16(&c, 22);
        wr16(&c, 16);
        wr32(&c, channel_mask(nch));
        memcpy(c, kSubtypePcm, 16); c += 16;
    }
    if (p->looping) {                                                   /* :116-129 */
        memcpy(c, "smpl", 4); c += 4;
        wr32(&c, 0x3c);
        for (int i = 0; i < 7; i++) wr32(&c, 0);
        wr32(&c, 1);
        for (int i = 0; i < 3; i++) wr32(&c, 0);
        wr32(&c, p->loop_start);
        wr32(&c, p->loop_end);
        wr32(&c, 0);
        wr32(&c, 0);
    }
    memcpy(c, "data", 4); c += 4;                                       /* :97-114 */
    wr32(&c, nch * p->sample_count * 2);
    for (int i = 0; i < p->sample_count; i++)                           /* ShortToInterleavedByte (Interleave.cs:168-186) */
        for (int j = 0; j < nch; j++) wr16(&c, pcm[j][i]);
    return 0;
}
