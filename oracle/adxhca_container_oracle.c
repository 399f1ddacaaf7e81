/* adxhca_container_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of the reference's ADX
 * and HCA container writers, plus readers restated from the reference's parsers for build -> parse round trips.
 *
 *   ADX  VGAudio/Containers/Adx/AdxWriter.cs:14-139  (reader: AdxReader.cs:14-127)
 *   HCA  VGAudio/Containers/Hca/HcaWriter.cs:12-185  (reader: HcaReader.cs:20-231)
 *
 * Both writers fill a zeroed byte[FileSize] through a positioned stream (Containers/AudioWriter.cs:24-44); the
 * restatement keeps the stream cursor explicit because the reference relies on it: the ADX header is written field
 * after field regardless of HeaderSize, "(c)CRI" and the audio then overwrite what ran past it (:81-124), and the
 * footer lands wherever the interleaver left the cursor (:126-131, Utilities/Interleave.cs:43-78).
 * Pinned by: the reference's interleave vectors (tests/golden/interleave_kats.json), hand-derived header bytes
 * (tests/test_oracle_containers.py), CRC-16 check value.  The reference has no ADX/HCA container tests.
 * Encryption (EncryptionKey != null) is not restated: SURVEY.md 8f rank 4. */
#include "oracle.h"

#include <string.h>

static int next_multiple(int value, int multiple)                       /* Utilities/Helpers.cs:71-80 */
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}
static int div_round_up(int v, int d) { return (v + d - 1) / d; }
static int imin(int a, int b) { return a < b ? a : b; }

/* a positioned big-endian writer over a fixed byte[] (MemoryStream(file): writing past the end throws) */
typedef struct { uint8_t *buf; int size, pos, overflow; } cursor;
static void put8(cursor *c, int v) { if (c->pos + 1 > c->size) { c->overflow = 1; return; } c->buf[c->pos++] = (uint8_t)v; }
static void put16(cursor *c, int v) { put8(c, v >> 8); put8(c, v); }
static void put32(cursor *c, int v) { put16(c, v >> 16); put16(c, v); }
static void putn(cursor *c, const void *p, int n) { for (int i = 0; i < n; i++) put8(c, ((const uint8_t *)p)[i]); }

/* ------------------------------------------------------------------ ADX */
int vgo_adxfile_layout_for(const vgo_adxfile_params *p, int nch, vgo_adxfile_layout *L)
{
    if (nch < 1 || nch > 255 || p->frame_size < 3 || p->frame_size > 255) return -1;
    memset(L, 0, sizeof *L);
    int spf = (p->frame_size - 2) * 2;                                                     /* :27 */
    L->sample_count = (p->trim_file && p->looping) ? p->loop_end + spf * 3 : p->sample_count;   /* :21 */
    L->frame_count = div_round_up(L->sample_count, spf);                                   /* :28 */
    L->base_header_size = p->looping ? (p->version == 4 ? 60 : 52) : (p->version == 4 ? 36 : 32);   /* :30 */
    if (p->looping) {                                                                      /* CalculateAlignmentBytes :58-69 */
        int start = vgo_adx_sample_count_to_byte_count(p->loop_start, p->frame_size) * nch + L->base_header_size + 4;
        L->alignment_bytes = next_multiple(start, 0x800) - start;
        if (p->version == 3) L->alignment_bytes += p->alignment_samples / spf * 0x800;
    }
    L->header_size = L->base_header_size + L->alignment_bytes;                             /* :31 */
    L->audio_offset = L->header_size + 4;
    L->audio_size = p->frame_size * L->frame_count * nch;
    L->footer_offset = L->audio_offset + L->audio_size;
    L->footer_size = p->looping ? next_multiple(L->footer_offset + p->frame_size, 0x800) - L->footer_offset : p->frame_size;   /* :35 */
    L->loop_start_offset = L->audio_offset + vgo_adx_sample_count_to_byte_count(p->loop_start, p->frame_size) * nch;
    L->loop_end_offset = L->audio_offset + next_multiple(vgo_adx_sample_count_to_byte_count(p->loop_end, p->frame_size), p->frame_size) * nch;
    L->file_size = L->audio_offset + L->audio_size + L->footer_size;                       /* :18 */
    return 0;
}

/* audio[c]: CriAdxChannel.Audio (audio_len bytes each); history[c]: CriAdxChannel.History.  file_out: file_size bytes. */
int vgo_adxfile_write(const uint8_t *const *audio, int audio_len, const int16_t *history, int nch,
                      const vgo_adxfile_params *p, uint8_t *file_out)
{
    vgo_adxfile_layout L;
    int rc = vgo_adxfile_layout_for(p, nch, &L);
    if (rc) return rc;
    memset(file_out, 0, (size_t)L.file_size);
    cursor c = {file_out, L.file_size, 0, 0};
    /* WriteHeader :81-117 */
    put16(&c, 0x8000);
    put16(&c, L.header_size);
    put8(&c, p->type);
    put8(&c, p->frame_size);
    put8(&c, 4);
    put8(&c, nch);
    put32(&c, p->sample_rate);
    put32(&c, L.sample_count);
    put16(&c, p->type != 2 ? p->highpass_frequency : 0);               /* CriAdxType.Fixed = 2 */
    put8(&c, p->version);
    put8(&c, p->encryption_type);
    if (p->version == 4) {
        put32(&c, 0);
        for (int i = 0; i < nch; i++) { put16(&c, history[i]); put16(&c, history[i]); }
        if (nch == 1) put32(&c, 0);
    }
    put16(&c, p->alignment_samples);
    put16(&c, p->looping ? 1 : 0);
    put32(&c, p->looping ? 1 : 0);
    put32(&c, p->loop_start);
    put32(&c, L.loop_start_offset);
    put32(&c, p->loop_end);
    put32(&c, L.loop_end_offset);
    c.pos = L.header_size - 2;
    putn(&c, "(c)CRI", 6);
    if (c.overflow) return -2;
    /* WriteData :119-131: Interleave(stream, FrameSize, FrameCount * FrameSize) from the cursor (= AudioOffset) */
    {
        int interleave = p->frame_size, input_size = audio_len, output_size = L.frame_count * p->frame_size;
        int in_blocks = div_round_up(input_size, interleave), out_blocks = div_round_up(output_size, interleave);
        int last_in = input_size - (in_blocks - 1) * interleave, last_out = output_size - (out_blocks - 1) * interleave;
        int blocks = imin(in_blocks, out_blocks);
        for (int b = 0; b < blocks; b++) {
            int cur_in = b == in_blocks - 1 ? last_in : interleave;
            int cur_out = b == out_blocks - 1 ? last_out : interleave;
            int n = imin(cur_in, cur_out);
            for (int i = 0; i < nch; i++) {
                putn(&c, audio[i] + (size_t)interleave * b, n);
                c.pos += cur_out - n;
            }
        }
        /* SetLength(max(outputSize * inputCount, Length)) leaves the cursor where the last block ended */
    }
    /* WriteFooter :133-138 */
    put16(&c, 0x8001);
    put16(&c, L.footer_size - 4);
    return c.overflow ? -2 : 0;
}

static int rd16s(const uint8_t *p) { return (int16_t)((p[0] << 8) | p[1]); }
static int rd32(const uint8_t *p) { return (int)(((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]); }

/* AdxReader.ReadHeader/ReadData (:73-127).  audio_out[c] (may be NULL): frame_size * frame_count bytes. */
int vgo_adxfile_read(const uint8_t *file, int file_len, vgo_adxfile_header *h, int16_t *history_out, uint8_t *const *audio_out)
{
    memset(h, 0, sizeof *h);
    if (file_len < 20 || ((file[0] << 8) | file[1]) != 0x8000) return -3;      /* :18-21 */
    int pos = 2;
    h->header_size = rd16s(file + pos); pos += 2;
    h->type = file[pos++];
    h->frame_size = file[pos++];
    h->bit_depth = file[pos++];
    h->channel_count = file[pos++];
    h->sample_rate = rd32(file + pos); pos += 4;
    h->sample_count = rd32(file + pos); pos += 4;
    h->highpass_frequency = rd16s(file + pos); pos += 2;
    h->version = file[pos++];
    h->revision = file[pos++];
    if (h->version >= 4) {
        pos += 4;
        for (int i = 0; i < h->channel_count; i++) {
            if (pos + 4 > file_len) return -3;
            if (history_out) history_out[i] = (int16_t)rd16s(file + pos);
            pos += 4;
        }
        if (h->channel_count == 1) pos += 4;
    }
    if (pos + 24 <= h->header_size && pos + 24 <= file_len) {              /* :103 */
        h->inserted_samples = rd16s(file + pos); pos += 2;
        h->loop_count = rd16s(file + pos); pos += 2;
        if (h->loop_count > 0) {
            h->looping = 1;
            h->loop_type = rd32(file + pos);
            h->loop_start_sample = rd32(file + pos + 4);
            h->loop_start_byte = rd32(file + pos + 8);
            h->loop_end_sample = rd32(file + pos + 12);
            h->loop_end_byte = rd32(file + pos + 16);
        }
    }
    if (!audio_out) return 0;
    int spf = (h->frame_size - 2) * 2;
    if (spf <= 0 || h->channel_count < 1) return -3;
    int audio_offset = h->header_size + 4;
    int frame_count = div_round_up(h->sample_count, spf);
    int audio_size = h->frame_size * frame_count * h->channel_count;
    if (audio_offset + audio_size > file_len) return -3;
    return vgo_deinterleave(file + audio_offset, audio_size, h->frame_size, h->channel_count, -1, audio_out);
}

/* ------------------------------------------------------------------ HCA */
int vgo_hcafile_size(const vgo_hca_info *h) { return h->header_size + h->frame_size * h->frame_count; }   /* :22 */

static int g_mask_ids;                                                  /* WriteChunkId :158-171 */
static void chunk_id(cursor *c, const char *id, int n)
{
    for (int i = 0; i < n; i++) put8(c, (id[i] && g_mask_ids) ? (id[i] | 0x80) : id[i]);
}

/* frames: frame_count * frame_size bytes (CriHcaFormat.AudioData flattened); comment: NUL-terminated or NULL;
 * volume: HcaInfo.Volume (1 = no rva chunk); encrypted_ids: an encryption key is configured (chunk ids get their
 * top bits); file_out: vgo_hcafile_size bytes. */
int vgo_hcafile_write(const vgo_hca_info *h, const uint8_t *frames, const char *comment, float volume, int encryption_type,
                      int encrypted_ids, uint8_t *file_out)
{
    g_mask_ids = encrypted_ids;                                        /* Configuration.EncryptionKey != null */
    int size = vgo_hcafile_size(h);
    memset(file_out, 0, (size_t)size);
    cursor c = {file_out, size, 0, 0};
    chunk_id(&c, "HCA\0", 4);                                          /* :84-89 */
    put16(&c, 0x0200);
    put16(&c, h->header_size);
    chunk_id(&c, "fmt\0", 4);                                          /* :91-103 */
    put8(&c, h->channel_count);
    put8(&c, h->sample_rate >> 16);
    put16(&c, h->sample_rate);
    put32(&c, h->frame_count);
    put16(&c, h->inserted_samples);
    put16(&c, h->appended_samples);
    chunk_id(&c, "comp", 4);                                           /* :105-118 */
    put16(&c, h->frame_size);
    put8(&c, h->min_resolution);
    put8(&c, h->max_resolution);
    put8(&c, h->track_count);
    put8(&c, h->channel_config);
    put8(&c, h->total_band_count);
    put8(&c, h->base_band_count);
    put8(&c, h->stereo_band_count);
    put8(&c, h->bands_per_hfr_group);
    put16(&c, 0);
    if (h->looping) {                                                  /* :120-129 */
        chunk_id(&c, "loop", 4);
        put32(&c, h->loop_start_frame);
        put32(&c, h->loop_end_frame);
        put16(&c, h->pre_loop_samples);
        put16(&c, h->post_loop_samples);
    }
    chunk_id(&c, "ciph", 4);                                           /* :131-135 */
    put16(&c, encryption_type);
    if (volume != 1.0f) {                                              /* :137-146 */
        uint32_t bits;
        memcpy(&bits, &volume, 4);
        chunk_id(&c, "rva\0", 4);
        put32(&c, (int)bits);
    }
    int blank = 1;                                                     /* string.IsNullOrWhiteSpace (:66) */
    if (comment)
        for (const char *s = comment; *s; s++)
            if (!(*s == ' ' || (*s >= 9 && *s <= 13))) blank = 0;
    if (blank) {
        chunk_id(&c, "pad", 3);                                        /* :154-157: three bytes, no terminator */
    } else {
        chunk_id(&c, "comm\0", 5);                                     /* :148-152 */
        putn(&c, comment, (int)strlen(comment) + 1);
    }
    if (c.overflow || c.pos > h->header_size - 2) return -2;
    c.pos = h->header_size - 2;                                        /* :75-79 */
    put16(&c, vgo_crc16(file_out, h->header_size - 2));
    putn(&c, frames, h->frame_size * h->frame_count);                  /* WriteData :173-179 */
    return c.overflow ? -2 : 0;
}

/* HcaReader.ReadHcaHeader (:60-121) for the chunks the writer emits; fills what the header carries.
 * comment_out: >= header_size bytes or NULL.  Returns -3 on a bad signature / unsupported chunk. */
int vgo_hcafile_read(const uint8_t *file, int file_len, vgo_hca_info *h, float *volume_out, int *encryption_type_out,
                     char *comment_out, int *version_out)
{
    memset(h, 0, sizeof *h);
    if (volume_out) *volume_out = 1.0f;
    if (encryption_type_out) *encryption_type_out = 0;
    if (comment_out) comment_out[0] = 0;
    if (file_len < 8) return -3;
    char id[5] = {0};
    for (int i = 0; i < 4; i++) id[i] = (char)(file[i] & 0x7f);
    if (memcmp(id, "HCA\0", 4) != 0) return -3;
    if (version_out) *version_out = rd16s(file + 4);
    h->header_size = rd16s(file + 6);
    if (h->header_size > file_len) return -3;
    int pos = 8;
    while (pos < h->header_size) {
        if (pos + 4 > h->header_size) return -3;
        for (int i = 0; i < 4; i++) id[i] = (char)(file[pos + i] & 0x7f);
        pos += 4;
        if (!memcmp(id, "fmt\0", 4)) {
            h->channel_count = file[pos];
            h->sample_rate = (file[pos + 1] << 16) | (file[pos + 2] << 8) | file[pos + 3];
            h->frame_count = rd32(file + pos + 4);
            h->inserted_samples = rd16s(file + pos + 8);
            h->appended_samples = rd16s(file + pos + 10);
            h->sample_count = h->frame_count * 1024 - h->inserted_samples - h->appended_samples;
            pos += 12;
        } else if (!memcmp(id, "comp", 4)) {
            h->frame_size = rd16s(file + pos);
            h->min_resolution = file[pos + 2]; h->max_resolution = file[pos + 3];
            h->track_count = file[pos + 4]; h->channel_config = file[pos + 5];
            h->total_band_count = file[pos + 6]; h->base_band_count = file[pos + 7];
            h->stereo_band_count = file[pos + 8]; h->bands_per_hfr_group = file[pos + 9];
            pos += 12;
        } else if (!memcmp(id, "loop", 4)) {
            h->looping = 1;
            h->loop_start_frame = rd32(file + pos); h->loop_end_frame = rd32(file + pos + 4);
            h->pre_loop_samples = rd16s(file + pos + 8); h->post_loop_samples = rd16s(file + pos + 10);
            int loop_end_sample = (h->loop_end_frame + 1) * 1024 - h->post_loop_samples - h->inserted_samples;   /* HcaInfo.LoopEndSample */
            if (loop_end_sample < h->sample_count) h->sample_count = loop_end_sample;
            pos += 12;
        } else if (!memcmp(id, "ciph", 4)) {
            if (encryption_type_out) *encryption_type_out = rd16s(file + pos);
            pos += 2;
        } else if (!memcmp(id, "rva\0", 4)) {
            uint32_t bits = (uint32_t)rd32(file + pos);
            if (volume_out) memcpy(volume_out, &bits, 4);
            pos += 4;
        } else if (!memcmp(id, "comm", 4)) {
            pos++;
            if (comment_out) {
                int k = 0;
                while (pos + k < h->header_size && file[pos + k]) { comment_out[k] = (char)file[pos + k]; k++; }
                comment_out[k] = 0;
                h->comment_length = k;
            }
            pos = h->header_size;
        } else if (!memcmp(id, "pad\0", 4)) {
            pos = h->header_size;
        } else {
            return -3;
        }
    }
    if (h->track_count < 1) h->track_count = 1;
    return 0;
}
