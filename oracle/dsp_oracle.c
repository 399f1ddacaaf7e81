/* dsp_oracle.c -- CPU restatement of the reference's DSP container writer/reader for GC-ADPCM
 * (SURVEY.md 8f rank 2).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows VGAudio/Containers/Dsp/DspWriter.cs:14-103, DspReader.cs:14-117,
 * Utilities/Interleave.cs:43-78 (Interleave to a stream) and :80-117 (DeInterleave).
 * Parity status: the reference's tests only round-trip this container (Tests/Containers/DspTests.cs:9-19), so
 * the byte layout is pinned by the restatement + the hand-derived header KATs in tests/test_oracle_dsp.py.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define HEADER_SIZE 0x60
#define BYTES_PER_FRAME 8

static int next_multiple(int value, int multiple)
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}
static int div_round_up(int v, int d) { return v / d + (v % d != 0); }
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }

static void be16(uint8_t *p, int v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
static void be32(uint8_t *p, int v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
static int rd16(const uint8_t *p) { return (int16_t)((p[0] << 8) | p[1]); }
static int rd32(const uint8_t *p) { return (int)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]); }

/* InterleaveExtensions.Interleave(byte[][], Stream, interleaveSize, outputSize) (Utilities/Interleave.cs:43-78).
 * out: output_size * count bytes, zero-initialised by the caller (the bytes the stream version skips over);
 * output_size -1 = input_size. */
void vgo_interleave(const uint8_t *const *inputs, int count, int input_size, int interleave, int output_size, uint8_t *out)
{
    if (output_size == -1) output_size = input_size;
    int in_blocks = div_round_up(input_size, interleave), out_blocks = div_round_up(output_size, interleave);
    int last_in = input_size - (in_blocks - 1) * interleave, last_out = output_size - (out_blocks - 1) * interleave;
    int blocks = imin(in_blocks, out_blocks);
    size_t pos = 0;
    for (int b = 0; b < blocks; b++) {
        int cur_in = b == in_blocks - 1 ? last_in : interleave;
        int cur_out = b == out_blocks - 1 ? last_out : interleave;
        int n = imin(cur_in, cur_out);
        for (int i = 0; i < count; i++) {
            memcpy(out + pos, inputs[i] + (size_t)interleave * b, (size_t)n);
            pos += (size_t)cur_out;                                    /* Write advances n, then Position += cur_out - n */
        }
    }
}

/* DeInterleave(byte[] input, interleaveSize, outputCount, outputSize) (Utilities/Interleave.cs:80-117).
 * outs[o]: output_size bytes, zero-initialised by the caller; output_size -1 = in_len / count.  Returns -1 when
 * in_len is not divisible by count (the reference throws). */
int vgo_deinterleave(const uint8_t *in, int in_len, int interleave, int count, int output_size, uint8_t *const *outs)
{
    if (in_len % count != 0) return -1;
    int input_size = in_len / count;
    if (output_size == -1) output_size = input_size;
    int in_blocks = div_round_up(input_size, interleave), out_blocks = div_round_up(output_size, interleave);
    int last_in = input_size - (in_blocks - 1) * interleave, last_out = output_size - (out_blocks - 1) * interleave;
    int blocks = imin(in_blocks, out_blocks);
    for (int b = 0; b < blocks; b++) {
        int cur_in = b == in_blocks - 1 ? last_in : interleave;
        int cur_out = b == out_blocks - 1 ? last_out : interleave;
        int n = imin(cur_in, cur_out);
        for (int o = 0; o < count; o++)
            memcpy(outs[o] + (size_t)interleave * b, in + (size_t)interleave * b * count + (size_t)cur_in * o, (size_t)n);
    }
    return 0;
}

/* DspWriter.cs:22-36 */
static void dsp_geometry(const vgo_dsp_params *p, int nch, vgo_dsp_layout *g)
{
    int alignment_samples = next_multiple(p->loop_start, p->loop_point_alignment) - p->loop_start;
    g->loop_start = p->loop_start + alignment_samples;
    g->loop_end = p->loop_end + alignment_samples;
    g->sample_count = (p->trim_file && p->looping) ? g->loop_end : imax(p->sample_count, g->loop_end);
    g->bytes_per_interleave = vgo_gc_sample_count_to_byte_count(p->samples_per_interleave);
    g->frames_per_interleave = g->bytes_per_interleave / BYTES_PER_FRAME;
    g->start_addr = vgo_gc_sample_to_nibble(p->looping ? g->loop_start : 0);
    g->end_addr = vgo_gc_sample_to_nibble(p->looping ? g->loop_end : g->sample_count - 1);
    g->cur_addr = vgo_gc_sample_to_nibble(0);
    /* AudioDataSize :99-100 */
    g->audio_data_size = next_multiple(vgo_gc_sample_count_to_byte_count(g->sample_count), nch == 1 ? 1 : BYTES_PER_FRAME);
    g->file_size = (HEADER_SIZE + g->audio_data_size) * nch;      /* :18 */
}

int vgo_dsp_layout_for(const vgo_dsp_params *p, int nch, vgo_dsp_layout *out)
{
    if (nch < 1 || p->samples_per_interleave < 1 || p->samples_per_interleave % 14 != 0) return -2;   /* DspConfiguration.cs:31-46 */
    dsp_geometry(p, nch, out);
    return 0;
}

/* DspWriter.cs:38-97.  adpcm[c]: GetAdpcmAudio() of channel c (adpcm_len bytes each); contexts: nch*3 shorts
 * (pred/scale, hist1, hist2).  file_out: layout.file_size bytes (zero-initialised here). */
int vgo_dsp_write(const uint8_t *const *adpcm, int adpcm_len, const int16_t *coefs, const int16_t *gain,
                  const int16_t *start_context, const int16_t *loop_context, int nch, const vgo_dsp_params *p,
                  uint8_t *file_out)
{
    vgo_dsp_layout g;
    int rc = vgo_dsp_layout_for(p, nch, &g);
    if (rc) return rc;
    memset(file_out, 0, (size_t)g.file_size);
    for (int i = 0; i < nch; i++) {                                    /* WriteHeader :52-80 */
        uint8_t *h = file_out + (size_t)HEADER_SIZE * i;
        be32(h + 0x00, g.sample_count);
        be32(h + 0x04, vgo_gc_sample_count_to_nibble_count(g.sample_count));
        be32(h + 0x08, p->sample_rate);
        be16(h + 0x0c, p->looping ? 1 : 0);
        be16(h + 0x0e, 0);                                             /* Format: 0 for ADPCM */
        be32(h + 0x10, g.start_addr);
        be32(h + 0x14, g.end_addr);
        be32(h + 0x18, g.cur_addr);
        for (int k = 0; k < 16; k++) be16(h + 0x1c + 2 * k, coefs[i * 16 + k]);
        be16(h + 0x3c, gain ? gain[i] : 0);
        for (int k = 0; k < 3; k++) be16(h + 0x3e + 2 * k, start_context[i * 3 + k]);
        if (p->looping)
            for (int k = 0; k < 3; k++) be16(h + 0x44 + 2 * k, loop_context[i * 3 + k]);
        be16(h + 0x4a, nch == 1 ? 0 : nch);
        be16(h + 0x4c, nch == 1 ? 0 : g.frames_per_interleave);
    }
    uint8_t *data = file_out + (size_t)HEADER_SIZE * nch;              /* WriteData :82-94 */
    if (nch == 1) {
        int n = vgo_gc_sample_count_to_byte_count(g.sample_count);
        if (n > adpcm_len) return -1;                                  /* Stream.Write would throw */
        memcpy(data, adpcm[0], (size_t)n);
    } else {
        vgo_interleave(adpcm, nch, adpcm_len, g.bytes_per_interleave, g.audio_data_size, data);
    }
    return 0;
}

/* DspReader.cs:38-117: header of channel 0 + per-channel info; audio de-interleaved (Interleave.cs:80-117).
 * adpcm_out[c]: SampleCountToByteCount(sample_count) bytes (from a first call with adpcm_out == NULL). */
int vgo_dsp_read(const uint8_t *file, int file_len, vgo_dsp_header *hdr, int16_t *coefs_out, int16_t *gain_out,
                 int16_t *start_context_out, int16_t *loop_context_out, uint8_t *const *adpcm_out)
{
    if (file_len < HEADER_SIZE) return -3;
    hdr->sample_count = rd32(file + 0x00);
    hdr->nibble_count = rd32(file + 0x04);
    hdr->sample_rate = rd32(file + 0x08);
    hdr->looping = rd16(file + 0x0c) == 1;
    hdr->format = rd16(file + 0x0e);
    hdr->start_addr = rd32(file + 0x10);
    hdr->end_addr = rd32(file + 0x14);
    hdr->cur_addr = rd32(file + 0x18);
    hdr->channel_count = rd16(file + 0x4a);
    hdr->frames_per_interleave = rd16(file + 0x4c);
    if (hdr->channel_count == 0) hdr->channel_count = 1;
    int nch = hdr->channel_count;
    if (file_len < HEADER_SIZE * nch) return -3;
    for (int i = 0; i < nch; i++) {
        const uint8_t *h = file + (size_t)HEADER_SIZE * i;
        if (coefs_out) for (int k = 0; k < 16; k++) coefs_out[i * 16 + k] = (int16_t)rd16(h + 0x1c + 2 * k);
        if (gain_out) gain_out[i] = (int16_t)rd16(h + 0x3c);
        if (start_context_out) for (int k = 0; k < 3; k++) start_context_out[i * 3 + k] = (int16_t)rd16(h + 0x3e + 2 * k);
        if (loop_context_out) for (int k = 0; k < 3; k++) loop_context_out[i * 3 + k] = (int16_t)rd16(h + 0x44 + 2 * k);
    }
    int nbytes = vgo_gc_sample_count_to_byte_count(hdr->sample_count);
    if (file_len < HEADER_SIZE + nbytes) return -3;                    /* :91-94 */
    if (vgo_gc_sample_count_to_nibble_count(hdr->sample_count) != hdr->nibble_count) return -3;   /* :96-99 */
    if (hdr->format != 0) return -3;                                   /* :101-104 */
    if (!adpcm_out) return 0;
    const uint8_t *data = file + (size_t)HEADER_SIZE * nch;
    int data_len = file_len - HEADER_SIZE * nch;
    if (nch == 1) {
        memcpy(adpcm_out[0], data, (size_t)nbytes);
    } else {                                                           /* ReadData :107-116 */
        int interleave = hdr->frames_per_interleave * BYTES_PER_FRAME;
        int data_length = next_multiple(nbytes, BYTES_PER_FRAME) * nch;
        if (data_len < data_length || interleave <= 0) return -3;
        vgo_deinterleave(data, data_length, interleave, nch, nbytes, adpcm_out);
    }
    return 0;
}
