/*
 * adx_oracle.c -- CPU restatement of VGAudio's CRI ADX codec.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 * PARITY UNPINNED: the reference has no tests at all for the ADX codec
 * (SURVEY.md 8c).  Pinned here only by literal restatement, hand-derived
 * vectors (tests/test_oracle_adx.py) and encode->decode self-consistency.
 *
 * Follows Codecs/CriAdx/CriAdxCodec.cs and Formats/CriAdx/CriAdxHelpers.cs.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

static inline int16_t clamp16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }
/* (int)double as RyuJIT x64 does it (cvttsd2si): NaN and values outside int32 give 0x80000000.  The plain C cast is
 * undefined there; rawDistance * gain reaches +-2^32 on hostile input (|rawDistance| <= 2^17, gain <= 32767). */
static inline int double_to_int(double d)
{
    if (!(d > -2147483649.0 && d < 2147483648.0)) return (int)0x80000000;
    return (int)d;
}
static inline int clamp4(int v) { return v > 7 ? 7 : (v < -8 ? -8 : v); }
static inline int divide_by_round_up(int value, int divisor) { return (int)ceil((double)value / divisor); }
static inline int divide_by2_round_up(int value) { return (value / 2) + (value & 1); }
static const int8_t SIGNED_NIBBLES[16] = {0, 1, 2, 3, 4, 5, 6, 7, -8, -7, -6, -5, -4, -3, -2, -1};
static inline uint8_t combine_nibbles(int high, int low) { return (uint8_t)((high << 4) | (low & 0xF)); }

/* Utilities/Helpers.cs:146-163 -- floor(log2) through the De Bruijn table */
static int log2_debruijn(int value)
{
    static const int tab[32] = {0, 9, 1, 10, 13, 21, 2, 29, 11, 14, 16, 18, 22, 25, 3, 30,
                                8, 12, 20, 28, 15, 17, 24, 7, 19, 27, 23, 6, 26, 5, 4, 31};
    value |= value >> 1;
    value |= value >> 2;
    value |= value >> 4;
    value |= value >> 8;
    value |= value >> 16;
    return tab[(uint32_t)((uint32_t)value * 0x07C4ACDDU) >> 27];
}

/* CriAdxCodec.cs:186-191 */
static const int16_t FIXED_COEFS[4][2] = {
    {0, 0}, {0x0F00, 0}, {0x1CC0, (int16_t)0xF300}, {0x1880, (int16_t)0xF240}};

void vgo_adx_default_params(vgo_adx_params *p)
{
    /* Codecs/CriAdx/CriAdxParameters.cs:5-12 */
    p->sample_rate = 48000;
    p->highpass_frequency = 500;
    p->frame_size = 18;
    p->version = 4;
    p->history = 0;
    p->padding = 0;
    p->type = 3;
    p->filter = 0;
}

/* CriAdxCodec.cs:173-184 */
void vgo_adx_calculate_coefficients(int highpass_freq, int sample_rate, int16_t coefs[2])
{
    double sqrt2 = sqrt(2);
    double a = sqrt2 - cos(2.0 * M_PI * highpass_freq / sample_rate);
    double b = sqrt2 - 1;
    double c = (a - sqrt((a + b) * (a - b))) / b;

    coefs[0] = (int16_t)double_to_int(c * 8192);
    coefs[1] = (int16_t)double_to_int(c * c * -4096);
}

/* Formats/CriAdx/CriAdxHelpers.cs:7-31 */
int vgo_adx_nibble_count_to_sample_count(int nibble_count, int frame_size)
{
    int nibbles_per_frame = frame_size * 2;
    int samples_per_frame = nibbles_per_frame - 4;
    int frames = nibble_count / nibbles_per_frame;
    int extra_nibbles = nibble_count % nibbles_per_frame;
    int extra_samples = extra_nibbles < 4 ? 0 : extra_nibbles - 4;
    return samples_per_frame * frames + extra_samples;
}
int vgo_adx_sample_count_to_nibble_count(int sample_count, int frame_size)
{
    int nibbles_per_frame = frame_size * 2;
    int samples_per_frame = nibbles_per_frame - 4;
    int frames = sample_count / samples_per_frame;
    int extra_samples = sample_count % samples_per_frame;
    int extra_nibbles = extra_samples == 0 ? 0 : extra_samples + 4;
    return nibbles_per_frame * frames + extra_nibbles;
}
int vgo_adx_sample_count_to_byte_count(int sample_count, int frame_size)
{
    return divide_by2_round_up(vgo_adx_sample_count_to_nibble_count(sample_count, frame_size));
}

/* CriAdxCodec.cs:149-165 */
static int calculate_scale(int max_distance, double *gain, int *scale_to_write, int exponential)
{
    int scale = (max_distance - 1) / 7 + 1;
    if (scale > 0x1000) scale = 0x1000;
    *scale_to_write = scale - 1;

    if (exponential) {
        int power = *scale_to_write == 0 ? 0 : log2_debruijn(*scale_to_write) + 1;
        scale = 1 << power;
        *scale_to_write = 12 - power;
        max_distance = 8 * scale - 1;
    }

    *gain = max_distance == 0 ? 0 : (double)32767 / max_distance;
    return scale;
}

/* CriAdxCodec.cs:167-171 */
static int scale_short_to_nibble(int sample)
{
    int sign = (sample > 0) - (sample < 0);
    sample = (sample + (32767 / 14) * sign) / (32767 / 7);
    return clamp4(sample);
}

/* CriAdxCodec.cs:107-147 */
static void encode_frame(int16_t *pcm, uint8_t *adpcm_out, const int16_t coefs[2],
                         int samples_per_frame, int type, int version)
{
    int max_distance = 0;
    int *adpcm = (int *)calloc((size_t)samples_per_frame, sizeof(int));

    for (int i = 0; i < samples_per_frame; i++) {
        int predicted_sample = (pcm[i + 1] * coefs[0] >> 12) + (pcm[i] * coefs[1] >> 12);
        int distance = pcm[i + 2] - predicted_sample;
        distance = abs((int)clamp16(distance));
        if (distance > max_distance) max_distance = distance;
    }

    double gain;
    int scale_out;
    int scale = calculate_scale(max_distance, &gain, &scale_out, type == 4);

    for (int i = 0; i < samples_per_frame; i++) {
        int predicted_sample = (pcm[i + 1] * coefs[0] >> 12) + (pcm[i] * coefs[1] >> 12);
        int raw_distance = pcm[i + 2] - predicted_sample;
        int scaled_distance = clamp16(double_to_int(raw_distance * gain));

        int adpcm_sample = scale_short_to_nibble(scaled_distance);
        adpcm[i] = adpcm_sample;

        int16_t decoded_distance = clamp16(scale * adpcm_sample);
        if (version == 4)
            predicted_sample = (pcm[i + 1] * coefs[0] + pcm[i] * coefs[1]) >> 12;
        int decoded_sample = decoded_distance + predicted_sample;
        pcm[i + 2] = clamp16(decoded_sample);
    }

    adpcm_out[0] = (uint8_t)((scale_out >> 8) & 0x1f);
    adpcm_out[1] = (uint8_t)scale_out;

    for (int i = 0; i < samples_per_frame / 2; i++)
        adpcm_out[i + 2] = combine_nibbles(adpcm[i * 2], adpcm[i * 2 + 1]);
    free(adpcm);
}

int vgo_adx_encoded_size(int pcm_length, const vgo_adx_params *c)
{
    int sample_count = pcm_length + c->padding;
    int samples_per_frame = (c->frame_size - 2) * 2;
    return divide_by_round_up(sample_count, samples_per_frame) * c->frame_size;
}

/* CriAdxCodec.cs:56-105 */
void vgo_adx_encode(const int16_t *pcm, int pcm_length, vgo_adx_params *c, uint8_t *adpcm_out)
{
    int sample_count = pcm_length + c->padding;
    int samples_per_frame = (c->frame_size - 2) * 2;
    int frame_count = divide_by_round_up(sample_count, samples_per_frame);
    int padding_remaining = c->padding;
    int16_t coefs[2];
    if (c->type == 2) { coefs[0] = FIXED_COEFS[c->filter & 3][0]; coefs[1] = FIXED_COEFS[c->filter & 3][1]; }
    else vgo_adx_calculate_coefficients(500, c->sample_rate, coefs);

    int16_t *pcm_buffer = (int16_t *)calloc((size_t)samples_per_frame + 2, sizeof(int16_t));
    uint8_t *adpcm_buffer = (uint8_t *)calloc((size_t)c->frame_size, 1);
    memset(adpcm_out, 0, (size_t)frame_count * c->frame_size);

    if (c->version == 4 && c->padding == 0 && pcm_length > 0) {
        pcm_buffer[0] = pcm[0];
        pcm_buffer[1] = pcm[0];
        c->history = pcm[0];
    }

    for (int i = 0; i < frame_count; i++) {
        int samples_to_copy = sample_count - i * samples_per_frame;
        if (samples_to_copy > samples_per_frame) samples_to_copy = samples_per_frame;
        int pcm_buffer_start = 2;
        if (padding_remaining != 0) {
            while (padding_remaining > 0 && samples_to_copy > 0) {
                padding_remaining--;
                samples_to_copy--;
                pcm_buffer_start++;
            }
            if (samples_to_copy == 0) continue;
        }
        int src = i * samples_per_frame - c->padding;
        if (src < 0) src = 0;
        memcpy(pcm_buffer + pcm_buffer_start, pcm + src, (size_t)samples_to_copy * sizeof(int16_t));
        memset(pcm_buffer + pcm_buffer_start + samples_to_copy, 0,
               (size_t)(samples_per_frame - samples_to_copy - pcm_buffer_start + 2) * sizeof(int16_t));

        encode_frame(pcm_buffer, adpcm_buffer, coefs, samples_per_frame, c->type, c->version);

        if (c->type == 2) adpcm_buffer[0] |= (uint8_t)(c->filter << 5);

        memcpy(adpcm_out + (size_t)i * c->frame_size, adpcm_buffer, (size_t)c->frame_size);
        pcm_buffer[0] = pcm_buffer[samples_per_frame];
        pcm_buffer[1] = pcm_buffer[samples_per_frame + 1];
    }
    free(pcm_buffer);
    free(adpcm_buffer);
}

/* CriAdxCodec.cs:9-54 */
void vgo_adx_decode(const uint8_t *adpcm, int sample_count, const vgo_adx_params *c, int16_t *pcm)
{
    int samples_per_frame = (c->frame_size - 2) * 2;
    int16_t calc[2];
    const int16_t(*coefs)[2];
    int ncoef;
    if (c->type == 2) { coefs = FIXED_COEFS; ncoef = 4; }
    else {
        vgo_adx_calculate_coefficients(c->highpass_frequency, c->sample_rate, calc);
        coefs = (const int16_t(*)[2])calc; ncoef = 1;
    }
    memset(pcm, 0, (size_t)sample_count * sizeof(int16_t));

    int hist1 = c->history;
    int hist2 = c->history;
    int frame_count = divide_by_round_up(sample_count, samples_per_frame);

    int current_sample = 0;
    int start_sample = c->padding > 0 ? c->padding % samples_per_frame : 0;
    size_t in_index = (size_t)(c->padding / samples_per_frame) * c->frame_size;

    for (int i = 0; i < frame_count; i++) {
        int filter_num = ((adpcm[in_index] >> 4) & 0xF) >> 1;
        /* the reference throws IndexOutOfRange for a filter the coef table lacks */
        if (filter_num >= ncoef) filter_num = ncoef - 1;
        int16_t scale = (int16_t)((adpcm[in_index] << 8 | adpcm[in_index + 1]) & 0x1FFF);
        scale = (int16_t)(c->type == 4 ? 1 << ((12 - scale) & 31) : scale + 1);
        in_index += 2 + start_sample / 2;

        int samples_to_read = sample_count - current_sample;
        if (samples_to_read > samples_per_frame) samples_to_read = samples_per_frame;

        for (int s = start_sample; s < samples_to_read; s++) {
            int sample = s % 2 == 0 ? SIGNED_NIBBLES[(adpcm[in_index] >> 4) & 0xF]
                                    : SIGNED_NIBBLES[adpcm[in_index++] & 0xF];
            if (c->version == 4)
                sample = scale * sample + ((hist1 * coefs[filter_num][0] + hist2 * coefs[filter_num][1]) >> 12);
            else
                sample = scale * sample + (hist1 * coefs[filter_num][0] >> 12) + (hist2 * coefs[filter_num][1] >> 12);

            int16_t final_sample = clamp16(sample);

            hist2 = hist1;
            hist1 = final_sample;
            pcm[current_sample++] = final_sample;
        }
        start_sample = 0;
    }
}

/* ---------------- batch drivers: Formats/CriAdx/CriAdxFormat.cs:57-88, :34-55 ---------------- */
typedef struct {
    const int16_t *pcm; long pitch; int nch; int n; const vgo_adx_params *p;
    uint8_t *out; long out_pitch; int16_t *hist;
    const uint8_t *dec_in; long dec_in_pitch; int16_t *dec_out; long dec_out_pitch;
    int next; pthread_mutex_t mu; int decode;
} adx_job;

static void *adx_worker(void *arg)
{
    adx_job *j = (adx_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int c = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (c >= j->nch) break;
        vgo_adx_params p = *j->p;
        if (!j->decode) {
            vgo_adx_encode(j->pcm + (size_t)c * j->pitch, j->n, &p, j->out + (size_t)c * j->out_pitch);
            if (j->hist) j->hist[c] = p.history;
        } else {
            vgo_adx_decode(j->dec_in + (size_t)c * j->dec_in_pitch, j->n, &p,
                           j->dec_out + (size_t)c * j->dec_out_pitch);
        }
    }
    return NULL;
}

static void adx_run(adx_job *j, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_mutex_init(&j->mu, NULL);
    j->next = 0;
    if (threads == 1) { adx_worker(j); pthread_mutex_destroy(&j->mu); return; }
    pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, adx_worker, j);
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    free(t);
    pthread_mutex_destroy(&j->mu);
}

void vgo_adx_encode_batch(const int16_t *pcm, long pitch, int nch, int pcm_length,
                          const vgo_adx_params *p, uint8_t *out, long out_pitch,
                          int16_t *history_out, int threads)
{
    adx_job j; memset(&j, 0, sizeof j);
    j.pcm = pcm; j.pitch = pitch; j.nch = nch; j.n = pcm_length; j.p = p;
    j.out = out; j.out_pitch = out_pitch; j.hist = history_out; j.decode = 0;
    adx_run(&j, threads);
}

void vgo_adx_decode_batch(const uint8_t *adpcm, long in_pitch, int nch, int sample_count,
                          const vgo_adx_params *p, int16_t *pcm_out, long out_pitch, int threads)
{
    adx_job j; memset(&j, 0, sizeof j);
    j.dec_in = adpcm; j.dec_in_pitch = in_pitch; j.nch = nch; j.n = sample_count; j.p = p;
    j.dec_out = pcm_out; j.dec_out_pitch = out_pitch; j.decode = 1;
    adx_run(&j, threads);
}
