/*
 * gcadpcm_oracle.c -- CPU restatement of VGAudio's GC-ADPCM ("DSP-ADPCM") path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity status: pinned by the
 * reference's size-math KATs and ascending-ramp encode->decode KATs
 * (Tests/Formats/GcAdpcm/GcAdpcmHelpersTests.cs:8-99,
 *  Tests/Formats/GcAdpcmFormatTests.cs:92-158); coefficient values and
 * bitstreams on general audio are NOT pinned by any reference test.
 *
 * Follows, function for function:
 *   Codecs/GcAdpcm/GcAdpcmMath.cs, GcAdpcmCoefficients.cs, GcAdpcmEncoder.cs,
 *   GcAdpcmDecoder.cs, Utilities/Helpers.cs:32-58, Utilities/Extensions.cs:145-146.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define BYTES_PER_FRAME 8
#define SAMPLES_PER_FRAME 14
#define NIBBLES_PER_FRAME 16

/* Utilities/Helpers.cs:32-48 */
static inline int16_t clamp16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }
static inline int clamp4(int v) { return v > 7 ? 7 : (v < -8 ? -8 : v); }
/* Utilities/Extensions.cs:145 -- through f64 Math.Ceiling */
static inline int divide_by_round_up(int value, int divisor) { return (int)ceil((double)value / divisor); }
/* Utilities/Extensions.cs:146 */
static inline int divide_by2_round_up(int value) { return (value / 2) + (value & 1); }
/* Utilities/Helpers.cs:50-58 */
static const int8_t SIGNED_NIBBLES[16] = {0, 1, 2, 3, 4, 5, 6, 7, -8, -7, -6, -5, -4, -3, -2, -1};
static inline uint8_t combine_nibbles(int high, int low) { return (uint8_t)((high << 4) | (low & 0xF)); }

/* ---------------- GcAdpcmMath.cs:11-47 ---------------- */
int vgo_gc_nibble_count_to_sample_count(int nibble_count)
{
    int frames = nibble_count / NIBBLES_PER_FRAME;
    int extra_nibbles = nibble_count % NIBBLES_PER_FRAME;
    int extra_samples = extra_nibbles < 2 ? 0 : extra_nibbles - 2;
    return SAMPLES_PER_FRAME * frames + extra_samples;
}
int vgo_gc_sample_count_to_nibble_count(int sample_count)
{
    int frames = sample_count / SAMPLES_PER_FRAME;
    int extra_samples = sample_count % SAMPLES_PER_FRAME;
    int extra_nibbles = extra_samples == 0 ? 0 : extra_samples + 2;
    return NIBBLES_PER_FRAME * frames + extra_nibbles;
}
int vgo_gc_nibble_to_sample(int nibble)
{
    int frames = nibble / NIBBLES_PER_FRAME;
    int extra_nibbles = nibble % NIBBLES_PER_FRAME;
    int samples = SAMPLES_PER_FRAME * frames;
    return samples + extra_nibbles - 2;
}
int vgo_gc_sample_to_nibble(int sample)
{
    int frames = sample / SAMPLES_PER_FRAME;
    int extra_samples = sample % SAMPLES_PER_FRAME;
    return NIBBLES_PER_FRAME * frames + extra_samples + 2;
}
int vgo_gc_sample_count_to_byte_count(int sample_count)
{
    return divide_by2_round_up(vgo_gc_sample_count_to_nibble_count(sample_count));
}
int vgo_gc_byte_count_to_sample_count(int byte_count)
{
    return vgo_gc_nibble_count_to_sample_count(byte_count * 2);
}

/* ---------------- GcAdpcmCoefficients.cs ---------------- */

/* :112-120 -- products are int (short*short), accumulated in f64 */
static void inner_product_merge(double vec_out[3], const int16_t pcm_buf[28])
{
    for (int i = 0; i <= 2; i++) {
        vec_out[i] = 0.0f;
        for (int x = 0; x < 14; x++)
            vec_out[i] -= (double)((int)pcm_buf[14 + x - i] * (int)pcm_buf[14 + x]);
    }
}

/* :122-131 */
static void outer_product_merge(double mtx_out[3][3], const int16_t pcm_buf[28])
{
    for (int x = 1; x <= 2; x++)
        for (int y = 1; y <= 2; y++) {
            mtx_out[x][y] = 0.0;
            for (int z = 0; z < 14; z++)
                mtx_out[x][y] += (double)((int)pcm_buf[14 + z - x] * (int)pcm_buf[14 + z - y]);
        }
}

/* :133-208.  double.Epsilon is the smallest denormal, so `val < Epsilon`
 * is `val == 0` for the non-negative val here. */
static int analyze_ranges(double mtx[3][3], int vec_idxs_out[3], double recips[3])
{
    double val, tmp, min, max;
    const double dbl_epsilon_cs = 4.9406564584124654e-324;

    for (int x = 1; x <= 2; x++) {
        val = fmax(fabs(mtx[x][1]), fabs(mtx[x][2]));
        if (val < dbl_epsilon_cs)
            return 1;
        recips[x] = 1.0 / val;
    }

    int max_index = 0;
    for (int i = 1; i <= 2; i++) {
        for (int x = 1; x < i; x++) {
            tmp = mtx[x][i];
            for (int y = 1; y < x; y++)
                tmp -= mtx[x][y] * mtx[y][i];
            mtx[x][i] = tmp;
        }

        val = 0.0;
        for (int x = i; x <= 2; x++) {
            tmp = mtx[x][i];
            for (int y = 1; y < i; y++)
                tmp -= mtx[x][y] * mtx[y][i];

            mtx[x][i] = tmp;
            tmp = fabs(tmp) * recips[x];
            if (tmp >= val) {
                val = tmp;
                max_index = x;
            }
        }

        if (max_index != i) {
            for (int y = 1; y <= 2; y++) {
                tmp = mtx[max_index][y];
                mtx[max_index][y] = mtx[i][y];
                mtx[i][y] = tmp;
            }
            recips[max_index] = recips[i];
        }

        vec_idxs_out[i] = max_index;

        if (i != 2) {
            tmp = 1.0 / mtx[i][i];
            for (int x = i + 1; x <= 2; x++)
                mtx[x][i] *= tmp;
        }
    }

    min = 1.0e10;
    max = 0.0;
    for (int i = 1; i <= 2; i++) {
        tmp = fabs(mtx[i][i]);
        if (tmp < min)
            min = tmp;
        if (tmp > max)
            max = tmp;
    }

    return min / max < 1.0e-10;
}

/* :210-237 */
static void bidirectional_filter(double mtx[3][3], const int vec_idxs[3], double vec_out[3])
{
    double tmp;

    for (int i = 1, x = 0; i <= 2; i++) {
        int index = vec_idxs[i];
        tmp = vec_out[index];
        vec_out[index] = vec_out[i];
        if (x != 0) {
            for (int y = x; y <= i - 1; y++)
                tmp -= vec_out[y] * mtx[i][y];
        } else if (tmp != 0.0) {
            x = i;
        }
        vec_out[i] = tmp;
    }

    for (int i = 2; i > 0; i--) {
        tmp = vec_out[i];
        for (int y = i + 1; y <= 2; y++)
            tmp -= vec_out[y] * mtx[i][y];
        vec_out[i] = tmp / mtx[i][i];
    }

    vec_out[0] = 1.0;
}

/* :239-255 */
static int quadratic_merge(double v[3])
{
    double v2 = v[2];
    double tmp = 1.0 - (v2 * v2);

    if (tmp == 0.0)
        return 1;

    double v0 = (v[0] - (v2 * v2)) / tmp;
    double v1 = (v[1] - (v[1] * v2)) / tmp;

    v[0] = v0;
    v[1] = v1;

    return fabs(v1) > 1.0;
}

/* :257-283 (both overloads) */
static void finish_record(double in_r[3], double out_r[3])
{
    for (int z = 1; z <= 2; z++) {
        if (in_r[z] >= 1.0)
            in_r[z] = 0.9999999999;
        else if (in_r[z] <= -1.0)
            in_r[z] = -0.9999999999;
    }
    out_r[0] = 1.0;
    out_r[1] = (in_r[2] * in_r[1]) + in_r[1];
    out_r[2] = in_r[2];
}

/* :285-305 */
static void matrix_filter(const double *src_row, double dst[3], double mtx[3][3])
{
    mtx[2][0] = 1.0;
    for (int i = 1; i <= 2; i++)
        mtx[2][i] = -src_row[i];

    for (int i = 2; i > 0; i--) {
        double val = 1.0 - (mtx[i][i] * mtx[i][i]);
        for (int y = 1; y <= i; y++)
            mtx[i - 1][y] = ((mtx[i][i] * mtx[i][y]) + mtx[i][y]) / val;
    }

    dst[0] = 1.0;
    for (int i = 1; i <= 2; i++) {
        dst[i] = 0.0;
        for (int y = 1; y <= i; y++)
            dst[i] += mtx[i][y] * dst[i - y];
    }
}

/* :307-333 */
static void merge_finish_record(const double src[3], double dst[3])
{
    double tmp[3] = {0.0, 0.0, 0.0};
    double val = src[0];

    dst[0] = 1.0;
    for (int i = 1; i <= 2; i++) {
        double v2 = 0.0;
        for (int y = 1; y < i; y++)
            v2 += dst[y] * src[i - y];

        if (val > 0.0)
            dst[i] = -(v2 + src[i]) / val;
        else
            dst[i] = 0.0;

        tmp[i] = dst[i];

        for (int y = 1; y < i; y++)
            dst[y] += dst[i] * dst[i - y];

        val *= 1.0 - (dst[i] * dst[i]);
    }

    finish_record(tmp, dst);
}

/* :335-342 */
static double contrast_vectors(const double s1[3], const double *rec)
{
    double val = (rec[2] * rec[1] + -rec[1]) / (1.0 - rec[2] * rec[2]);
    double val1 = (s1[0] * s1[0]) + (s1[1] * s1[1]) + (s1[2] * s1[2]);
    double val2 = (s1[0] * s1[1]) + (s1[1] * s1[2]);
    double val3 = s1[0] * s1[2];
    return val1 + (2.0 * val * val2) + (2.0 * (-rec[1] * val + -rec[2]) * val3);
}

/* :344-396 */
static void filter_records(double vec_best[8][3], int exp, const double *records, int record_count)
{
    double buffer_list[8][3];
    double mtx[3][3];
    int buffer1[8];
    double buffer2[3];

    memset(buffer_list, 0, sizeof buffer_list);
    memset(mtx, 0, sizeof mtx);
    memset(buffer1, 0, sizeof buffer1);

    for (int x = 0; x < 2; x++) {
        for (int y = 0; y < exp; y++) {
            buffer1[y] = 0;
            for (int i = 0; i <= 2; i++)
                buffer_list[y][i] = 0.0;
        }
        for (int z = 0; z < record_count; z++) {
            int index = 0;
            double value = 1.0e30;
            for (int i = 0; i < exp; i++) {
                double temp_val = contrast_vectors(vec_best[i], records + 3 * (size_t)z);
                if (temp_val < value) {
                    value = temp_val;
                    index = i;
                }
            }
            buffer1[index]++;
            matrix_filter(records + 3 * (size_t)z, buffer2, mtx);
            for (int i = 0; i <= 2; i++)
                buffer_list[index][i] += buffer2[i];
        }

        for (int i = 0; i < exp; i++)
            if (buffer1[i] > 0)
                for (int y = 0; y <= 2; y++)
                    buffer_list[i][y] /= buffer1[i];

        for (int i = 0; i < exp; i++)
            merge_finish_record(buffer_list[i], vec_best[i]);
    }
}

/* (short)Math.Round(d) after the range checks of :99-107.  Math.Round is
 * ties-to-even.  RyuJIT x64 converts NaN through cvttsd2si -> 0x80000000 ->
 * low 16 bits 0; NaN cannot reach here for finite records. */
static int16_t round_to_short(double d)
{
    if (d != d) return 0;
    return (int16_t)nearbyint(d);
}

/* :9-110 */
void vgo_gc_calculate_coefficients(const int16_t *source, int length, int16_t coefs[16])
{
    int frame_count = divide_by_round_up(length, SAMPLES_PER_FRAME);

    int16_t pcm_hist_buffer[28];
    double vec1[3] = {0, 0, 0}, vec2[3] = {0, 0, 0}, buffer[3] = {0, 0, 0};
    double mtx[3][3];
    int vec_idxs[3] = {0, 0, 0};
    double vec_best[8][3];
    int record_count = 0;

    memset(pcm_hist_buffer, 0, sizeof pcm_hist_buffer);
    memset(mtx, 0, sizeof mtx);
    memset(vec_best, 0, sizeof vec_best);
    memset(coefs, 0, 16 * sizeof(int16_t));

    double *records = (double *)calloc((size_t)(frame_count > 0 ? frame_count : 1) * 2 * 3, sizeof(double));

    for (int sample = 0, remaining = length; sample < length; sample += 14, remaining -= 14) {
        memset(pcm_hist_buffer + 14, 0, 14 * sizeof(int16_t));
        memcpy(pcm_hist_buffer + 14, source + sample, (size_t)(remaining < 14 ? remaining : 14) * sizeof(int16_t));

        inner_product_merge(vec1, pcm_hist_buffer);
        if (fabs(vec1[0]) > 10.0) {
            outer_product_merge(mtx, pcm_hist_buffer);
            if (!analyze_ranges(mtx, vec_idxs, buffer)) {
                bidirectional_filter(mtx, vec_idxs, vec1);
                if (!quadratic_merge(vec1)) {
                    finish_record(vec1, records + 3 * (size_t)record_count);
                    record_count++;
                }
            }
        }

        memmove(pcm_hist_buffer, pcm_hist_buffer + 14, 14 * sizeof(int16_t));
    }

    vec1[0] = 1.0;
    vec1[1] = 0.0;
    vec1[2] = 0.0;

    for (int z = 0; z < record_count; z++) {
        matrix_filter(records + 3 * (size_t)z, vec_best[0], mtx);
        for (int y = 1; y <= 2; y++)
            vec1[y] += vec_best[0][y];
    }
    for (int y = 1; y <= 2; y++)
        vec1[y] /= record_count;

    merge_finish_record(vec1, vec_best[0]);

    int exp = 1;
    for (int w = 0; w < 3;) {
        vec2[0] = 0.0;
        vec2[1] = -1.0;
        vec2[2] = 0.0;
        for (int i = 0; i < exp; i++)
            for (int y = 0; y <= 2; y++)
                vec_best[exp + i][y] = (0.01 * vec2[y]) + vec_best[i][y];
        ++w;
        exp = 1 << w;
        filter_records(vec_best, exp, records, record_count);
    }

    for (int z = 0; z < 8; z++) {
        double d;
        d = -vec_best[z][1] * 2048.0;
        if (d > 0.0)
            coefs[z * 2] = (d > 32767) ? 32767 : round_to_short(d);
        else
            coefs[z * 2] = (d < -32768) ? -32768 : round_to_short(d);

        d = -vec_best[z][2] * 2048.0;
        if (d > 0.0)
            coefs[z * 2 + 1] = (d > 32767) ? 32767 : round_to_short(d);
        else
            coefs[z * 2 + 1] = (d < -32768) ? -32768 : round_to_short(d);
    }
    free(records);
}

/* ---------------- GcAdpcmEncoder.cs ---------------- */

static __thread uint64_t g_trip_hist[16];
static __thread int g_nonterminating;

int vgo_gc_last_encode_hit_nontermination(void) { return g_nonterminating; }

void vgo_gc_trip_histogram(uint64_t hist_out[16]) { memcpy(hist_out, g_trip_hist, sizeof g_trip_hist); }

/* :96-171 -- all int arithmetic is unchecked C# int32 (wraps; -fwrapv) */
static void dsp_encode_coef(const int16_t pcm_in[16], int sample_count, const int16_t coefs[2],
                            int pcm_out[16], int adpcm_out[14], int *scale_power_out,
                            double *total_distance_out)
{
    int max_overflow;
    int max_distance = 0;
    int scale_power;
    double total_distance;
    int trips = 0;

    pcm_out[0] = pcm_in[0];
    pcm_out[1] = pcm_in[1];

    for (int s = 0; s < sample_count; s++) {
        int input_sample = pcm_in[s + 2];
        int predicted_sample = (pcm_in[s] * coefs[1] + pcm_in[s + 1] * coefs[0]) / 2048;
        int distance = input_sample - predicted_sample;
        distance = clamp16(distance);
        if (abs(distance) > abs(max_distance))
            max_distance = distance;
    }

    scale_power = 0;
    while (scale_power <= 12 && (max_distance > 7 || max_distance < -8)) {
        max_distance /= 2;
        scale_power++;
    }
    scale_power = scale_power <= 1 ? -1 : scale_power - 2;

    do {
        scale_power++;
        trips++;
        int scale = (1 << scale_power) * 2048;
        total_distance = 0;
        max_overflow = 0;

        for (int s = 0; s < sample_count; s++) {
            int input_sample = pcm_in[s + 2] * 2048;
            int predicted_sample = pcm_out[s] * coefs[1] + pcm_out[s + 1] * coefs[0];
            int distance = input_sample - predicted_sample;
            int unclamped_adpcm_sample = (distance > 0)
                ? (int)((double)((float)distance / (float)scale) + (double)0.4999999f)
                : (int)((double)((float)distance / (float)scale) - (double)0.4999999f);

            int adpcm_sample = clamp4(unclamped_adpcm_sample);
            if (adpcm_sample != unclamped_adpcm_sample) {
                int overflow = abs(unclamped_adpcm_sample - adpcm_sample);
                if (overflow > max_overflow) max_overflow = overflow;
            }

            adpcm_out[s] = adpcm_sample;

            int decoded_distance = adpcm_sample * scale;
            int corrected_sample = predicted_sample + decoded_distance;
            int scaled_sample = (corrected_sample + 1024) >> 11;
            pcm_out[s + 2] = clamp16(scaled_sample);
            double actual_distance = pcm_in[s + 2] - pcm_out[s + 2];
            total_distance += actual_distance * actual_distance;
        }

        for (int x = max_overflow + 8; x > 256; x >>= 1)
            if (++scale_power >= 12)
                scale_power = 11;

        /* NON-TERMINATION HAZARD of the reference: when the pass at scalePower 12 still has
           maxOverflow > 248, the bump loop above resets scalePower to 11 and the identical pass
           repeats forever (reachable only when hostile coefs make the int32 predictor wrap; not
           with CalculateCoefficients output, whose |c0| < 4096, |c1| < 2048).  The oracle stops
           after the first pass at 12 and flags it; the HIP kernel has the same guard. */
        if (scale_power < 12 && max_overflow > 1 && scale == (1 << 12) * 2048) {
            g_nonterminating = 1;
            break;
        }
    } while (scale_power < 12 && max_overflow > 1);

    g_trip_hist[trips < 15 ? trips : 15]++;
    *scale_power_out = scale_power;
    *total_distance_out = total_distance;
}

/* :48-94 */
void vgo_gc_encode_frame(int16_t pcm_inout[16], int sample_count, uint8_t adpcm_out[8],
                         const int16_t coefs_in[16])
{
    int16_t coefs[8][2];
    int pcm_out[8][16];
    int adpcm[8][14];
    int scale[8];
    double total_distance[8];

    memset(pcm_out, 0, sizeof pcm_out);
    memset(adpcm, 0, sizeof adpcm);

    for (int i = 0; i < 8; i++) {
        coefs[i][0] = coefs_in[i * 2];
        coefs[i][1] = coefs_in[i * 2 + 1];
    }

    for (int i = 0; i < 8; i++)
        dsp_encode_coef(pcm_inout, sample_count, coefs[i], pcm_out[i], adpcm[i], &scale[i],
                        &total_distance[i]);

    int best_coef = 0;
    double min = 1.7976931348623157e308; /* double.MaxValue */
    for (int i = 0; i < 8; i++) {
        if (total_distance[i] < min) {
            min = total_distance[i];
            best_coef = i;
        }
    }

    for (int s = 0; s < sample_count; s++)
        pcm_inout[s + 2] = (int16_t)pcm_out[best_coef][s + 2];

    adpcm_out[0] = combine_nibbles(best_coef, scale[best_coef]);

    for (int s = sample_count; s < 14; s++)
        adpcm[best_coef][s] = 0;

    for (int i = 0; i < 7; i++)
        adpcm_out[i + 1] = combine_nibbles(adpcm[best_coef][i * 2], adpcm[best_coef][i * 2 + 1]);
}

/* :14-46 */
int vgo_gc_encode(const int16_t *pcm, int pcm_length, const int16_t coefs[16], int sample_count,
                  int16_t hist1, int16_t hist2, uint8_t *adpcm)
{
    if (sample_count == -1) sample_count = pcm_length;
    if (sample_count > pcm_length || sample_count < 0) return -1;
    memset(g_trip_hist, 0, sizeof g_trip_hist);
    g_nonterminating = 0;

    int16_t pcm_buffer[2 + SAMPLES_PER_FRAME];
    uint8_t adpcm_buffer[BYTES_PER_FRAME];

    pcm_buffer[0] = hist2;
    pcm_buffer[1] = hist1;

    int frame_count = divide_by_round_up(sample_count, SAMPLES_PER_FRAME);

    for (int frame = 0; frame < frame_count; frame++) {
        int samples_to_copy = sample_count - frame * SAMPLES_PER_FRAME;
        if (samples_to_copy > SAMPLES_PER_FRAME) samples_to_copy = SAMPLES_PER_FRAME;
        memcpy(pcm_buffer + 2, pcm + (size_t)frame * SAMPLES_PER_FRAME, (size_t)samples_to_copy * sizeof(int16_t));
        memset(pcm_buffer + 2 + samples_to_copy, 0, (size_t)(SAMPLES_PER_FRAME - samples_to_copy) * sizeof(int16_t));

        vgo_gc_encode_frame(pcm_buffer, SAMPLES_PER_FRAME, adpcm_buffer, coefs);

        memcpy(adpcm + (size_t)frame * BYTES_PER_FRAME, adpcm_buffer,
               (size_t)vgo_gc_sample_count_to_byte_count(samples_to_copy));

        pcm_buffer[0] = pcm_buffer[14];
        pcm_buffer[1] = pcm_buffer[15];
    }
    return 0;
}

/* ---------------- GcAdpcmDecoder.cs:10-54 ---------------- */
void vgo_gc_decode(const uint8_t *adpcm, const int16_t coefficients[16], int sample_count,
                   int16_t h1, int16_t h2, int16_t *pcm)
{
    if (sample_count == 0) return;

    int frame_count = divide_by_round_up(sample_count, SAMPLES_PER_FRAME);
    int current_sample = 0;
    size_t out_index = 0;
    size_t in_index = 0;
    int16_t hist1 = h1;
    int16_t hist2 = h2;

    for (int i = 0; i < frame_count; i++) {
        uint8_t predictor_scale = adpcm[in_index++];
        int scale = (1 << (predictor_scale & 0xF)) * 2048;
        int predictor = (predictor_scale >> 4) & 0xF;
        /* the reference indexes coefficients[predictor*2] of a 16-entry array and would
           throw for predictor > 7; the oracle masks to stay in bounds. */
        int16_t coef1 = coefficients[(predictor & 7) * 2];
        int16_t coef2 = coefficients[(predictor & 7) * 2 + 1];

        int samples_to_read = sample_count - current_sample;
        if (samples_to_read > SAMPLES_PER_FRAME) samples_to_read = SAMPLES_PER_FRAME;

        for (int s = 0; s < samples_to_read; s++) {
            int adpcm_sample = s % 2 == 0 ? SIGNED_NIBBLES[(adpcm[in_index] >> 4) & 0xF]
                                          : SIGNED_NIBBLES[adpcm[in_index++] & 0xF];
            int distance = scale * adpcm_sample;
            int predicted_sample = coef1 * hist1 + coef2 * hist2;
            int corrected_sample = predicted_sample + distance;
            int scaled_sample = (corrected_sample + 1024) >> 11;
            int16_t clamped_sample = clamp16(scaled_sample);

            hist2 = hist1;
            hist1 = clamped_sample;

            pcm[out_index++] = clamped_sample;
            current_sample++;
        }
    }
}

/* Formats/GcAdpcm/GcAdpcmSeekTable.cs:25-38 */
void vgo_gc_create_seek_table(const int16_t *pcm, int n, int samples_per_entry, int16_t *table_out)
{
    int entry_count = divide_by_round_up(n, samples_per_entry);
    memset(table_out, 0, (size_t)entry_count * 2 * sizeof(int16_t));
    for (int i = 1; i < entry_count; i++) {
        table_out[i * 2] = pcm[i * samples_per_entry - 1];
        table_out[i * 2 + 1] = pcm[i * samples_per_entry - 2];
    }
}

/* ---------------- channel metadata (Formats/GcAdpcm) ---------------- */
static int get_next_multiple_i(int value, int multiple)      /* Utilities/Helpers.cs:71-80 */
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}

/* Utilities/Helpers.cs:82-83 */
static int loop_points_are_aligned(int loop_start, int multiple)
{
    return !(multiple != 0 && loop_start % multiple != 0);
}

int vgo_gc_channel_layout_for(const vgo_gc_channel_params *p, vgo_gc_channel_layout *out)
{
    memset(out, 0, sizeof *out);
    out->alignment_needed = !loop_points_are_aligned(p->loop_start, p->loop_alignment_multiple);
    out->loop_start_aligned = p->loop_start;
    out->sample_count_aligned = p->sample_count;
    if (out->alignment_needed) {                                 /* GcAdpcmAlignment.cs:29-31 */
        out->loop_start_aligned = get_next_multiple_i(p->loop_start, p->loop_alignment_multiple);
        out->sample_count_aligned = p->loop_end + (out->loop_start_aligned - p->loop_start);
    }
    if (p->samples_per_seek_table_entry != 0)                    /* GcAdpcmSeekTable.cs:27 */
        out->seek_table_entries = divide_by_round_up(out->sample_count_aligned, p->samples_per_seek_table_entry);
    return 0;
}

/* GcAdpcmAlignment.cs:20-63 */
int vgo_gc_alignment(int multiple, int loop_start, int loop_end, const uint8_t *adpcm, const int16_t coefs[16],
                     vgo_gc_channel_layout *L, uint8_t *adpcm_aligned, int16_t *pcm_aligned)
{
    vgo_gc_channel_params p = {loop_end, 1, loop_start, loop_end, multiple, 0};
    vgo_gc_channel_layout_for(&p, L);
    if (!L->alignment_needed) return 0;

    int loop_length = loop_end - loop_start;
    int sample_count_aligned = L->sample_count_aligned;
    int frames_to_keep = loop_end / SAMPLES_PER_FRAME;
    int bytes_to_keep = frames_to_keep * BYTES_PER_FRAME;
    int samples_to_keep = frames_to_keep * SAMPLES_PER_FRAME;
    int samples_to_encode = sample_count_aligned - samples_to_keep;
    if (loop_length <= 0 && loop_end - samples_to_keep < samples_to_encode) return -4;   /* :47 would spin */

    int16_t *old_pcm = (int16_t *)calloc((size_t)(loop_end > 0 ? loop_end : 1), sizeof(int16_t));
    vgo_gc_decode(adpcm, coefs, loop_end, 0, 0, old_pcm);                       /* :41-42 */
    memcpy(pcm_aligned, old_pcm, (size_t)loop_end * sizeof(int16_t));           /* :43 */
    int16_t *new_pcm = (int16_t *)calloc((size_t)(samples_to_encode > 0 ? samples_to_encode : 1), sizeof(int16_t));
    memcpy(new_pcm, old_pcm + samples_to_keep, (size_t)(loop_end - samples_to_keep) * sizeof(int16_t));   /* :46 */
    for (int cur = loop_end - samples_to_keep; cur < samples_to_encode; cur += loop_length) {              /* :48-51 */
        int len = samples_to_encode - cur < loop_length ? samples_to_encode - cur : loop_length;
        memcpy(new_pcm + cur, pcm_aligned + loop_start, (size_t)len * sizeof(int16_t));
    }
    int16_t h1 = samples_to_keep < 1 ? 0 : old_pcm[samples_to_keep - 1];       /* :54-55 */
    int16_t h2 = samples_to_keep < 2 ? 0 : old_pcm[samples_to_keep - 2];

    int new_bytes = vgo_gc_sample_count_to_byte_count(samples_to_encode);
    uint8_t *new_adpcm = (uint8_t *)calloc((size_t)(new_bytes > 0 ? new_bytes : 1), 1);
    vgo_gc_encode(new_pcm, samples_to_encode, coefs, samples_to_encode, h1, h2, new_adpcm);   /* :57 */
    memcpy(adpcm_aligned, adpcm, (size_t)bytes_to_keep);                                       /* :58 */
    memcpy(adpcm_aligned + bytes_to_keep, new_adpcm, (size_t)new_bytes);                       /* :59 */

    int16_t *decoded = (int16_t *)calloc((size_t)(samples_to_encode > 0 ? samples_to_encode : 1), sizeof(int16_t));
    vgo_gc_decode(new_adpcm, coefs, samples_to_encode, h1, h2, decoded);                       /* :61 */
    memcpy(pcm_aligned + samples_to_keep, decoded, (size_t)samples_to_encode * sizeof(int16_t));   /* :62 */
    free(old_pcm); free(new_pcm); free(new_adpcm); free(decoded);
    return 0;
}

/* GcAdpcmLoopContext.cs:17-26, GcAdpcmDecoder.GetPredictorScale (GcAdpcmDecoder.cs:56-59) */
void vgo_gc_loop_context(const uint8_t *adpcm, const int16_t *pcm, int loop_start, int16_t out[3])
{
    out[0] = (int16_t)adpcm[loop_start / SAMPLES_PER_FRAME * BYTES_PER_FRAME];
    out[1] = loop_start < 1 ? 0 : (pcm ? pcm[loop_start - 1] : 0);
    out[2] = loop_start < 2 ? 0 : (pcm ? pcm[loop_start - 2] : 0);
}

/* GcAdpcmChannel.cs:31-55 with GcAdpcmChannelBuilder.cs:148-202 for a channel that has no previous
 * alignment / loop context / seek table (GetCloneBuilder of a 3-argument GcAdpcmChannel, :62-95) */
int vgo_gc_build_channel(const uint8_t *adpcm, const int16_t coefs[16], const vgo_gc_channel_params *p,
                         vgo_gc_channel_layout *layout_out, uint8_t *adpcm_out, int16_t *pcm_out,
                         int16_t *seek_table_out, int16_t loop_context_out[3])
{
    vgo_gc_channel_layout L;
    vgo_gc_channel_layout_for(p, &L);
    if (layout_out) *layout_out = L;
    int rc = 0;
    int nbytes = vgo_gc_sample_count_to_byte_count(L.sample_count_aligned);
    uint8_t *aligned_adpcm = (uint8_t *)calloc((size_t)(nbytes > 0 ? nbytes : 1), 1);
    int16_t *aligned_pcm = (int16_t *)calloc((size_t)(L.sample_count_aligned > 0 ? L.sample_count_aligned : 1), sizeof(int16_t));
    int have_pcm = 0;
    if (L.alignment_needed) {                                    /* GetAlignment :148-165 */
        vgo_gc_channel_layout tmp;
        rc = vgo_gc_alignment(p->loop_alignment_multiple, p->loop_start, p->loop_end, adpcm, coefs, &tmp, aligned_adpcm,
                              aligned_pcm);
        have_pcm = 1;
    } else {
        memcpy(aligned_adpcm, adpcm, (size_t)nbytes);
    }
    if (!rc) {
        /* GetLoopContext :167-181: LoopContextStart defaults to 0, so a loop start of 0 takes the
         * "current loop context is valid" branch and yields the default context (0, 0, 0) */
        int16_t ctx[3] = {0, 0, 0};
        if (L.loop_start_aligned != 0) {
            if (!have_pcm) {                                     /* EnsurePcmDecoded :202 */
                vgo_gc_decode(aligned_adpcm, coefs, L.sample_count_aligned, 0, 0, aligned_pcm);
                have_pcm = 1;
            }
            /* :179 passes Adpcm (the ORIGINAL stream), AlignedPcm, AlignedLoopStart */
            if (L.loop_start_aligned / SAMPLES_PER_FRAME * BYTES_PER_FRAME >= vgo_gc_sample_count_to_byte_count(p->sample_count))
                rc = -2;
            else
                vgo_gc_loop_context(adpcm, aligned_pcm, L.loop_start_aligned, ctx);
        }
        if (loop_context_out) memcpy(loop_context_out, ctx, sizeof ctx);
    }
    if (!rc && p->samples_per_seek_table_entry != 0) {           /* GetSeekTable :183-200 */
        if (!have_pcm) {
            vgo_gc_decode(aligned_adpcm, coefs, L.sample_count_aligned, 0, 0, aligned_pcm);
            have_pcm = 1;
        }
        if (seek_table_out)
            vgo_gc_create_seek_table(aligned_pcm, L.sample_count_aligned, p->samples_per_seek_table_entry, seek_table_out);
    }
    if (!rc) {
        if (adpcm_out) memcpy(adpcm_out, aligned_adpcm, (size_t)nbytes);
        if (pcm_out) {                                           /* GetPcmAudio (GcAdpcmChannel.cs:57-60) */
            if (!have_pcm) vgo_gc_decode(aligned_adpcm, coefs, L.sample_count_aligned, 0, 0, aligned_pcm);
            memcpy(pcm_out, aligned_pcm, (size_t)L.sample_count_aligned * sizeof(int16_t));
        }
    }
    free(aligned_adpcm); free(aligned_pcm);
    return rc;
}

/* ---------------- batch drivers (GcAdpcmFormat.cs:58-74,129-135; :42-54) ---------------- */
typedef struct {
    const int16_t *pcm; long pitch; int nch; int n;
    int16_t *coefs; uint8_t *adpcm; long out_pitch;
    const uint8_t *dec_in; long dec_in_pitch; int16_t *dec_out; long dec_out_pitch;
    int next; pthread_mutex_t mu; int decode;
} gc_job;

static void *gc_worker(void *arg)
{
    gc_job *j = (gc_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int c = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (c >= j->nch) break;
        if (!j->decode) {
            /* EncodeChannel, GcAdpcmFormat.cs:129-135 */
            const int16_t *p = j->pcm + (size_t)c * j->pitch;
            vgo_gc_calculate_coefficients(p, j->n, j->coefs + 16 * (size_t)c);
            vgo_gc_encode(p, j->n, j->coefs + 16 * (size_t)c, -1, 0, 0, j->adpcm + (size_t)c * j->out_pitch);
        } else {
            vgo_gc_decode(j->dec_in + (size_t)c * j->dec_in_pitch, j->coefs + 16 * (size_t)c, j->n, 0, 0,
                          j->dec_out + (size_t)c * j->dec_out_pitch);
        }
    }
    return NULL;
}

static void gc_run(gc_job *j, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_mutex_init(&j->mu, NULL);
    j->next = 0;
    if (threads == 1) { gc_worker(j); pthread_mutex_destroy(&j->mu); return; }
    pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, gc_worker, j);
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    free(t);
    pthread_mutex_destroy(&j->mu);
}

void vgo_gc_encode_batch(const int16_t *pcm, long pitch, int nch, int sample_count,
                         int16_t *coefs_out, uint8_t *adpcm_out, long out_pitch, int threads)
{
    gc_job j; memset(&j, 0, sizeof j);
    j.pcm = pcm; j.pitch = pitch; j.nch = nch; j.n = sample_count;
    j.coefs = coefs_out; j.adpcm = adpcm_out; j.out_pitch = out_pitch; j.decode = 0;
    gc_run(&j, threads);
}

void vgo_gc_decode_batch(const uint8_t *adpcm, long in_pitch, const int16_t *coefs, int nch,
                         int sample_count, int16_t *pcm_out, long out_pitch, int threads)
{
    gc_job j; memset(&j, 0, sizeof j);
    j.dec_in = adpcm; j.dec_in_pitch = in_pitch; j.coefs = (int16_t *)coefs; j.nch = nch; j.n = sample_count;
    j.dec_out = pcm_out; j.dec_out_pitch = out_pitch; j.decode = 1;
    gc_run(&j, threads);
}
