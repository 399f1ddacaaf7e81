/*
 * hca_oracle.c -- CPU restatement of VGAudio's CRI HCA encoder and decoder.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY STATUS: the reference's tests pin only the constant tables (CriHcaTableTests.cs:8-115,
 * MdctTests.cs:18-59) -- checked in tests/test_oracle_hca.py against tests/golden/hca_tables.json.
 * Encoder/decoder outputs, RunMdct/RunImdct, BitWriter and Crc16 have NO reference tests
 * (SURVEY.md 8c): for those parity is UNPINNED and rests on this literal restatement plus
 * hand-derivable vectors and invariants (CRC check value, pack/unpack round trip, used-bits
 * bound, decode(encode(x)) ~ x).
 *
 * Follows: Codecs/CriHca/{CriHcaEncoder,CriHcaDecoder,CriHcaPacking,CriHcaTables,CriHcaFrame,
 * CriHcaChannel,HcaInfo}.cs, Utilities/{Mdct,BitWriter,BitReader,Crc16,Helpers}.cs,
 * Formats/CriHca/CriHcaFormat.cs:26-84.
 */
#include "oracle.h"
#include "hca_tables_data.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define SUBFRAMES 8
#define SPSF 128          /* SamplesPerSubFrame */
#define SPF 1024          /* SamplesPerFrame */

/* ------------------------------------------------------------------ tables */
static double T_DequantizerScaling[64], T_QuantizerStepSize[16], T_QuantizerDeadZone[16];
static double T_QuantizerScaling[64], T_QuantizerInverseStepSize[16];
static double T_IntensityRatio[15], T_IntensityRatioBounds[14], T_ScaleConversion[128];
static double T_MdctWindow[128];
static double T_Sin[8][128], T_Cos[8][128];
static uint16_t T_Crc[256];
static pthread_once_t g_tables_once = PTHREAD_ONCE_INIT;

static double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

static void init_tables(void)
{
    for (int i = 0; i < 64; i++) {
        T_DequantizerScaling[i] = bits_to_double(HCA_DequantizerScalingTableBits[i]);
        T_QuantizerScaling[i] = bits_to_double(HCA_QuantizerScalingTableBits[i]);
    }
    for (int i = 0; i < 16; i++) {
        T_QuantizerStepSize[i] = bits_to_double(HCA_QuantizerStepSizeBits[i]);
        T_QuantizerInverseStepSize[i] = bits_to_double(HCA_QuantizerInverseStepSizeBits[i]);
        /* CriHcaTables.cs:68-78: boundary's bit pattern minus `steps` */
        int steps = HCA_ResolutionMaxValue[i] + 1;
        double boundary = T_QuantizerStepSize[i] / 2;
        T_QuantizerDeadZone[i] = bits_to_double((uint64_t)((int64_t)double_to_bits(boundary) - steps));
    }
    for (int i = 0; i < 15; i++) T_IntensityRatio[i] = bits_to_double(HCA_IntensityRatioTableBits[i]);
    for (int i = 0; i < 14; i++) T_IntensityRatioBounds[i] = bits_to_double(HCA_IntensityRatioBoundsTableBits[i]);
    for (int i = 0; i < 128; i++) {
        T_ScaleConversion[i] = bits_to_double(HCA_ScaleConversionTableBits[i]);
        float f;
        uint32_t u = HCA_MdctWindowF32Bits[i];
        memcpy(&f, &u, 4);
        T_MdctWindow[i] = (double)f;
    }
    for (int b = 0; b < 8; b++)
        for (int i = 0; i < (1 << b); i++) {
            T_Sin[b][i] = bits_to_double(MDCT_SinBits[(1 << b) - 1 + i]);
            T_Cos[b][i] = bits_to_double(MDCT_CosBits[(1 << b) - 1 + i]);
        }
    /* Utilities/Crc16.cs:20-38, polynomial 0x8005 */
    for (int i = 0; i < 256; i++) {
        uint16_t cur = (uint16_t)(i << 8);
        for (int j = 0; j < 8; j++) {
            int x = (cur & 0x8000) != 0;
            cur = (uint16_t)(cur << 1);
            if (x) cur ^= 0x8005;
        }
        T_Crc[i] = cur;
    }
}

static void ensure_tables(void) { pthread_once(&g_tables_once, init_tables); }

/* table accessors for the tests */
int vgo_hca_table(const char *name, double *out, int cap)
{
    ensure_tables();
    struct { const char *n; const double *p; int len; } t[] = {
        {"DequantizerScalingTable", T_DequantizerScaling, 64}, {"QuantizerStepSize", T_QuantizerStepSize, 16},
        {"QuantizerDeadZone", T_QuantizerDeadZone, 16}, {"QuantizerScalingTable", T_QuantizerScaling, 64},
        {"QuantizerInverseStepSize", T_QuantizerInverseStepSize, 16}, {"IntensityRatioTable", T_IntensityRatio, 15},
        {"IntensityRatioBoundsTable", T_IntensityRatioBounds, 14}, {"ScaleConversionTable", T_ScaleConversion, 128},
        {"MdctWindow", T_MdctWindow, 128}};
    for (unsigned i = 0; i < sizeof t / sizeof t[0]; i++)
        if (!strcmp(name, t[i].n)) {
            int n = t[i].len < cap ? t[i].len : cap;
            memcpy(out, t[i].p, (size_t)n * sizeof(double));
            return t[i].len;
        }
    return -1;
}

/* Utilities/Crc16.cs:12-18 */
uint16_t vgo_crc16(const uint8_t *data, int size)
{
    ensure_tables();
    uint16_t crc = 0;
    for (int i = 0; i < size; i++)
        crc = (uint16_t)((crc << 8) ^ T_Crc[(crc >> 8) ^ data[i]]);
    return crc;
}

/* ------------------------------------------------------------------ helpers */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int16_t clamp16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : (int16_t)v); }
static inline int divide_by_round_up(int value, int divisor) { return (int)ceil((double)value / divisor); }
static inline int get_next_multiple(int value, int multiple)
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}
/* (int)double as RyuJIT x64 does it (cvttsd2si): out of range / NaN -> 0x80000000 */
static inline int double_to_int(double d)
{
    if (!(d > -2147483649.0 && d < 2147483648.0)) return (int)0x80000000;
    return (int)d;
}
/* Math.Round(double): ties to even */
static inline double cs_round(double d) { return nearbyint(d); }

/* ------------------------------------------------------------------ Mdct (Utilities/Mdct.cs) */
typedef struct {
    double mdct_previous[SPSF];
    double imdct_previous[SPSF];
} mdct_state;

/* Mdct.cs:126-181, MdctBits = 7, Scale = sqrt(2/128) (CriHcaChannel.cs:19) */
static double g_mdct_scale;
static pthread_once_t g_scale_once = PTHREAD_ONCE_INIT;
static void init_scale(void) { g_mdct_scale = sqrt(2.0 / SPSF); }

static void dct4(const double *input, double *output)
{
    double tmp[SPSF];
    const int size = SPSF, last = size - 1, half = size / 2;
    const double *sin_t = T_Sin[7], *cos_t = T_Cos[7];
    for (int i = 0; i < half; i++) {
        int i2 = i * 2;
        double a = input[i2];
        double b = input[last - i2];
        double s = sin_t[i], c = cos_t[i];
        tmp[i2] = a * c + b * s;
        tmp[i2 + 1] = a * s - b * c;
    }
    const int stage_count = 7 - 1;
    for (int stage = 0; stage < stage_count; stage++) {
        int block_count = 1 << stage;
        int block_size_bits = stage_count - stage;
        int block_half_size_bits = block_size_bits - 1;
        int block_size = 1 << block_size_bits;
        int block_half_size = 1 << block_half_size_bits;
        sin_t = T_Sin[block_half_size_bits];
        cos_t = T_Cos[block_half_size_bits];
        for (int block = 0; block < block_count; block++) {
            for (int i = 0; i < block_half_size; i++) {
                int front = (block * block_size + i) * 2;
                int back = front + block_size;
                double a = tmp[front] - tmp[back];
                double b = tmp[front + 1] - tmp[back + 1];
                double s = sin_t[i], c = cos_t[i];
                tmp[front] += tmp[back];
                tmp[front + 1] += tmp[back + 1];
                tmp[back] = a * c + b * s;
                tmp[back + 1] = a * s - b * c;
            }
        }
    }
    for (int i = 0; i < size; i++)
        output[i] = tmp[MDCT_Shuffle128[i]] * g_mdct_scale;
}

/* Mdct.cs:63-92 */
static void run_mdct(mdct_state *m, const double *input, double *output)
{
    const int size = SPSF, half = size / 2;
    double dct_in[SPSF];
    const double *w = T_MdctWindow;
    for (int i = 0; i < half; i++) {
        double a = w[half - i - 1] * -input[half + i];
        double b = w[half + i] * input[half - i - 1];
        double c = w[i] * m->mdct_previous[i];
        double d = w[size - i - 1] * m->mdct_previous[size - i - 1];
        dct_in[i] = a - b;
        dct_in[half + i] = c - d;
    }
    dct4(dct_in, output);
    memcpy(m->mdct_previous, input, sizeof(double) * SPSF);
}

/* Mdct.cs:94-119 */
static void run_imdct(mdct_state *m, const double *input, double *output)
{
    const int size = SPSF, half = size / 2;
    double dct_out[SPSF];
    const double *w = T_MdctWindow;
    dct4(input, dct_out);
    for (int i = 0; i < half; i++) {
        output[i] = w[i] * dct_out[i + half] + m->imdct_previous[i];
        output[i + half] = w[i + half] * -dct_out[size - 1 - i] - m->imdct_previous[i + half];
        m->imdct_previous[i] = w[size - 1 - i] * -dct_out[half - i - 1];
        m->imdct_previous[i + half] = w[half - i - 1] * dct_out[i];
    }
}

/* test hooks: n consecutive 128-sample blocks through one Mdct instance */
void vgo_mdct_run(const double *in, int blocks, double *out, int inverse)
{
    ensure_tables();
    pthread_once(&g_scale_once, init_scale);
    mdct_state m;
    memset(&m, 0, sizeof m);
    for (int b = 0; b < blocks; b++) {
        if (inverse) run_imdct(&m, in + b * SPSF, out + b * SPSF);
        else run_mdct(&m, in + b * SPSF, out + b * SPSF);
    }
}

/* ------------------------------------------------------------------ frame / channel state */
enum { CH_DISCRETE = 0, CH_STEREO_PRIMARY = 1, CH_STEREO_SECONDARY = 2 };

typedef struct {
    int type;
    int coded_sf_count;
    double pcm_float[SUBFRAMES][SPSF];
    double spectra[SUBFRAMES][SPSF];
    double scaled_spectra[SPSF][SUBFRAMES];
    int quantized[SUBFRAMES][SPSF];
    double gain[SPSF];
    int intensity[SUBFRAMES];
    int hfr_scales[8];
    double hfr_group_avg[8];
    mdct_state mdct;
    int scale_factors[SPSF];
    int resolution[SPSF];
    int header_length_bits;
    int sf_delta_bits;
} hca_channel;

typedef struct {
    vgo_hca_info hca;
    int nch;
    hca_channel *ch;
    uint8_t ath_curve[SPSF];
    int acceptable_noise_level;
    int evaluation_boundary;
} hca_frame;

/* CriHcaFrame.cs:33-52 */
static void get_channel_types(const vgo_hca_info *h, int types[8])
{
    static const int tab[9][8] = {{0}};
    (void)tab;
    for (int i = 0; i < 8; i++) types[i] = CH_DISCRETE;
    int cpt = h->channel_count / h->track_count;
    if (h->stereo_band_count == 0 || cpt == 1) return;
    const int P = CH_STEREO_PRIMARY, S = CH_STEREO_SECONDARY, D = CH_DISCRETE;
    switch (cpt) {
    case 2: { int t[] = {P, S}; memcpy(types, t, sizeof t); break; }
    case 3: { int t[] = {P, S, D}; memcpy(types, t, sizeof t); break; }
    case 4:
        if (h->channel_config != 0) { int t[] = {P, S, D, D}; memcpy(types, t, sizeof t); }
        else { int t[] = {P, S, P, S}; memcpy(types, t, sizeof t); }
        break;
    case 5:
        if (h->channel_config > 2) { int t[] = {P, S, D, D, D}; memcpy(types, t, sizeof t); }
        else { int t[] = {P, S, D, P, S}; memcpy(types, t, sizeof t); }
        break;
    case 6: { int t[] = {P, S, D, D, P, S}; memcpy(types, t, sizeof t); break; }
    case 7: { int t[] = {P, S, D, D, P, S, D}; memcpy(types, t, sizeof t); break; }
    case 8: { int t[] = {P, S, D, D, P, S, P, S}; memcpy(types, t, sizeof t); break; }
    default: break;
    }
}

/* CriHcaFrame.cs:60-83 */
static void scale_ath_curve(int frequency, uint8_t ath[SPSF])
{
    int acc = 0, i;
    for (i = 0; i < SPSF; i++) {
        acc += frequency;
        int index = acc >> 13;
        if (index >= 654) break;
        ath[i] = HCA_AthCurve[index];
    }
    for (; i < SPSF; i++) ath[i] = 0xff;
}

/* CriHcaFrame.cs:14-31 */
static hca_frame *frame_new(const vgo_hca_info *h)
{
    hca_frame *f = (hca_frame *)calloc(1, sizeof *f);
    f->hca = *h;
    f->nch = h->channel_count;
    f->ch = (hca_channel *)calloc((size_t)f->nch, sizeof(hca_channel));
    int types[8];
    get_channel_types(h, types);
    for (int i = 0; i < f->nch; i++) {
        f->ch[i].type = types[i];
        f->ch[i].coded_sf_count = types[i] == CH_STEREO_SECONDARY ? h->base_band_count
                                                                  : h->base_band_count + h->stereo_band_count;
    }
    if (h->use_ath_curve) scale_ath_curve(h->sample_rate, f->ath_curve);
    return f;
}
static void frame_free(hca_frame *f) { if (f) { free(f->ch); free(f); } }

/* CriHcaPacking.cs:60-69 */
static int calculate_resolution(int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int curve_position = noise_level - 5 * scale_factor / 2 + 2;
    curve_position = clampi(curve_position, 0, 58);
    return HCA_ScaleToResolutionCurve[curve_position];
}

/* ------------------------------------------------------------------ BitWriter (Utilities/BitWriter.cs) */
typedef struct { uint8_t *buf; int length_bits; int position; } bit_writer;

static void bw_write_fallback(bit_writer *w, int value, int bit_count)
{
    int byte_index = w->position / 8;
    int bit_index = w->position % 8;
    while (bit_count > 0) {
        if (bit_index >= 8) { bit_index = 0; byte_index++; }
        int to_shift = 8 - bit_index - bit_count;
        int shifted = to_shift < 0 ? value >> -to_shift : value << to_shift;
        int bits_to_write = bit_count < 8 - bit_index ? bit_count : 8 - bit_index;
        int mask = ((1 << bits_to_write) - 1) << (8 - bit_index - bits_to_write);
        int out_byte = w->buf[byte_index] & ~mask;
        out_byte |= shifted & mask;
        w->buf[byte_index] = (uint8_t)out_byte;
        bit_index += bits_to_write;
        bit_count -= bits_to_write;
    }
}

/* BitWriter.cs:26-70; returns -1 where the reference throws InvalidOperationException */
static int bw_write(bit_writer *w, int value, int bit_count)
{
    int remaining = w->length_bits - w->position;
    if (bit_count > remaining) return -1;
    int byte_index = w->position / 8;
    int bit_index = w->position % 8;
    if (bit_count <= 9 && remaining >= 16) {
        int out = (int)(((uint32_t)value << (16 - bit_count)) & 0xFFFF) >> bit_index;
        w->buf[byte_index] |= (uint8_t)(out >> 8);
        w->buf[byte_index + 1] = (uint8_t)out;
    } else if (bit_count <= 17 && remaining >= 24) {
        int out = (int)(((uint32_t)value << (24 - bit_count)) & 0xFFFFFF) >> bit_index;
        w->buf[byte_index] |= (uint8_t)(out >> 16);
        w->buf[byte_index + 1] = (uint8_t)(out >> 8);
        w->buf[byte_index + 2] = (uint8_t)out;
    } else if (bit_count <= 25 && remaining >= 32) {
        /* (int)(((value << (32 - bitCount)) & 0xFFFFFFFF) >> bitIndex): the & promotes to long */
        int64_t v = (int64_t)(int32_t)((uint32_t)value << ((32 - bit_count) & 31));
        int out = (int)((v & 0xFFFFFFFFLL) >> bit_index);
        w->buf[byte_index] |= (uint8_t)(out >> 24);
        w->buf[byte_index + 1] = (uint8_t)(out >> 16);
        w->buf[byte_index + 2] = (uint8_t)(out >> 8);
        w->buf[byte_index + 3] = (uint8_t)out;
    } else {
        bw_write_fallback(w, value, bit_count);
    }
    w->position += bit_count;
    return 0;
}

int vgo_bitwriter_write(uint8_t *buf, int buf_len, int position, int value, int bit_count)
{
    bit_writer w = {buf, buf_len * 8, position};
    if (bw_write(&w, value, bit_count)) return -1;
    return w.position;
}

/* ------------------------------------------------------------------ BitReader (Utilities/BitReader.cs) */
typedef struct { const uint8_t *buf; int length_bits; int position; } bit_reader;

static int br_peek_fallback(const bit_reader *r, int bit_count)
{
    int value = 0;
    int byte_index = r->position / 8;
    int bit_index = r->position % 8;
    while (bit_count > 0) {
        if (bit_index >= 8) { bit_index = 0; byte_index++; }
        int bits_to_read = bit_count < 8 - bit_index ? bit_count : 8 - bit_index;
        int mask = 0xFF >> bit_index;
        int current = (mask & r->buf[byte_index]) >> (8 - bit_index - bits_to_read);
        value = (value << bits_to_read) | current;
        bit_index += bits_to_read;
        bit_count -= bits_to_read;
    }
    return value;
}

/* BitReader.cs:51-92 */
static int br_peek(const bit_reader *r, int bit_count)
{
    int remaining = r->length_bits - r->position;
    if (bit_count > remaining) {
        if (r->position >= r->length_bits) return 0;
        int extra = bit_count - remaining;
        return br_peek_fallback(r, remaining) << extra;
    }
    int byte_index = r->position / 8;
    int bit_index = r->position % 8;
    const uint8_t *b = r->buf;
    if (bit_count <= 9 && remaining >= 16) {
        int value = b[byte_index] << 8 | b[byte_index + 1];
        value &= 0xFFFF >> bit_index;
        value >>= 16 - bit_count - bit_index;
        return value;
    }
    if (bit_count <= 17 && remaining >= 24) {
        int value = b[byte_index] << 16 | b[byte_index + 1] << 8 | b[byte_index + 2];
        value &= 0xFFFFFF >> bit_index;
        value >>= 24 - bit_count - bit_index;
        return value;
    }
    if (bit_count <= 25 && remaining >= 32) {
        int value = (int)((uint32_t)b[byte_index] << 24 | b[byte_index + 1] << 16 | b[byte_index + 2] << 8 | b[byte_index + 3]);
        value &= (int)(0xFFFFFFFFu >> bit_index);
        value >>= 32 - bit_count - bit_index;
        return value;
    }
    return br_peek_fallback(r, bit_count);
}
static int br_read(bit_reader *r, int bit_count)
{
    int v = br_peek(r, bit_count);
    r->position += bit_count;
    return v;
}
/* BitReader.cs:36-42, OffsetBias.Positive = 1 */
static int br_read_offset_binary_positive(bit_reader *r, int bit_count)
{
    int offset = (1 << (bit_count - 1)) - 1;
    int value = br_peek(r, bit_count) - offset;
    r->position += bit_count;
    return value;
}

/* ------------------------------------------------------------------ encoder: HcaInfo derivation */
/* CriHcaEncoder.cs:288-324 */
static int calculate_bitrate(const vgo_hca_info *h, int quality, int bitrate, int limit_bitrate)
{
    int pcm_bitrate = h->sample_rate * h->channel_count * 16;
    int max_bitrate = pcm_bitrate / 4;
    int min_bitrate = 0;
    int ratio = 6;
    switch (quality) {
    case 1: ratio = 4; break;                                  /* Highest */
    case 2: ratio = 6; break;                                  /* High */
    case 3: ratio = 8; break;                                  /* Middle */
    case 4: ratio = h->channel_count == 1 ? 10 : 12; break;    /* Low */
    case 5: ratio = h->channel_count == 1 ? 12 : 16; break;    /* Lowest */
    default: break;
    }
    bitrate = bitrate != 0 ? bitrate : pcm_bitrate / ratio;
    if (limit_bitrate) {
        int a = h->channel_count == 1 ? 42666 : 32000 * h->channel_count;
        int b = pcm_bitrate / 6;
        min_bitrate = a < b ? a : b;
    }
    return clampi(bitrate, min_bitrate, max_bitrate);
}

/* CriHcaEncoder.cs:326-368 */
static void calculate_band_counts(vgo_hca_info *h, int bitrate, int cutoff_freq)
{
    h->frame_size = bitrate * 1024 / h->sample_rate / 8;
    int num_groups = 0;
    int pcm_bitrate = h->sample_rate * h->channel_count * 16;
    int hfr_ratio, cutoff_ratio;
    if (h->channel_count <= 1 || pcm_bitrate / bitrate <= 6) { hfr_ratio = 6; cutoff_ratio = 12; }
    else { hfr_ratio = 8; cutoff_ratio = 16; }
    if (bitrate < pcm_bitrate / cutoff_ratio) {
        int alt = cutoff_ratio * bitrate / (32 * h->channel_count);
        cutoff_freq = cutoff_freq < alt ? cutoff_freq : alt;
    }
    int total_band_count = (int)cs_round(cutoff_freq * 256.0 / h->sample_rate);
    double hb = cs_round((hfr_ratio * bitrate * 128.0) / pcm_bitrate);
    int hfr_start_band = (int)((double)total_band_count < hb ? (double)total_band_count : hb);
    int stereo_start_band = hfr_ratio == 6 ? hfr_start_band : (hfr_start_band + 1) / 2;
    int hfr_band_count = total_band_count - hfr_start_band;
    int bands_per_group = divide_by_round_up(hfr_band_count, 8);
    if (bands_per_group > 0) num_groups = divide_by_round_up(hfr_band_count, bands_per_group);
    h->total_band_count = total_band_count;
    h->base_band_count = stereo_start_band;
    h->stereo_band_count = hfr_start_band - stereo_start_band;
    h->hfr_group_count = num_groups;
    h->bands_per_hfr_group = bands_per_group;
}

/* CriHcaEncoder.Initialize :61-114 (+ :370-418).  Returns 0, -2 ArgumentOutOfRange. */
int vgo_hca_encoder_init(const vgo_hca_params *c, vgo_hca_info *h, int *post_samples_out, int *buffer_pre_samples_out)
{
    ensure_tables();
    memset(h, 0, sizeof *h);
    if (c->channel_count > 8 || c->channel_count < 1) return -2;
    int cutoff = c->sample_rate / 2;
    int post_samples = 128;
    h->channel_count = c->channel_count;
    h->track_count = 1;
    h->sample_count = c->sample_count;
    h->sample_rate = c->sample_rate;
    h->min_resolution = 1;
    h->max_resolution = 15;
    h->inserted_samples = SPSF;
    int bitrate = calculate_bitrate(h, c->quality, c->bitrate, c->limit_bitrate);
    if (bitrate <= 0) return -2;
    calculate_band_counts(h, bitrate, cutoff);
    /* HcaInfo.CalculateHfrValues :52-58 */
    if (h->bands_per_hfr_group > 0) {
        h->hfr_band_count = h->total_band_count - h->base_band_count - h->stereo_band_count;
        h->hfr_group_count = divide_by_round_up(h->hfr_band_count, h->bands_per_hfr_group);
    }
    /* SetChannelConfiguration :370-381 */
    {
        int cpt = h->channel_count / h->track_count;
        int cfg = HCA_DefaultChannelMapping[cpt];
        if (HCA_ValidChannelMappings[cpt - 1][cfg] != 1) return -2;
        h->channel_config = cfg;
    }
    int input_sample_count = h->sample_count;
    if (c->looping) {
        h->looping = 1;
        h->sample_count = c->loop_end < c->sample_count ? c->loop_end : c->sample_count;
        h->inserted_samples += get_next_multiple(c->loop_start, SPF) - c->loop_start;
        /* CalculateLoopInfo :383-398 */
        {
            int ls = c->loop_start + h->inserted_samples, le = c->loop_end + h->inserted_samples;
            h->loop_start_frame = ls / SPF;
            h->pre_loop_samples = ls % SPF;
            h->loop_end_frame = le / SPF;
            h->post_loop_samples = SPF - le % SPF;
            if (h->post_loop_samples == SPF) { h->loop_end_frame--; h->post_loop_samples = 0; }
        }
        int a = get_next_multiple(h->sample_count, SPSF);
        input_sample_count = a < c->sample_count ? a : c->sample_count;
        input_sample_count += SPSF * 2;
        post_samples = input_sample_count - h->sample_count;
    }
    /* CalculateHeaderSize :400-418 */
    {
        h->header_size = get_next_multiple(96 + h->comment_length, 32);
        if (h->looping) {
            int off = h->header_size + h->frame_size * h->loop_start_frame;
            int padding_bytes = get_next_multiple(off, 2048) - off;
            int padding_frames = padding_bytes / h->frame_size;
            h->inserted_samples += padding_frames * SPF;
            h->loop_start_frame += padding_frames;
            h->loop_end_frame += padding_frames;
            h->header_size += padding_bytes % h->frame_size;
        }
    }
    int total_samples = input_sample_count + h->inserted_samples;
    h->frame_count = divide_by_round_up(total_samples, SPF);
    h->appended_samples = h->frame_count * SPF - h->inserted_samples - input_sample_count;
    if (post_samples_out) *post_samples_out = post_samples;
    if (buffer_pre_samples_out) *buffer_pre_samples_out = h->inserted_samples - 128;
    return 0;
}

/* ------------------------------------------------------------------ encoder: frame stages */
/* :691-709 */
static int find_scale_factor(double value)
{
    const double *sf = T_DequantizerScaling;
    uint32_t low = 0, high = 63;
    while (low < high) {
        uint32_t mid = (low + high) / 2;
        if (sf[mid] <= value) low = mid + 1;
        else high = mid;
    }
    return (int)low;
}

/* :711-764 */
static void encode_intensity_stereo(hca_frame *f)
{
    if (f->hca.stereo_band_count <= 0) return;
    for (int c = 0; c < f->nch; c++) {
        if (f->ch[c].type != CH_STEREO_PRIMARY) continue;
        for (int sf = 0; sf < SUBFRAMES; sf++) {
            double *l = f->ch[c].spectra[sf];
            double *r = f->ch[c + 1].spectra[sf];
            double energy_l = 0, energy_r = 0, energy_total = 0;
            for (int b = f->hca.base_band_count; b < f->hca.total_band_count; b++) {
                energy_l += fabs(l[b]);
                energy_r += fabs(r[b]);
                energy_total += fabs(l[b] + r[b]);
            }
            energy_total *= 2;
            double energy_lr = energy_r + energy_l;
            double stored_value = 2 * energy_l / energy_lr;
            double energy_ratio = energy_lr / energy_total;
            energy_ratio = clampd(energy_ratio, 0.5, sqrt(2) / 2);
            int quantized = 1;
            if (energy_r > 0 || energy_l > 0) {
                while (quantized < 13 && T_IntensityRatioBounds[quantized] >= stored_value) quantized++;
            } else {
                quantized = 0;
                energy_ratio = 1;
            }
            f->ch[c + 1].intensity[sf] = quantized;
            for (int b = f->hca.base_band_count; b < f->hca.total_band_count; b++) {
                l[b] = (l[b] + r[b]) * energy_ratio;
                r[b] = 0;
            }
        }
    }
}

/* :673-689 */
static void calculate_scale_factors(hca_frame *f)
{
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        for (int b = 0; b < ch->coded_sf_count; b++) {
            double max = 0;
            for (int sf = 0; sf < SUBFRAMES; sf++) {
                double coeff = fabs(ch->spectra[sf][b]);
                max = coeff > max ? coeff : max;      /* Math.Max; spectra are never NaN */
            }
            ch->scale_factors[b] = find_scale_factor(max);
        }
        for (int b = ch->coded_sf_count; b < SPSF; b++) ch->scale_factors[b] = 0;
    }
}

/* :651-671 */
static void scale_spectra(hca_frame *f)
{
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        for (int b = 0; b < ch->coded_sf_count; b++) {
            int scale_factor = ch->scale_factors[b];
            for (int sf = 0; sf < SUBFRAMES; sf++) {
                double coeff = ch->spectra[sf][b];
                ch->scaled_spectra[b][sf] = scale_factor == 0 ? 0
                    : clampd(coeff * T_QuantizerScaling[scale_factor], -0.999999999999, 0.999999999999);
            }
        }
    }
}

/* :766-793 */
static void calculate_hfr_group_averages(hca_frame *f)
{
    const vgo_hca_info *h = &f->hca;
    if (h->hfr_group_count == 0) return;
    int hfr_start_band = h->stereo_band_count + h->base_band_count;
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        if (ch->type == CH_STEREO_SECONDARY) continue;
        for (int group = 0, band = hfr_start_band; group < h->hfr_group_count; group++) {
            double sum = 0.0;
            int count = 0;
            for (int i = 0; i < h->bands_per_hfr_group && band < SPSF; band++, i++) {
                for (int sf = 0; sf < SUBFRAMES; sf++) sum += fabs(ch->spectra[sf][band]);
                count += SUBFRAMES;
            }
            ch->hfr_group_avg[group] = sum / count;
        }
    }
}

/* :795-832 */
static void calculate_hfr_scale(hca_frame *f)
{
    const vgo_hca_info *h = &f->hca;
    if (h->hfr_group_count == 0) return;
    int hfr_start_band = h->stereo_band_count + h->base_band_count;
    int a = h->hfr_band_count, b2 = h->total_band_count - h->hfr_band_count;
    int hfr_band_count = a < b2 ? a : b2;
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        if (ch->type == CH_STEREO_SECONDARY) continue;
        double *group_spectra = ch->hfr_group_avg;
        for (int group = 0, band = 0; group < h->hfr_group_count; group++) {
            double sum = 0.0;
            int count = 0;
            for (int i = 0; i < h->bands_per_hfr_group && band < hfr_band_count; band++, i++) {
                for (int sf = 0; sf < SUBFRAMES; sf++) sum += fabs(ch->scaled_spectra[hfr_start_band - band - 1][sf]);
                count += SUBFRAMES;
            }
            double average = sum / count;
            if (average > 0.0) {
                double inv = 1.0 / average, s2 = sqrt(2);
                group_spectra[group] *= inv < s2 ? inv : s2;
            }
            ch->hfr_scales[group] = find_scale_factor(group_spectra[group]);
        }
    }
}

/* :609-649 */
static void calculate_optimal_delta_length(hca_channel *ch)
{
    int empty = 1;
    for (int i = 0; i < ch->coded_sf_count; i++)
        if (ch->scale_factors[i] != 0) { empty = 0; break; }
    if (empty) { ch->header_length_bits = 3; ch->sf_delta_bits = 0; return; }
    int min_delta_bits = 6;
    int min_length = 3 + 6 * ch->coded_sf_count;
    for (int delta_bits = 1; delta_bits < 6; delta_bits++) {
        int max_delta = (1 << (delta_bits - 1)) - 1;
        int length = 3 + 6;
        for (int band = 1; band < ch->coded_sf_count; band++) {
            int delta = ch->scale_factors[band] - ch->scale_factors[band - 1];
            length += abs(delta) > max_delta ? delta_bits + 6 : delta_bits;
        }
        if (length < min_length) { min_length = length; min_delta_bits = delta_bits; }
    }
    ch->header_length_bits = min_length;
    ch->sf_delta_bits = min_delta_bits;
}

/* :599-607 */
static void calculate_frame_header_length(hca_frame *f)
{
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        calculate_optimal_delta_length(ch);
        if (ch->type == CH_STEREO_SECONDARY) ch->header_length_bits += 32;
        else if (f->hca.hfr_group_count > 0) ch->header_length_bits += 6 * f->hca.hfr_group_count;
    }
}

/* :554-597 */
static int calculate_used_bits(hca_frame *f, int noise_level, int eval_boundary)
{
    int length = 16 + 16 + 16;
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        length += ch->header_length_bits;
        for (int i = 0; i < ch->coded_sf_count; i++) {
            int noise = i < eval_boundary ? noise_level - 1 : noise_level;
            int resolution = calculate_resolution(ch->scale_factors[i], noise);
            if (resolution >= 8) {
                int bits = HCA_QuantizedSpectrumMaxBits[resolution] - 1;
                double dead_zone = T_QuantizerDeadZone[resolution];
                for (int sf = 0; sf < SUBFRAMES; sf++) {
                    length += bits;
                    if (fabs(ch->scaled_spectra[i][sf]) >= dead_zone) length++;
                }
            } else {
                double step_size_inv = T_QuantizerInverseStepSize[resolution];
                double shift_up = step_size_inv + 1;
                int shift_down = (int)(step_size_inv + 0.5 - 8);
                for (int sf = 0; sf < SUBFRAMES; sf++) {
                    int q = (int)(ch->scaled_spectra[i][sf] * step_size_inv + shift_up) - shift_down;
                    length += HCA_QuantizeSpectrumBits[resolution][q];
                }
            }
        }
    }
    return length;
}

/* :502-523 */
static int binary_search_level(hca_frame *f, int available_bits, int low, int high)
{
    int max = high;
    int mid_value = 0;
    while (low != high) {
        int mid = (low + high) / 2;
        mid_value = calculate_used_bits(f, mid, 0);
        if (mid_value > available_bits) low = mid + 1;
        else if (mid_value <= available_bits) high = mid;
    }
    return low == max && mid_value > available_bits ? -1 : low;
}

/* :525-552 */
static int binary_search_boundary(hca_frame *f, int available_bits, int noise_level, int low, int high)
{
    int max = high;
    while (abs(high - low) > 1) {
        int mid = (low + high) / 2;
        int mid_value = calculate_used_bits(f, noise_level, mid);
        if (available_bits < mid_value) high = mid - 1;
        else if (available_bits >= mid_value) low = mid;
    }
    if (low == high) return low < max ? low : -1;
    int hi_value = calculate_used_bits(f, noise_level, high);
    return hi_value > available_bits ? low : high;
}

/* :457-485; returns -3 (InvalidDataException "Bitrate is set too low.") */
static int calculate_noise_level(hca_frame *f)
{
    int highest_band = f->hca.base_band_count + f->hca.stereo_band_count - 1;
    int available_bits = f->hca.frame_size * 8;
    int level = binary_search_level(f, available_bits, 0, 255);
    while (level < 0) {
        highest_band -= 2;
        if (highest_band < 0) return -3;
        for (int c = 0; c < f->nch; c++) {
            f->ch[c].scale_factors[highest_band + 1] = 0;
            f->ch[c].scale_factors[highest_band + 2] = 0;
        }
        calculate_frame_header_length(f);
        level = binary_search_level(f, available_bits, 0, 255);
    }
    f->acceptable_noise_level = level;
    return 0;
}

/* :487-500; -4 where the reference throws NotImplementedException */
static int calculate_evaluation_boundary(hca_frame *f)
{
    if (f->acceptable_noise_level == 0) { f->evaluation_boundary = 0; return 0; }
    int available_bits = f->hca.frame_size * 8;
    int level = binary_search_boundary(f, available_bits, f->acceptable_noise_level, 0, 127);
    if (level < 0) return -4;
    f->evaluation_boundary = level;
    return 0;
}

/* :441-455 */
static void calculate_frame_resolutions(hca_frame *f)
{
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        for (int i = 0; i < f->evaluation_boundary; i++)
            ch->resolution[i] = calculate_resolution(ch->scale_factors[i], f->acceptable_noise_level - 1);
        for (int i = f->evaluation_boundary; i < ch->coded_sf_count; i++)
            ch->resolution[i] = calculate_resolution(ch->scale_factors[i], f->acceptable_noise_level);
        for (int i = ch->coded_sf_count; i < SPSF; i++) ch->resolution[i] = 0;
    }
}

/* :420-439 */
static void quantize_spectra(hca_frame *f)
{
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        for (int i = 0; i < ch->coded_sf_count; i++) {
            int resolution = ch->resolution[i];
            double step_size_inv = T_QuantizerInverseStepSize[resolution];
            double shift_up = step_size_inv + 1;
            int shift_down = (int)(step_size_inv + 0.5);
            for (int sf = 0; sf < SUBFRAMES; sf++)
                ch->quantized[sf][i] = (int)(ch->scaled_spectra[i][sf] * step_size_inv + shift_up) - shift_down;
        }
    }
}

/* CriHcaPacking.cs:262-295 */
static void write_scale_factors(bit_writer *w, const hca_channel *ch)
{
    int delta_bits = ch->sf_delta_bits;
    const int *scales = ch->scale_factors;
    bw_write(w, delta_bits, 3);
    if (delta_bits == 0) return;
    if (delta_bits == 6) {
        for (int i = 0; i < ch->coded_sf_count; i++) bw_write(w, scales[i], 6);
        return;
    }
    bw_write(w, scales[0], 6);
    int max_delta = (1 << (delta_bits - 1)) - 1;
    int escape_value = (1 << delta_bits) - 1;
    for (int i = 1; i < ch->coded_sf_count; i++) {
        int delta = scales[i] - scales[i - 1];
        if (abs(delta) > max_delta) {
            bw_write(w, escape_value, delta_bits);
            bw_write(w, scales[i], 6);
        } else {
            bw_write(w, max_delta + delta, delta_bits);
        }
    }
}

/* CriHcaPacking.cs:238-260 */
static void write_spectra(bit_writer *w, const hca_channel *ch, int sub_frame)
{
    for (int i = 0; i < ch->coded_sf_count; i++) {
        int resolution = ch->resolution[i];
        int q = ch->quantized[sub_frame][i];
        if (resolution == 0) continue;
        if (resolution < 8) {
            int bits = HCA_QuantizeSpectrumBits[resolution][q + 8];
            bw_write(w, HCA_QuantizeSpectrumValue[resolution][q + 8], bits);
        } else if (resolution < 16) {
            int bits = HCA_QuantizedSpectrumMaxBits[resolution] - 1;
            bw_write(w, abs(q), bits);
            if (q != 0) bw_write(w, q > 0 ? 0 : 1, 1);
        }
    }
}

/* CriHcaPacking.cs:17-58, :231-236.  out must be zero-filled (the reference allocates new arrays). */
static void pack_frame(hca_frame *f, uint8_t *out)
{
    bit_writer w = {out, f->hca.frame_size * 8, 0};
    bw_write(&w, 0xffff, 16);
    bw_write(&w, f->acceptable_noise_level, 9);
    bw_write(&w, f->evaluation_boundary, 7);
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        write_scale_factors(&w, ch);
        if (ch->type == CH_STEREO_SECONDARY) {
            for (int i = 0; i < SUBFRAMES; i++) bw_write(&w, ch->intensity[i], 4);
        } else if (f->hca.hfr_group_count > 0) {
            for (int i = 0; i < f->hca.hfr_group_count; i++) bw_write(&w, ch->hfr_scales[i], 6);
        }
    }
    for (int sf = 0; sf < SUBFRAMES; sf++)
        for (int c = 0; c < f->nch; c++) write_spectra(&w, &f->ch[c], sf);
    /* AlignPosition(8) */
    {
        int np = get_next_multiple(w.position, 8);
        bw_write(&w, 0, np - w.position);
    }
    for (int i = w.position / 8; i < f->hca.frame_size - 2; i++) out[i] = 0;
    w.position = w.length_bits - 16;
    uint16_t crc = vgo_crc16(out, f->hca.frame_size - 2);
    bw_write(&w, crc, 16);
}

/* CriHcaEncoder.EncodeFrame :271-286 (+ PcmToFloat :845-858, RunMdct :834-843) */
static int encode_frame(hca_frame *f, int16_t *const *pcm, uint8_t *out)
{
    for (int c = 0; c < f->nch; c++) {
        int idx = 0;
        for (int sf = 0; sf < SUBFRAMES; sf++)
            for (int i = 0; i < SPSF; i++)
                f->ch[c].pcm_float[sf][i] = pcm[c][idx++] * (1.0 / 32768.0);
    }
    for (int c = 0; c < f->nch; c++)
        for (int sf = 0; sf < SUBFRAMES; sf++)
            run_mdct(&f->ch[c].mdct, f->ch[c].pcm_float[sf], f->ch[c].spectra[sf]);
    encode_intensity_stereo(f);
    calculate_scale_factors(f);
    scale_spectra(f);
    calculate_hfr_group_averages(f);
    calculate_hfr_scale(f);
    calculate_frame_header_length(f);
    int rc = calculate_noise_level(f);
    if (rc) return rc;
    rc = calculate_evaluation_boundary(f);
    if (rc) return rc;
    calculate_frame_resolutions(f);
    quantize_spectra(f);
    pack_frame(f, out);
    return 0;
}

/* ------------------------------------------------------------------ encoder: streaming shell (:126-269) */
typedef struct {
    vgo_hca_info hca;
    hca_frame *frame;
    int16_t *pcm_buffer[8];
    int buffer_position;
    int buffer_pre_samples;
    int samples_processed;
    int frames_processed;
    int post_samples;
    int16_t *post_audio[8];
    uint8_t *out;          /* frame_count * frame_size, zero-filled; frames land in order */
    int error;
} hca_encoder;

static int enc_output_frame(hca_encoder *e, int frames_output)
{
    if (SPF - e->buffer_position != 0) return frames_output;
    if (e->frames_processed < e->hca.frame_count) {
        int rc = encode_frame(e->frame, e->pcm_buffer, e->out + (size_t)e->frames_processed * e->hca.frame_size);
        if (rc && !e->error) e->error = rc;
    }
    e->buffer_position = 0;
    e->frames_processed++;
    return frames_output + 1;
}

/* one CriHcaEncoder.Encode(pcm, hcaOut) call; pcm = [channel][1024] */
static int enc_encode(hca_encoder *e, int16_t *const *pcm)
{
    const int nch = e->hca.channel_count;
    if (e->frames_processed >= e->hca.frame_count) return -4;   /* InvalidOperationException */
    int frames_output = 0;
    int pcm_position = 0;

    if (e->buffer_pre_samples > 0) {                            /* EncodePreAudio :170-190 */
        while (e->buffer_pre_samples > SPF) {
            e->buffer_position = SPF;
            frames_output = enc_output_frame(e, frames_output);
            e->buffer_pre_samples -= SPF;
        }
        for (int j = 0; j < e->buffer_pre_samples; j++)
            for (int i = 0; i < nch; i++) e->pcm_buffer[i][j] = pcm[i][0];
        e->buffer_position = e->buffer_pre_samples;
        e->buffer_pre_samples = 0;
    }

    int loop_start_sample = e->hca.loop_start_frame * 1024 + e->hca.pre_loop_samples - e->hca.inserted_samples;
    if (e->hca.looping && loop_start_sample + e->post_samples >= e->samples_processed &&
        loop_start_sample < e->samples_processed + SPF) {       /* SaveLoopAudio :244-254 */
        int a = loop_start_sample - e->samples_processed;
        int start_pos = a > 0 ? a : 0;
        int b = e->samples_processed - loop_start_sample;
        int loop_pos = b > 0 ? b : 0;
        int c = loop_start_sample - e->samples_processed + e->post_samples;
        int end_pos = c < SPF ? c : SPF;
        int length = end_pos - start_pos;
        for (int i = 0; i < nch; i++)
            memcpy(e->post_audio[i] + loop_pos, pcm[i] + start_pos, (size_t)length * sizeof(int16_t));
    }

    while (SPF - pcm_position > 0 && e->hca.sample_count > e->samples_processed) {   /* EncodeMainAudio :192-207 */
        int to_copy = SPF - e->buffer_position;
        if (SPF - pcm_position < to_copy) to_copy = SPF - pcm_position;
        if (e->hca.sample_count - e->samples_processed < to_copy) to_copy = e->hca.sample_count - e->samples_processed;
        for (int i = 0; i < nch; i++)
            memcpy(e->pcm_buffer[i] + e->buffer_position, pcm[i] + pcm_position, (size_t)to_copy * sizeof(int16_t));
        e->buffer_position += to_copy;
        e->samples_processed += to_copy;
        pcm_position += to_copy;
        frames_output = enc_output_frame(e, frames_output);
    }

    if (e->hca.sample_count == e->samples_processed) {         /* EncodePostAudio :209-242 */
        int post_pos = 0;
        int remaining = e->post_samples;
        while (post_pos < remaining) {
            int to_copy = SPF - e->buffer_position;
            if (remaining - post_pos < to_copy) to_copy = remaining - post_pos;
            for (int i = 0; i < nch; i++)
                memcpy(e->pcm_buffer[i] + e->buffer_position, e->post_audio[i] + post_pos, (size_t)to_copy * sizeof(int16_t));
            e->buffer_position += to_copy;
            post_pos += to_copy;
            frames_output = enc_output_frame(e, frames_output);
        }
        while (e->frames_processed < e->hca.frame_count) {
            for (int i = 0; i < nch; i++)
                memset(e->pcm_buffer[i] + e->buffer_position, 0, (size_t)(SPF - e->buffer_position) * sizeof(int16_t));
            e->buffer_position = SPF;
            frames_output = enc_output_frame(e, frames_output);
        }
    }
    return frames_output;
}

/* CriHcaFormat.EncodeFromPcm16 :34-84.  pcm planar (channel c at pcm + c*pitch), frames_out holds
 * frame_count*frame_size bytes (sizes from vgo_hca_encoder_init).  Returns 0 or the error code. */
int vgo_hca_encode(const int16_t *pcm, long pitch, const vgo_hca_params *c, vgo_hca_info *info_out, uint8_t *frames_out)
{
    ensure_tables();
    pthread_once(&g_scale_once, init_scale);
    hca_encoder e;
    memset(&e, 0, sizeof e);
    int rc = vgo_hca_encoder_init(c, &e.hca, &e.post_samples, &e.buffer_pre_samples);
    if (rc) return rc;
    const int nch = e.hca.channel_count;
    e.frame = frame_new(&e.hca);
    for (int i = 0; i < nch; i++) {
        e.pcm_buffer[i] = (int16_t *)calloc(SPF, sizeof(int16_t));
        e.post_audio[i] = (int16_t *)calloc((size_t)(e.post_samples > 0 ? e.post_samples : 1), sizeof(int16_t));
    }
    e.out = frames_out;
    memset(frames_out, 0, (size_t)e.hca.frame_count * e.hca.frame_size);

    int16_t *chunk[8];
    for (int i = 0; i < nch; i++) chunk[i] = (int16_t *)calloc(SPF, sizeof(int16_t));
    int frame_num = 0;
    for (int i = 0; frame_num < e.hca.frame_count; i++) {
        int samples_to_copy = c->sample_count - i * SPF;
        if (samples_to_copy > SPF) samples_to_copy = SPF;
        if (samples_to_copy < 0) { rc = -2; break; }           /* Array.Copy would throw */
        for (int ch = 0; ch < nch; ch++)
            memcpy(chunk[ch], pcm + (size_t)ch * pitch + (size_t)SPF * i, (size_t)samples_to_copy * sizeof(int16_t));
        int written = enc_encode(&e, chunk);
        if (written <= 0) { rc = written == 0 ? -5 : written; break; }   /* "Encoder returned no audio" */
        frame_num += written;
    }
    if (!rc) rc = e.error;
    if (info_out) *info_out = e.hca;
    for (int i = 0; i < nch; i++) { free(e.pcm_buffer[i]); free(e.post_audio[i]); free(chunk[i]); }
    frame_free(e.frame);
    return rc;
}

/* ------------------------------------------------------------------ decoder */
/* CriHcaPacking.cs:185-211 */
static int delta_decode(bit_reader *r, int delta_bits, int data_bits, int count, int *output)
{
    output[0] = br_read(r, data_bits);
    int max_delta = 1 << (delta_bits - 1);
    int max_value = (1 << data_bits) - 1;
    for (int i = 1; i < count; i++) {
        int delta = br_read_offset_binary_positive(r, delta_bits);
        if (delta < max_delta) {
            int value = output[i - 1] + delta;
            if (value < 0 || value > max_value) return 0;
            output[i] = value;
        } else {
            output[i] = br_read(r, data_bits);
        }
    }
    return 1;
}

/* CriHcaPacking.cs:111-130 */
static int read_scale_factors(hca_channel *ch, bit_reader *r)
{
    ch->sf_delta_bits = br_read(r, 3);
    if (ch->sf_delta_bits == 0) {
        memset(ch->scale_factors, 0, sizeof ch->scale_factors);
        return 1;
    }
    if (ch->sf_delta_bits >= 6) {
        for (int i = 0; i < ch->coded_sf_count; i++) ch->scale_factors[i] = br_read(r, 6);
        return 1;
    }
    return delta_decode(r, ch->sf_delta_bits, 6, ch->coded_sf_count, ch->scale_factors);
}

/* CriHcaPacking.cs:10-15, :71-183, :213-229.  -3: InvalidDataException("Invalid frame header") */
static int unpack_frame(hca_frame *f, bit_reader *r)
{
    int sync = br_read(r, 16);
    if (sync != 0xffff) return -3;
    f->acceptable_noise_level = br_read(r, 9);
    f->evaluation_boundary = br_read(r, 7);
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        if (!read_scale_factors(ch, r)) return 1;                     /* UnpackFrameHeader returns false */
        for (int i = 0; i < f->evaluation_boundary; i++)
            ch->resolution[i] = calculate_resolution(ch->scale_factors[i], f->ath_curve[i] + f->acceptable_noise_level - 1);
        for (int i = f->evaluation_boundary; i < ch->coded_sf_count; i++)
            ch->resolution[i] = calculate_resolution(ch->scale_factors[i], f->ath_curve[i] + f->acceptable_noise_level);
        if (ch->type == CH_STEREO_SECONDARY) {
            for (int i = 0; i < SUBFRAMES; i++) ch->intensity[i] = br_read(r, 4);
        } else if (f->hca.hfr_group_count > 0) {
            for (int i = 0; i < f->hca.hfr_group_count; i++) ch->hfr_scales[i] = br_read(r, 6);
        }
    }
    /* ReadSpectralCoefficients :148-183 */
    for (int sf = 0; sf < SUBFRAMES; sf++) {
        for (int c = 0; c < f->nch; c++) {
            hca_channel *ch = &f->ch[c];
            for (int s = 0; s < ch->coded_sf_count; s++) {
                int resolution = ch->resolution[s];
                int bits = HCA_QuantizedSpectrumMaxBits[resolution];
                int code = br_peek(r, bits);
                if (resolution < 8) {
                    bits = HCA_QuantizedSpectrumBits[resolution][code];
                    ch->quantized[sf][s] = HCA_QuantizedSpectrumValue[resolution][code];
                } else {
                    int q = code / 2 * (1 - (code % 2 * 2));
                    if (q == 0) bits--;
                    ch->quantized[sf][s] = q;
                }
                r->position += bits;
            }
            for (int s = ch->coded_sf_count; s < 0x80; s++) ch->spectra[sf][s] = 0;
        }
    }
    return 0;
}

/* CriHcaPacking.UnpackingWasSuccessful / FrameEmpty (CriHcaPacking.cs:213-237) */
static int unpacking_was_successful(const hca_frame *f, const bit_reader *r)
{
    int remaining = r->length_bits - r->position;
    int empty = f->acceptable_noise_level <= 0;
    for (int c = 0; c < f->nch && empty; c++)
        if (f->ch[c].sf_delta_bits > 0) empty = 0;
    return (remaining >= 16 && remaining <= 128) || empty || (f->acceptable_noise_level == 0 && remaining >= 16);
}

/* CriHcaEncryption.FindKey / TestKey / FindFirstNonEmptyFrame / FrameEmpty (CriHcaEncryption.cs:34-88) over a
 * caller-supplied list of 256-byte DECRYPTION tables.  Returns the index of the first key under which the first ten
 * non-empty frames unpack, -1 when none does, -3 when a frame's sync word is wrong (InvalidDataException). */
int vgo_hca_find_key(const vgo_hca_info *h, const uint8_t *frames, int frame_count, const uint8_t *tables, int nkeys)
{
    ensure_tables();
    const int fs = h->frame_size;
    int start = 0;
    for (int i = 0; i < frame_count; i++) {
        int empty = 1;
        for (int b = 2; b < fs - 2; b++)
            if (frames[(size_t)i * fs + b]) { empty = 0; break; }
        if (!empty) { start = i; break; }
    }
    int end = frame_count < start + 10 ? frame_count : start + 10;
    hca_frame *f = frame_new(h);
    uint8_t *buffer = (uint8_t *)malloc((size_t)fs);
    int found = -1;
    for (int k = 0; k < nkeys && found == -1; k++) {
        const uint8_t *table = tables + (size_t)k * 256;
        int ok = 1;
        for (int i = start; i < end && ok; i++) {
            memcpy(buffer, frames + (size_t)i * fs, (size_t)fs);
            vgo_hca_crypt(buffer, 1, fs, table);                          /* CryptFrame(.., doDecrypt: true) */
            bit_reader r = {buffer, fs * 8, 0};
            int rc = unpack_frame(f, &r);
            if (rc == -3) { found = -3; ok = 0; break; }
            if (rc == 1 || !unpacking_was_successful(f, &r)) ok = 0;
        }
        if (ok && found == -1) found = k;
    }
    free(buffer);
    frame_free(f);
    return found;
}

/* CriHcaDecoder.DecodeFrame :72-192 */
static int decode_frame(hca_frame *f, const uint8_t *audio, int16_t *const *pcm_out)
{
    bit_reader r = {audio, f->hca.frame_size * 8, 0};
    int rc = unpack_frame(f, &r);
    if (rc < 0) return rc;
    /* DequantizeFrame :83-114 */
    for (int c = 0; c < f->nch; c++) {
        hca_channel *ch = &f->ch[c];
        for (int i = 0; i < ch->coded_sf_count; i++)
            ch->gain[i] = T_DequantizerScaling[ch->scale_factors[i]] * T_QuantizerStepSize[ch->resolution[i]];
    }
    for (int sf = 0; sf < SUBFRAMES; sf++)
        for (int c = 0; c < f->nch; c++) {
            hca_channel *ch = &f->ch[c];
            for (int s = 0; s < ch->coded_sf_count; s++) ch->spectra[sf][s] = ch->quantized[sf][s] * ch->gain[s];
        }
    /* ReconstructHighFrequency :116-145 */
    const vgo_hca_info *h = &f->hca;
    if (h->hfr_group_count != 0) {
        int total_band_count = h->total_band_count < 127 ? h->total_band_count : 127;
        int hfr_start_band = h->base_band_count + h->stereo_band_count;
        int a = h->hfr_band_count, b = total_band_count - h->hfr_band_count;
        int hfr_band_count = a < b ? a : b;
        for (int c = 0; c < f->nch; c++) {
            hca_channel *ch = &f->ch[c];
            if (ch->type == CH_STEREO_SECONDARY) continue;
            for (int group = 0, band = 0; group < h->hfr_group_count; group++) {
                for (int i = 0; i < h->bands_per_hfr_group && band < hfr_band_count; band++, i++) {
                    int high_band = hfr_start_band + band;
                    int low_band = hfr_start_band - band - 1;
                    int index = ch->hfr_scales[group] - ch->scale_factors[low_band] + 64;
                    for (int sf = 0; sf < SUBFRAMES; sf++)
                        ch->spectra[sf][high_band] = T_ScaleConversion[index & 127] * ch->spectra[sf][low_band];
                }
            }
        }
    }
    /* ApplyIntensityStereo :147-166 */
    if (h->stereo_band_count > 0) {
        for (int c = 0; c < f->nch; c++) {
            if (f->ch[c].type != CH_STEREO_PRIMARY) continue;
            for (int sf = 0; sf < SUBFRAMES; sf++) {
                double *l = f->ch[c].spectra[sf];
                double *r2 = f->ch[c + 1].spectra[sf];
                int iq = f->ch[c + 1].intensity[sf];
                double ratio_l = T_IntensityRatio[iq < 15 ? iq : 14];
                double ratio_r = ratio_l - 2.0;
                for (int b = h->base_band_count; b < h->total_band_count; b++) {
                    r2[b] = l[b] * ratio_r;
                    l[b] *= ratio_l;
                }
            }
        }
    }
    /* RunImdct :168-177 */
    for (int sf = 0; sf < SUBFRAMES; sf++)
        for (int c = 0; c < f->nch; c++)
            run_imdct(&f->ch[c].mdct, f->ch[c].spectra[sf], f->ch[c].pcm_float[sf]);
    /* PcmFloatToShort :179-192 */
    for (int c = 0; c < f->nch; c++)
        for (int sf = 0; sf < SUBFRAMES; sf++)
            for (int s = 0; s < SPSF; s++) {
                int sample = double_to_int(f->ch[c].pcm_float[sf][s] * (32767 + 1));
                pcm_out[c][sf * SPSF + s] = clamp16(sample);
            }
    return 0;
}

/* CriHcaDecoder.Decode :11-45.  frames: frame_count*frame_size bytes; pcm_out planar with pitch. */
int vgo_hca_decode(const vgo_hca_info *h, const uint8_t *frames, int16_t *pcm_out, long pitch)
{
    ensure_tables();
    pthread_once(&g_scale_once, init_scale);
    const int nch = h->channel_count;
    if (nch < 1 || nch > 8) return -2;
    hca_frame *f = frame_new(h);
    int16_t *buf[8];
    for (int c = 0; c < nch; c++) {
        buf[c] = (int16_t *)calloc(SPF, sizeof(int16_t));
        memset(pcm_out + (size_t)c * pitch, 0, (size_t)h->sample_count * sizeof(int16_t));
    }
    int rc = 0;
    for (int i = 0; i < h->frame_count; i++) {
        rc = decode_frame(f, frames + (size_t)i * h->frame_size, buf);
        if (rc) break;
        /* CopyPcmToOutput :31-45 */
        int current = i * SPF - h->inserted_samples;
        int remaining = h->sample_count - current;
        if (h->sample_count < remaining) remaining = h->sample_count;
        int src_start = clampi(0 - current, 0, SPF);
        int dest_start = current > 0 ? current : 0;
        int length = SPF - src_start;
        if (remaining < length) length = remaining;
        if (length <= 0) continue;
        for (int c = 0; c < nch; c++)
            memcpy(pcm_out + (size_t)c * pitch + dest_start, buf[c] + src_start, (size_t)length * sizeof(int16_t));
    }
    for (int c = 0; c < nch; c++) free(buf[c]);
    frame_free(f);
    return rc;
}

/* ------------------------------------------------------------------ batch drivers
 * HCA has no intra-file parallelism in the reference (CriHcaFormat.cs:53-81); batches of files run
 * under Parallel.ForEach with ProcessorCount-1 workers (Cli/Batch.cs:24-25): one task per stream. */
typedef struct {
    const int16_t *pcm; long stream_pitch; long ch_pitch; const vgo_hca_params *p; int nstreams;
    uint8_t *frames; long frames_pitch; const uint8_t *dec_frames; const vgo_hca_info *info; int16_t *dec_out;
    int next; pthread_mutex_t mu; int decode; int error;
} hca_job;

static void *hca_worker(void *arg)
{
    hca_job *j = (hca_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int s = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (s >= j->nstreams) break;
        int rc;
        if (!j->decode) {
            vgo_hca_info info;
            rc = vgo_hca_encode(j->pcm + (size_t)s * j->stream_pitch, j->ch_pitch, j->p, &info,
                                j->frames + (size_t)s * j->frames_pitch);
        } else {
            rc = vgo_hca_decode(j->info, j->dec_frames + (size_t)s * j->frames_pitch,
                                j->dec_out + (size_t)s * j->stream_pitch, j->ch_pitch);
        }
        if (rc) j->error = rc;
    }
    return NULL;
}

static int hca_run(hca_job *j, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 1024) threads = 1024;
    pthread_mutex_init(&j->mu, NULL);
    j->next = 0;
    j->error = 0;
    if (threads == 1) hca_worker(j);
    else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, hca_worker, j);
        for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
        free(t);
    }
    pthread_mutex_destroy(&j->mu);
    return j->error;
}

/* pcm: stream s, channel c at pcm + s*stream_pitch + c*ch_pitch; frames: stream s at frames + s*frames_pitch */
int vgo_hca_encode_batch(const int16_t *pcm, long stream_pitch, long ch_pitch, int nstreams, const vgo_hca_params *p,
                         uint8_t *frames, long frames_pitch, int threads)
{
    hca_job j;
    memset(&j, 0, sizeof j);
    j.pcm = pcm; j.stream_pitch = stream_pitch; j.ch_pitch = ch_pitch; j.p = p; j.nstreams = nstreams;
    j.frames = frames; j.frames_pitch = frames_pitch; j.decode = 0;
    return hca_run(&j, threads);
}

int vgo_hca_decode_batch(const vgo_hca_info *info, const uint8_t *frames, long frames_pitch, int nstreams,
                         int16_t *pcm_out, long stream_pitch, long ch_pitch, int threads)
{
    hca_job j;
    memset(&j, 0, sizeof j);
    j.info = info; j.dec_frames = frames; j.frames_pitch = frames_pitch; j.nstreams = nstreams;
    j.dec_out = pcm_out; j.stream_pitch = stream_pitch; j.ch_pitch = ch_pitch; j.decode = 1;
    return hca_run(&j, threads);
}

/* introspection for the tests: stage outputs of one frame encode of a fresh encoder fed `frames`
 * consecutive 1024-sample blocks (mono/stereo...), returning the LAST frame's per-channel values */
int vgo_hca_debug_last_frame(const int16_t *pcm, long pitch, const vgo_hca_params *c, int frames,
                             int *noise_level, int *eval_boundary, int *scale_factors /*nch*128*/,
                             int *resolution /*nch*128*/, int *quantized /*nch*8*128*/, double *spectra /*nch*8*128*/)
{
    ensure_tables();
    pthread_once(&g_scale_once, init_scale);
    vgo_hca_info h;
    int rc = vgo_hca_encoder_init(c, &h, NULL, NULL);
    if (rc) return rc;
    hca_frame *f = frame_new(&h);
    uint8_t *out = (uint8_t *)calloc((size_t)h.frame_size, 1);
    int16_t *chp[8];
    for (int k = 0; k < frames; k++) {
        for (int ch = 0; ch < h.channel_count; ch++) chp[ch] = (int16_t *)(pcm + (size_t)ch * pitch + (size_t)k * SPF);
        memset(out, 0, (size_t)h.frame_size);
        rc = encode_frame(f, chp, out);
        if (rc) break;
    }
    if (!rc) {
        *noise_level = f->acceptable_noise_level;
        *eval_boundary = f->evaluation_boundary;
        for (int ch = 0; ch < h.channel_count; ch++) {
            memcpy(scale_factors + ch * 128, f->ch[ch].scale_factors, sizeof(int) * 128);
            memcpy(resolution + ch * 128, f->ch[ch].resolution, sizeof(int) * 128);
            memcpy(quantized + ch * 1024, f->ch[ch].quantized, sizeof(int) * 1024);
            memcpy(spectra + ch * 1024, f->ch[ch].spectra, sizeof(double) * 1024);
        }
    }
    free(out);
    frame_free(f);
    return rc;
}
