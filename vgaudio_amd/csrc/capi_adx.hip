// capi_adx.hip -- C-ABI entry points for CRI ADX (see include/vgaudio_hip.h).
#include "common.hpp"
#include "host_batch.hpp"
#include "adx_kernels.hpp"

#include <cmath>

using namespace vga;

namespace {

// CriAdxCodec.cs:186-191
const int16_t kFixedCoefs[4][2] = {{0, 0}, {0x0F00, 0}, {0x1CC0, (int16_t)0xF300}, {0x1880, (int16_t)0xF240}};

int divide_by_round_up(int v, int d) { return (int)std::ceil((double)v / d); }   // Extensions.cs:145

// CriAdxCodec.cs:173-184 (host side: libm cos/sqrt, as the reference uses Math.Cos/Math.Sqrt)
void calculate_coefficients(int highpass_freq, int sample_rate, int16_t coefs[2])
{
    const double sqrt2 = std::sqrt(2.0);
    const double a = sqrt2 - std::cos(2.0 * M_PI * highpass_freq / sample_rate);
    const double b = sqrt2 - 1;
    const double c = (a - std::sqrt((a + b) * (a - b))) / b;
    coefs[0] = (int16_t)(int)(c * 8192);
    coefs[1] = (int16_t)(int)(c * c * -4096);
}

int validate(const vga_adx_params *p)
{
    if (!p) { set_error("null ADX parameters"); return VGA_ERR_ARGUMENT; }
    if (p->frame_size < 4 || (p->frame_size & 1) || p->frame_size > 255) {
        set_error("ADX frame size %d unsupported (even, 4..254)", p->frame_size);
        return VGA_ERR_ARGUMENT;
    }
    if (p->type != 2 && p->type != 3 && p->type != 4) { set_error("ADX type %d unknown", p->type); return VGA_ERR_ARGUMENT; }
    if (p->type == 2 && (p->filter < 0 || p->filter > 3)) {
        set_error("ADX fixed filter %d out of range", p->filter);       // Coefs[c.Filter] throws
        return VGA_ERR_ARGUMENT;
    }
    if (p->padding < 0) { set_error("negative padding"); return VGA_ERR_ARGUMENT; }
    if (p->type != 2 && p->sample_rate <= 0) { set_error("sample rate must be positive"); return VGA_ERR_ARGUMENT; }
    return VGA_OK;
}

adx::AdxDeviceParams device_params(const vga_adx_params *p, bool encode)
{
    adx::AdxDeviceParams d;
    d.frame_size = p->frame_size;
    d.version = p->version;
    d.type = p->type;
    d.filter = p->filter;
    d.padding = p->padding;
    d.history = p->history;
    int16_t c[2];
    if (p->type == 2) { c[0] = kFixedCoefs[p->filter & 3][0]; c[1] = kFixedCoefs[p->filter & 3][1]; }
    else calculate_coefficients(encode ? 500 : p->highpass_frequency, p->sample_rate, c);   // :64 vs :13
    d.coef0 = c[0];
    d.coef1 = c[1];
    return d;
}

}  // namespace

// channels per chunk of the host pipeline (host_pipeline.hpp): the tiled kernels fill the chip from 1024 channels on
static constexpr int ADX_CHUNK_CHANNELS = 1024;

extern "C" {

void vga_adx_default_params(vga_adx_params *p)
{
    if (!p) return;
    p->sample_rate = 48000; p->highpass_frequency = 500; p->frame_size = 18; p->version = 4;
    p->history = 0; p->padding = 0; p->type = 3; p->filter = 0;
}

int vga_adx_calculate_coefficients(int highpass_freq, int sample_rate, int16_t *coefs_out)
{
    if (!coefs_out || sample_rate <= 0) { set_error("bad arguments"); return VGA_ERR_ARGUMENT; }
    calculate_coefficients(highpass_freq, sample_rate, coefs_out);
    return VGA_OK;
}

// Formats/CriAdx/CriAdxHelpers.cs:7-31
int vga_adx_nibble_count_to_sample_count(int nibble_count, int frame_size)
{
    const int npf = frame_size * 2, spf = npf - 4;
    const int frames = nibble_count / npf, extra = nibble_count % npf;
    return spf * frames + (extra < 4 ? 0 : extra - 4);
}
int vga_adx_sample_count_to_nibble_count(int sample_count, int frame_size)
{
    const int npf = frame_size * 2, spf = npf - 4;
    const int frames = sample_count / spf, extra = sample_count % spf;
    return npf * frames + (extra == 0 ? 0 : extra + 4);
}
int vga_adx_sample_count_to_byte_count(int sample_count, int frame_size)
{
    const int n = vga_adx_sample_count_to_nibble_count(sample_count, frame_size);
    return (n / 2) + (n & 1);
}

int vga_adx_encoded_byte_count(int pcm_length, const vga_adx_params *p)
{
    if (validate(p) != VGA_OK || pcm_length < 0) return VGA_ERR_ARGUMENT;
    const int spf = (p->frame_size - 2) * 2;
    return divide_by_round_up(pcm_length + p->padding, spf) * p->frame_size;
}

int vga_adx_encode_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length, const vga_adx_params *p,
                          uint8_t *d_out, int64_t out_pitch, int16_t *d_history_out, void *stream)
{
    if (int rc = validate(p)) return rc;
    if (nch < 0 || pcm_length < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (pcm_length == 0 && p->version == 4 && p->padding == 0) {
        set_error("empty PCM: the reference reads pcm[0] (CriAdxCodec.cs:71)");
        return VGA_ERR_ARGUMENT;
    }
    const int nbytes = vga_adx_encoded_byte_count(pcm_length, p);
    if (pcm_pitch < pcm_length || out_pitch < nbytes || (out_pitch & 1) || ((uintptr_t)d_out & 1)) {
        set_error("bad pitch/alignment (pcm_pitch=%lld, out_pitch=%lld, need >= %d and even)", (long long)pcm_pitch,
                  (long long)out_pitch, nbytes);
        return VGA_ERR_ARGUMENT;
    }
    return adx::launch_encode(d_pcm, pcm_pitch, nch, pcm_length, device_params(p, true), d_out, out_pitch,
                              d_history_out, (hipStream_t)stream);
}

int vga_adx_decode_device(const uint8_t *d_adpcm, int64_t in_pitch, int adpcm_length, int nch, int sample_count,
                          const vga_adx_params *p, int16_t *d_pcm, int64_t pcm_pitch, int *d_status, void *stream)
{
    if (int rc = validate(p)) return rc;
    if (nch < 0 || sample_count < 0 || adpcm_length < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0 || sample_count == 0) return VGA_OK;
    const int spf = (p->frame_size - 2) * 2;
    const long long need = (long long)(p->padding / spf) * p->frame_size +
                           (long long)divide_by_round_up(sample_count, spf) * p->frame_size;
    if (adpcm_length < need || in_pitch < adpcm_length || pcm_pitch < sample_count) {
        set_error("ADX stream too short: %d bytes, decoder reads %lld", adpcm_length, need);   // IndexOutOfRange in C#
        return VGA_ERR_ARGUMENT;
    }
    return adx::launch_decode(d_adpcm, in_pitch, nch, sample_count, device_params(p, false), d_pcm, pcm_pitch, d_status,
                              (hipStream_t)stream);
}

static constexpr int ADX_MIN_SHARE_CHANNELS = 128;     // channels per share of a call spread over several GPUs (vga_set_devices)

static int adx_encode_batch_one(const int16_t *const *pcm, int nch, int pcm_length, const vga_adx_params *p, uint8_t *const *out,
                                int16_t *history_out);
int vga_adx_encode_batch(const int16_t *const *pcm, int nch, int pcm_length, const vga_adx_params *p,
                         uint8_t *const *out, int16_t *history_out)
{
    if (nch <= 0 || !pcm || !out) return adx_encode_batch_one(pcm, nch, pcm_length, p, out, history_out);
    return for_each_device_share(nch, ADX_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return adx_encode_batch_one(pcm + first, count, pcm_length, p, out + first, history_out ? history_out + first : nullptr);
    });
}
static int adx_encode_batch_one(const int16_t *const *pcm, int nch, int pcm_length, const vga_adx_params *p, uint8_t *const *out,
                                int16_t *history_out)
{
    if (int rc = validate(p)) return rc;
    if (nch < 0 || pcm_length < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (!pcm || !out) { set_error("null channel array"); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < nch; c++)
        if ((!pcm[c] && pcm_length > 0) || !out[c]) { set_error("channel %d is null", c); return VGA_ERR_ARGUMENT; }
    if (pcm_length == 0 && p->version == 4 && p->padding == 0) {
        set_error("empty PCM: the reference reads pcm[0] (CriAdxCodec.cs:71)");
        return VGA_ERR_ARGUMENT;
    }
    if (int rc = require_device()) return rc;
    DevBuf d_pcm, d_out, d_hist;
    const int64_t pcm_pitch = round_up(pcm_length > 0 ? pcm_length : 1, 8);
    const int nbytes = vga_adx_encoded_byte_count(pcm_length, p);
    const int64_t out_pitch = round_up(nbytes > 0 ? nbytes : 2, 16);
    VGA_HIP_TRY(d_pcm.alloc((size_t)nch * pcm_pitch * 2));
    VGA_HIP_TRY(d_out.alloc((size_t)nch * out_pitch));
    VGA_HIP_TRY(d_hist.alloc((size_t)nch * 2));
    const adx::AdxDeviceParams dp = device_params(p, true);
    pipe::Job job;
    job.units = nch;
    if (pcm_length > 0) {
        job.in_rows = (const void *const *)pcm;
        job.in_row_bytes = (size_t)pcm_length * 2;
        job.d_in = d_pcm.as<char>();
        job.d_in_pitch = (size_t)pcm_pitch * 2;
    }
    if (nbytes > 0) {
        job.out_rows = (void *const *)out;
        job.out_row_bytes = (size_t)nbytes;
        job.d_out = d_out.as<char>();
        job.d_out_pitch = (size_t)out_pitch;
    }
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = adx::launch_encode(d_pcm.as<int16_t>() + (int64_t)first * pcm_pitch, pcm_pitch, count, pcm_length, dp,
                                          d_out.as<uint8_t>() + (int64_t)first * out_pitch, out_pitch, d_hist.as<int16_t>() + first, s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, ADX_CHUNK_CHANNELS)) return rc;
    if (history_out) VGA_HIP_TRY(hipMemcpy(history_out, d_hist.p, (size_t)nch * 2, hipMemcpyDeviceToHost));
    return VGA_OK;
}

static int adx_decode_batch_one(const uint8_t *const *adpcm, int adpcm_length, int nch, int sample_count, const vga_adx_params *p,
                                int16_t *const *pcm_out);
int vga_adx_decode_batch(const uint8_t *const *adpcm, int adpcm_length, int nch, int sample_count,
                         const vga_adx_params *p, int16_t *const *pcm_out)
{
    if (nch <= 0 || !adpcm || !pcm_out) return adx_decode_batch_one(adpcm, adpcm_length, nch, sample_count, p, pcm_out);
    return for_each_device_share(nch, ADX_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return adx_decode_batch_one(adpcm + first, adpcm_length, count, sample_count, p, pcm_out + first);
    });
}
static int adx_decode_batch_one(const uint8_t *const *adpcm, int adpcm_length, int nch, int sample_count, const vga_adx_params *p,
                                int16_t *const *pcm_out)
{
    if (int rc = validate(p)) return rc;
    if (nch < 0 || sample_count < 0 || adpcm_length < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nch == 0 || sample_count == 0) return VGA_OK;
    if (!adpcm || !pcm_out) { set_error("null channel array"); return VGA_ERR_ARGUMENT; }
    for (int c = 0; c < nch; c++)
        if (!adpcm[c] || !pcm_out[c]) { set_error("channel %d is null", c); return VGA_ERR_ARGUMENT; }
    const int spf = (p->frame_size - 2) * 2;
    const long long need = (long long)(p->padding / spf) * p->frame_size +
                           (long long)divide_by_round_up(sample_count, spf) * p->frame_size;
    if (adpcm_length < need) {
        set_error("ADX stream too short: %d bytes, decoder reads %lld", adpcm_length, need);
        return VGA_ERR_ARGUMENT;
    }
    if (int rc = require_device()) return rc;
    DevBuf d_in, d_pcm, d_status;
    const int64_t in_pitch = round_up(adpcm_length, 16);
    const int64_t pcm_pitch = round_up(sample_count, 8);
    VGA_HIP_TRY(d_in.alloc((size_t)nch * in_pitch));
    VGA_HIP_TRY(d_pcm.alloc((size_t)nch * pcm_pitch * 2));
    VGA_HIP_TRY(d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
    const adx::AdxDeviceParams dp = device_params(p, false);
    pipe::Job job;
    job.units = nch;
    job.in_rows = (const void *const *)adpcm;
    job.in_row_bytes = (size_t)adpcm_length;
    job.d_in = d_in.as<char>();
    job.d_in_pitch = (size_t)in_pitch;
    job.out_rows = (void *const *)pcm_out;
    job.out_row_bytes = (size_t)sample_count * 2;
    job.d_out = d_pcm.as<char>();
    job.d_out_pitch = (size_t)pcm_pitch * 2;
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = adx::launch_decode(d_in.as<uint8_t>() + (int64_t)first * in_pitch, in_pitch, count, sample_count, dp,
                                          d_pcm.as<int16_t>() + (int64_t)first * pcm_pitch, pcm_pitch, d_status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, ADX_CHUNK_CHANNELS)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
    if (status != 0) {
        set_error("a frame names a filter the coefficient table lacks (IndexOutOfRangeException in the reference)");
        return VGA_ERR_ARGUMENT;
    }
    return VGA_OK;
}

}  // extern "C"
