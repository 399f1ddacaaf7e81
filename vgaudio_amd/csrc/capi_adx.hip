// capi_adx.hip -- C-ABI entry points for CRI ADX (see include/vgaudio_hip.h).
#include "common.hpp"
#include "host_batch.hpp"
#include "adx_kernels.hpp"

#include <cmath>
#include <cstring>
#include <vector>

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

// channels per chunk of the host pipeline (host_pipeline.hpp): the piece-wise kernels fill the chip from 1024 channels on
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

// ---------------------------------------------------------------- ragged batches (VGAudio.Cli/Batch.cs:24-25: a worker per FILE)
// Every channel with its own length and its own CriAdxParameters (a file's sample rate sets the high-pass coefficients,
// CriAdxCodec.cs:64): the channels are sorted into buckets of one parameter set and similar length (host_batch.hpp,
// plan_buckets), a bucket's rows are zero-padded on the device to its longest channel and run through the equal-length
// kernels; every channel receives the prefix that is its own encoding / decoding.
namespace {

// parameter groups: channels whose device parameters are the same bytes
int group_of(std::vector<adx::AdxDeviceParams> &seen, const adx::AdxDeviceParams &d)
{
    for (size_t i = 0; i < seen.size(); i++)
        if (memcmp(&seen[i], &d, sizeof d) == 0) return (int)i;
    seen.push_back(d);
    return (int)seen.size() - 1;
}

constexpr int64_t ADX_BUCKET_VOLUME = (int64_t)1024 * 2880000;   // padded samples per chunk: what ADX_CHUNK_CHANNELS x 60 s hold

int adx_encode_batch_v_one(const int16_t *const *pcm, const int *lengths, int nch, const vga_adx_params *params, uint8_t *const *out,
                           int16_t *history_out)
{
    if (nch < 0) { set_error("negative channel count"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (!pcm || !lengths || !params || !out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    std::vector<adx::AdxDeviceParams> dps;
    std::vector<int> group(nch), length(nch);
    for (int c = 0; c < nch; c++) {
        if (int rc = validate(&params[c])) return rc;
        if (lengths[c] < 0) { set_error("channel %d: negative length", c); return VGA_ERR_ARGUMENT; }
        if (lengths[c] == 0 && params[c].version == 4 && params[c].padding == 0) {
            set_error("channel %d: empty PCM: the reference reads pcm[0] (CriAdxCodec.cs:71)", c);
            return VGA_ERR_ARGUMENT;
        }
        if ((!pcm[c] && lengths[c] > 0) || !out[c]) { set_error("channel %d is null", c); return VGA_ERR_ARGUMENT; }
        adx::AdxDeviceParams d;
        memset(&d, 0, sizeof d);
        d = device_params(&params[c], true);
        // An empty channel keeps buckets of its own: with padding % samplesPerFrame != 0 the reference SKIPS the frame in
        // which the padding ends (`if (samplesToCopy == 0) continue`, CriAdxCodec.cs:84) and leaves zero bytes there, while
        // the zero-padded run of a longer bucket would encode silence into it (a non-zero header for the Exponential and
        // Fixed types) -- the one case in which a channel's output is not a prefix of the padded channel's.
        group[c] = 2 * group_of(dps, d) + (lengths[c] == 0 ? 1 : 0);
        length[c] = lengths[c];
    }
    if (int rc = require_device()) return rc;
    // chunks of at most 256 channels: what is left after the upload (the last, largest chunk's kernels and its download) is
    // shorter, and the equal-length kernels still fill their launch (10 008 files: 578 ms against 591-601 with 1024, 647-651
    // with 128 -- 79 launches of ~10 ms are more than the upload hides; profiles/r05_q_ragged_host_orders.log)
    const BucketPlan plan = plan_buckets(group, length, 256, ADX_BUCKET_VOLUME, false);
    const int chunks = (int)plan.chunk_begin.size() - 1;
    // device layout: chunk k's rows pitch_k apart behind the chunks before it
    std::vector<int64_t> pcm_base(chunks + 1, 0), out_base(chunks + 1, 0), pcm_pitch(chunks), out_pitch(chunks);
    std::vector<const vga_adx_params *> chunk_params(chunks);
    for (int k = 0; k < chunks; k++) {
        const int count = plan.chunk_begin[k + 1] - plan.chunk_begin[k];
        chunk_params[k] = &params[plan.order[plan.chunk_begin[k]]];
        pcm_pitch[k] = round_up(std::max(plan.chunk_length[k], 1), 8);
        out_pitch[k] = round_up(std::max(vga_adx_encoded_byte_count(plan.chunk_length[k], chunk_params[k]), 2), 16);
        pcm_base[k + 1] = pcm_base[k] + pcm_pitch[k] * count;
        out_base[k + 1] = out_base[k] + out_pitch[k] * count;
    }
    std::vector<const void *> in_rows(nch);
    std::vector<void *> out_rows(nch);
    std::vector<size_t> in_size(nch), in_off(nch), out_size(nch), out_off(nch);
    size_t max_in = 16, max_out = 16;
    for (int k = 0; k < chunks; k++)
        for (int i = plan.chunk_begin[k]; i < plan.chunk_begin[k + 1]; i++) {
            const int c = plan.order[i], j = i - plan.chunk_begin[k];
            in_rows[i] = pcm[c];
            out_rows[i] = out[c];
            in_size[i] = (size_t)lengths[c] * 2;
            in_off[i] = (size_t)(pcm_base[k] + j * pcm_pitch[k]) * 2;
            out_size[i] = (size_t)vga_adx_encoded_byte_count(lengths[c], &params[c]);
            out_off[i] = (size_t)(out_base[k] + j * out_pitch[k]);
            max_in = std::max(max_in, (size_t)pcm_pitch[k] * 2);
            max_out = std::max(max_out, (size_t)out_pitch[k]);
        }
    DevBuf d_pcm, d_out, d_hist, d_own;
    // every channel's own frame count, in the plan's order: the seams in a channel's padding are left alone (adx_kernels.hpp)
    std::vector<int> own(nch);
    for (int i = 0; i < nch; i++) own[i] = divide_by_round_up(lengths[plan.order[i]] + params[plan.order[i]].padding, 32);   // (frames of the padded stream)
    VGA_HIP_TRY(d_own.alloc((size_t)nch * sizeof(int)));
    VGA_HIP_TRY(hipMemcpy(d_own.p, own.data(), (size_t)nch * sizeof(int), hipMemcpyHostToDevice));
    VGA_HIP_TRY(d_pcm.alloc((size_t)pcm_base[chunks] * 2 + 64));
    VGA_HIP_TRY(hipMemset(d_pcm.p, 0, (size_t)pcm_base[chunks] * 2 + 64));           // the padding behind every row is silence
    VGA_HIP_TRY(d_out.alloc((size_t)out_base[chunks] + 64));
    VGA_HIP_TRY(d_hist.alloc((size_t)nch * 2));
    pipe::Job job;
    job.units = nch;
    job.chunk_begin = plan.chunk_begin;
    job.in_rows = in_rows.data();
    job.in_row_sizes = in_size.data();
    job.d_in_offsets = in_off.data();
    job.in_row_bytes = max_in;
    job.d_in_pitch = max_in;
    job.d_in = d_pcm.as<char>();
    job.out_rows = out_rows.data();
    job.out_row_sizes = out_size.data();
    job.d_out_offsets = out_off.data();
    job.out_row_bytes = max_out;
    job.d_out_pitch = max_out;
    job.d_out = d_out.as<char>();
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int k = plan.chunk_of(first);
        const int rc = adx::launch_encode(d_pcm.as<int16_t>() + pcm_base[k], pcm_pitch[k], count, plan.chunk_length[k], dps[plan.chunk_group[k] / 2],
                                          d_out.as<uint8_t>() + out_base[k], out_pitch[k], d_hist.as<int16_t>() + first, s,
                                          d_own.as<int>() + first);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, ADX_CHUNK_CHANNELS)) return rc;
    if (history_out) {
        std::vector<int16_t> h(nch);
        VGA_HIP_TRY(hipMemcpy(h.data(), d_hist.p, (size_t)nch * 2, hipMemcpyDeviceToHost));
        for (int i = 0; i < nch; i++) history_out[plan.order[i]] = h[i];
    }
    return VGA_OK;
}

int adx_decode_batch_v_one(const uint8_t *const *adpcm, const int *adpcm_lengths, int nch, const int *sample_counts,
                           const vga_adx_params *params, int16_t *const *pcm_out)
{
    if (nch < 0) { set_error("negative channel count"); return VGA_ERR_ARGUMENT; }
    if (nch == 0) return VGA_OK;
    if (!adpcm || !adpcm_lengths || !sample_counts || !params || !pcm_out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    std::vector<adx::AdxDeviceParams> dps;
    std::vector<int> group(nch), length(nch);
    std::vector<size_t> need(nch);
    for (int c = 0; c < nch; c++) {
        if (int rc = validate(&params[c])) return rc;
        if (sample_counts[c] < 0 || adpcm_lengths[c] < 0) { set_error("channel %d: negative size", c); return VGA_ERR_ARGUMENT; }
        const int spf = (params[c].frame_size - 2) * 2;
        need[c] = sample_counts[c] == 0 ? 0 : (size_t)(params[c].padding / spf) * params[c].frame_size +
                                                  (size_t)divide_by_round_up(sample_counts[c], spf) * params[c].frame_size;
        if ((size_t)adpcm_lengths[c] < need[c]) {
            set_error("channel %d: ADX stream too short: %d bytes, decoder reads %zu", c, adpcm_lengths[c], need[c]);
            return VGA_ERR_ARGUMENT;
        }
        if (sample_counts[c] > 0 && (!adpcm[c] || !pcm_out[c])) { set_error("channel %d is null", c); return VGA_ERR_ARGUMENT; }
        adx::AdxDeviceParams d;
        memset(&d, 0, sizeof d);
        d = device_params(&params[c], false);
        group[c] = group_of(dps, d);
        length[c] = sample_counts[c];
    }
    if (int rc = require_device()) return rc;
    const BucketPlan plan = plan_buckets(group, length, ADX_CHUNK_CHANNELS, ADX_BUCKET_VOLUME, false);
    const int chunks = (int)plan.chunk_begin.size() - 1;
    std::vector<int64_t> in_base(chunks + 1, 0), pcm_base(chunks + 1, 0), in_pitch(chunks), pcm_pitch(chunks);
    for (int k = 0; k < chunks; k++) {
        const int count = plan.chunk_begin[k + 1] - plan.chunk_begin[k];
        const vga_adx_params &p = params[plan.order[plan.chunk_begin[k]]];
        const int spf = (p.frame_size - 2) * 2;
        const int64_t bytes = (int64_t)(p.padding / spf) * p.frame_size + (int64_t)divide_by_round_up(plan.chunk_length[k], spf) * p.frame_size;
        in_pitch[k] = round_up(std::max<int64_t>(bytes, 2), 16);
        pcm_pitch[k] = round_up(std::max(plan.chunk_length[k], 1), 8);
        in_base[k + 1] = in_base[k] + in_pitch[k] * count;
        pcm_base[k + 1] = pcm_base[k] + pcm_pitch[k] * count;
    }
    std::vector<const void *> in_rows(nch);
    std::vector<void *> out_rows(nch);
    std::vector<size_t> in_size(nch), in_off(nch), out_size(nch), out_off(nch);
    size_t max_in = 16, max_out = 16;
    for (int k = 0; k < chunks; k++)
        for (int i = plan.chunk_begin[k]; i < plan.chunk_begin[k + 1]; i++) {
            const int c = plan.order[i], j = i - plan.chunk_begin[k];
            in_rows[i] = adpcm[c];
            out_rows[i] = pcm_out[c];
            in_size[i] = need[c];
            in_off[i] = (size_t)(in_base[k] + j * in_pitch[k]);
            out_size[i] = (size_t)sample_counts[c] * 2;
            out_off[i] = (size_t)(pcm_base[k] + j * pcm_pitch[k]) * 2;
            max_in = std::max(max_in, (size_t)in_pitch[k]);
            max_out = std::max(max_out, (size_t)pcm_pitch[k] * 2);
        }
    DevBuf d_in, d_pcm, d_status, d_own;
    std::vector<int> own(nch);                                                      // (as the encoder's: the plan's order)
    for (int i = 0; i < nch; i++) own[i] = sample_counts[plan.order[i]];
    VGA_HIP_TRY(d_own.alloc((size_t)nch * sizeof(int)));
    VGA_HIP_TRY(hipMemcpy(d_own.p, own.data(), (size_t)nch * sizeof(int), hipMemcpyHostToDevice));
    VGA_HIP_TRY(d_in.alloc((size_t)in_base[chunks] + 64));
    VGA_HIP_TRY(hipMemset(d_in.p, 0, (size_t)in_base[chunks] + 64));                 // frames behind a row's end: scale 0, filter 0
    VGA_HIP_TRY(d_pcm.alloc((size_t)pcm_base[chunks] * 2 + 64));
    VGA_HIP_TRY(d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
    pipe::Job job;
    job.units = nch;
    job.chunk_begin = plan.chunk_begin;
    job.in_rows = in_rows.data();
    job.in_row_sizes = in_size.data();
    job.d_in_offsets = in_off.data();
    job.in_row_bytes = max_in;
    job.d_in_pitch = max_in;
    job.d_in = d_in.as<char>();
    job.out_rows = out_rows.data();
    job.out_row_sizes = out_size.data();
    job.d_out_offsets = out_off.data();
    job.out_row_bytes = max_out;
    job.d_out_pitch = max_out;
    job.d_out = d_pcm.as<char>();
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int k = plan.chunk_of(first);
        int rc = VGA_OK;
        if (plan.chunk_length[k] > 0)
            rc = adx::launch_decode(d_in.as<uint8_t>() + in_base[k], in_pitch[k], count, plan.chunk_length[k], dps[plan.chunk_group[k]],
                                    d_pcm.as<int16_t>() + pcm_base[k], pcm_pitch[k], d_status.as<int>(), s, d_own.as<int>() + first);
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

}  // namespace

extern "C" {

int vga_adx_encode_batch_v(const int16_t *const *pcm, const int *pcm_lengths, int nch, const vga_adx_params *params,
                           uint8_t *const *out, int16_t *history_out)
{
    if (nch <= 0 || !pcm || !pcm_lengths || !params || !out) return adx_encode_batch_v_one(pcm, pcm_lengths, nch, params, out, history_out);
    return for_each_device_share(nch, ADX_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return adx_encode_batch_v_one(pcm + first, pcm_lengths + first, count, params + first, out + first,
                                      history_out ? history_out + first : nullptr);
    });
}

int vga_adx_decode_batch_v(const uint8_t *const *adpcm, const int *adpcm_lengths, int nch, const int *sample_counts,
                           const vga_adx_params *params, int16_t *const *pcm_out)
{
    if (nch <= 0 || !adpcm || !adpcm_lengths || !sample_counts || !params || !pcm_out)
        return adx_decode_batch_v_one(adpcm, adpcm_lengths, nch, sample_counts, params, pcm_out);
    return for_each_device_share(nch, ADX_MIN_SHARE_CHANNELS, [&](int first, int count) {
        return adx_decode_batch_v_one(adpcm + first, adpcm_lengths + first, count, sample_counts + first, params + first, pcm_out + first);
    });
}

}  // extern "C"
