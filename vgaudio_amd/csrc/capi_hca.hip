// capi_hca.hip -- C-ABI entry points for CRI HCA (see include/vgaudio_hip.h).
// Host side: CriHcaEncoder.Initialize (stream parameters), channel typing, ATH curve; the per-frame
// work is entirely in hca_encode_kernel.hip / hca_decode_kernels.hip.
#include "common.hpp"
#include <deque>
#include <memory>
#include "host_batch.hpp"
#include "hca_kernels.hpp"

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace hosttab {
#include "hca_tables_host.inc"
}

using namespace vga;

namespace {

int divide_by_round_up(int v, int d) { return (int)std::ceil((double)v / d); }        // Extensions.cs:145
int get_next_multiple(int value, int multiple)                                        // Helpers.cs:71-80
{
    if (multiple <= 0) return value;
    if (value % multiple == 0) return value;
    return value + multiple - value % multiple;
}
int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// CriHcaEncoder.cs:288-324
int calculate_bitrate(const vga_hca_info &h, int quality, int bitrate, int limit_bitrate)
{
    const int pcm_bitrate = h.sample_rate * h.channel_count * 16;
    const int max_bitrate = pcm_bitrate / 4;
    int min_bitrate = 0;
    int ratio = 6;
    switch (quality) {
    case 1: ratio = 4; break;
    case 2: ratio = 6; break;
    case 3: ratio = 8; break;
    case 4: ratio = h.channel_count == 1 ? 10 : 12; break;
    case 5: ratio = h.channel_count == 1 ? 12 : 16; break;
    default: break;
    }
    bitrate = bitrate != 0 ? bitrate : pcm_bitrate / ratio;
    if (limit_bitrate) min_bitrate = std::min(h.channel_count == 1 ? 42666 : 32000 * h.channel_count, pcm_bitrate / 6);
    return clampi(bitrate, min_bitrate, max_bitrate);
}

// CriHcaEncoder.cs:326-368
void calculate_band_counts(vga_hca_info &h, int bitrate, int cutoff_freq)
{
    h.frame_size = bitrate * 1024 / h.sample_rate / 8;
    int num_groups = 0;
    const int pcm_bitrate = h.sample_rate * h.channel_count * 16;
    int hfr_ratio, cutoff_ratio;
    if (h.channel_count <= 1 || pcm_bitrate / bitrate <= 6) { hfr_ratio = 6; cutoff_ratio = 12; }
    else { hfr_ratio = 8; cutoff_ratio = 16; }
    if (bitrate < pcm_bitrate / cutoff_ratio)
        cutoff_freq = std::min(cutoff_freq, cutoff_ratio * bitrate / (32 * h.channel_count));
    const int total_band_count = (int)std::nearbyint(cutoff_freq * 256.0 / h.sample_rate);        // Math.Round
    const int hfr_start_band = (int)std::min((double)total_band_count,
                                             std::nearbyint((hfr_ratio * bitrate * 128.0) / pcm_bitrate));
    const int stereo_start_band = hfr_ratio == 6 ? hfr_start_band : (hfr_start_band + 1) / 2;
    const int hfr_band_count = total_band_count - hfr_start_band;
    const int bands_per_group = divide_by_round_up(hfr_band_count, 8);
    if (bands_per_group > 0) num_groups = divide_by_round_up(hfr_band_count, bands_per_group);
    h.total_band_count = total_band_count;
    h.base_band_count = stereo_start_band;
    h.stereo_band_count = hfr_start_band - stereo_start_band;
    h.hfr_group_count = num_groups;
    h.bands_per_hfr_group = bands_per_group;
}

// CriHcaFrame.cs:33-52
void channel_types(const vga_hca_info &h, int types[8])
{
    for (int i = 0; i < 8; i++) types[i] = hca::CH_DISCRETE;
    const int cpt = h.channel_count / (h.track_count > 0 ? h.track_count : 1);
    if (h.stereo_band_count == 0 || cpt == 1) return;
    const int P = hca::CH_STEREO_PRIMARY, S = hca::CH_STEREO_SECONDARY, D = hca::CH_DISCRETE;
    const int t2[] = {P, S}, t3[] = {P, S, D}, t4a[] = {P, S, D, D}, t4b[] = {P, S, P, S}, t5a[] = {P, S, D, D, D},
              t5b[] = {P, S, D, P, S}, t6[] = {P, S, D, D, P, S}, t7[] = {P, S, D, D, P, S, D},
              t8[] = {P, S, D, D, P, S, P, S};
    const int *src = nullptr;
    switch (cpt) {
    case 2: src = t2; break;
    case 3: src = t3; break;
    case 4: src = h.channel_config != 0 ? t4a : t4b; break;
    case 5: src = h.channel_config > 2 ? t5a : t5b; break;
    case 6: src = t6; break;
    case 7: src = t7; break;
    case 8: src = t8; break;
    default: break;
    }
    if (src) for (int i = 0; i < cpt; i++) types[i] = src[i];
}

int make_device_info(const vga_hca_info &h, hca::DeviceInfo &d)
{
    if (h.channel_count < 1 || h.channel_count > 8 || h.frame_size < 8 || h.frame_size > 0xFFFF || h.frame_count < 0 ||
        h.total_band_count < 0 || h.total_band_count > 128 || h.base_band_count < 0 || h.stereo_band_count < 0 ||
        h.base_band_count + h.stereo_band_count > 128 || h.hfr_group_count < 0 || h.hfr_group_count > 8 ||
        (h.hfr_group_count > 0 && h.bands_per_hfr_group <= 0)) {
        set_error("HcaInfo is inconsistent (channels %d, frame size %d, bands %d/%d/%d, hfr groups %d)", h.channel_count,
                  h.frame_size, h.total_band_count, h.base_band_count, h.stereo_band_count, h.hfr_group_count);
        return VGA_ERR_ARGUMENT;
    }
    memset(&d, 0, sizeof d);
    d.nch = h.channel_count;
    d.frame_size = h.frame_size;
    d.frame_count = h.frame_count;
    d.sample_count = h.sample_count;
    d.inserted_samples = h.inserted_samples;
    d.total_band_count = h.total_band_count;
    d.base_band_count = h.base_band_count;
    d.stereo_band_count = h.stereo_band_count;
    d.hfr_band_count = h.hfr_band_count;
    d.bands_per_hfr_group = h.bands_per_hfr_group;
    d.hfr_group_count = h.hfr_group_count;
    int types[8];
    channel_types(h, types);
    for (int i = 0; i < 8; i++) {
        d.channel_type[i] = types[i];
        d.coded_count[i] = types[i] == hca::CH_STEREO_SECONDARY ? h.base_band_count
                                                                 : h.base_band_count + h.stereo_band_count;
    }
    if (h.use_ath_curve) {                                     // CriHcaFrame.ScaleAthCurve :60-83
        int acc = 0, i;
        for (i = 0; i < 128; i++) {
            acc += h.sample_rate;
            const int index = acc >> 13;
            if (index >= 654) break;
            d.ath_curve[i] = hosttab::HCA_AthCurve[index];
        }
        for (; i < 128; i++) d.ath_curve[i] = 0xff;
    }
    return VGA_OK;
}

// x^(8k) mod (x^16 + x^15 + x^2 + 1), k = 0..4095, uploaded once per device
struct CrcPow {
    std::mutex mu;
    uint16_t *dev[64] = {};
} g_crc_pow;

}  // namespace

// shared with the encryption pass (capi_crypt.hip)
namespace vga { namespace hca { int crc_pow_table(const uint16_t **out); } }
// used by capi_crypt.hip (vga_hca_find_key)
namespace vga { namespace hca { int device_info_from(const vga_hca_info &h, DeviceInfo &d) { return make_device_info(h, d); } } }

extern "C" int vga_testing_hca_device_info(const void *hca_info, void *out, int out_bytes)
{
    if (!hca_info || !out || out_bytes < (int)sizeof(hca::DeviceInfo)) { set_error("bad arguments"); return VGA_ERR_ARGUMENT; }
    hca::DeviceInfo d;
    if (int rc = make_device_info(*static_cast<const vga_hca_info *>(hca_info), d)) return rc;
    memcpy(out, &d, sizeof d);
    return VGA_OK;
}

int vga::hca::crc_pow_table(const uint16_t **out)
{
    int device = 0;
    VGA_HIP_TRY(hipGetDevice(&device));
    if (device < 0 || device >= 64) { set_error("device index out of range"); return VGA_ERR_DEVICE; }
    std::lock_guard<std::mutex> lock(g_crc_pow.mu);
    if (!g_crc_pow.dev[device]) {
        static uint16_t host[4096];
        unsigned v = 1;                      // x^0
        for (int k = 0; k < 4096; k++) {
            host[k] = (uint16_t)v;
            for (int j = 0; j < 8; j++) v = ((v << 1) ^ ((v & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        }
        uint16_t *d = nullptr;
        VGA_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof host));
        VGA_HIP_TRY(hipMemcpy(d, host, sizeof host, hipMemcpyHostToDevice));
        g_crc_pow.dev[device] = d;
    }
    *out = g_crc_pow.dev[device];
    return VGA_OK;
}

namespace {

using vga::hca::crc_pow_table;

int status_to_error(int status)
{
    if (status & 16) { set_error("internal: the encoder's bit-cost table could not be built"); return VGA_ERR_DEVICE; }   // (hca_encode_kernel.hip: cost_lut_build)
    if (status & 4) { set_error("Bitrate is set too low."); return VGA_ERR_INVALID_DATA; }     // CriHcaEncoder.cs:471
    if (status & 8) { set_error("evaluation boundary search failed (NotImplementedException in the reference)"); return VGA_ERR_INVALID_OP; }
    if (status & 1) { set_error("Invalid frame header"); return VGA_ERR_INVALID_DATA; }        // CriHcaPacking.cs:76
    if (status & 2) { set_error("scale-factor delta out of range (frame state would be stale in the reference)"); return VGA_ERR_INVALID_DATA; }
    return VGA_OK;
}

// the encoder's input stream (hca_device.hpp PcmMap), from the fields CriHcaEncoder.Initialize derived
int make_pcm_map(const vga_hca_info &h, int pcm_length, hca::PcmMap &m)
{
    const int input_samples = h.frame_count * hca::SPF - h.inserted_samples - h.appended_samples;
    const int pre = h.inserted_samples - hca::SPSF;
    if (pre < 0 || h.sample_count < 0 || h.sample_count > pcm_length || input_samples < h.sample_count) {
        set_error("HcaInfo does not describe this PCM (sample count %d of %d, inserted %d, appended %d)", h.sample_count,
                  pcm_length, h.inserted_samples, h.appended_samples);
        return VGA_ERR_ARGUMENT;
    }
    m.zero_pre = pre > hca::SPF ? (divide_by_round_up(pre, hca::SPF) - 1) * hca::SPF : 0;
    m.pre_end = pre;
    m.main_end = pre + h.sample_count;
    m.post_end = m.main_end + (h.looping ? input_samples - h.sample_count : 0);   // not looping: _postAudio is all zero
    m.loop_start = h.loop_start_frame * hca::SPF + h.pre_loop_samples - h.inserted_samples;
    m.last_chunk = h.sample_count > 0 ? (h.sample_count - 1) / hca::SPF : 0;
    m.raw_len = pcm_length;
    return VGA_OK;
}

}  // namespace

// ---------------------------------------------------------------- CriHcaEncoder's streaming shell (CriHcaEncoder.cs:126-269)
// The reference's encoder is a stateful object fed [channels][1024] blocks; Encode() returns how many frames the block
// completed -- none while the 1024-sample buffer fills, several when the pre-audio of a looping stream or the post-audio at
// the end flush whole frames -- the first into the caller's buffer, the rest into a queue (GetPendingFrame).  Here a frame is
// a function of the stream's PCM alone (hca_device.hpp PcmMap; every frame independent), so the shell keeps what the caller
// has fed in HBM (whole blocks: SaveLoopAudio :244-254 reads a block beyond the stream's last sample), walks the reference's
// counters to know which frames this call completes, and runs hca_encode_kernel on exactly those.
struct vga_hca_stream {
    vga_hca_info info;
    hca::DeviceInfo dev;
    hca::PcmMap map;
    int device = 0;
    int nch = 0, chunks = 0, chunks_fed = 0;
    // the reference's counters (Initialize :61-114): BufferPreSamples, BufferPosition, SamplesProcessed, FramesProcessed, PostSamples
    int buffer_pre = 0, buffer_pos = 0, samples_processed = 0, frames_processed = 0, post_samples = 0;
    std::deque<std::vector<uint8_t>> pending;
    DevBuf d_pcm, d_frames, d_status;
    std::vector<uint8_t> host_frames;
    hipStream_t s = nullptr;
};

extern "C" {

int vga_hca_stream_create(const vga_hca_params *c, vga_hca_info *info_out, vga_hca_stream **out)
{
    if (!out) { set_error("null output"); return VGA_ERR_ARGUMENT; }
    *out = nullptr;
    vga_hca_info h;
    if (int rc = vga_hca_encoder_initialize(c, &h)) return rc;
    if (info_out) *info_out = h;
    if (h.channel_count >= 1 && h.frame_size * 8 < 48 + 3 * h.channel_count + 16) {       // (vga_hca_encode_device)
        set_error("Bitrate is set too low.");
        return VGA_ERR_INVALID_DATA;
    }
    if (int rc = require_device()) return rc;
    std::unique_ptr<vga_hca_stream> st(new vga_hca_stream);
    st->info = h;
    if (int rc = make_device_info(h, st->dev)) return rc;
    st->nch = h.channel_count;
    // the blocks the reference consumes: one per started 1024 samples of the (loop-trimmed) stream, at least one
    st->chunks = std::max(1, divide_by_round_up(h.sample_count, hca::SPF));
    if (int rc = make_pcm_map(h, st->chunks * hca::SPF, st->map)) return rc;
    st->map.last_chunk = st->chunks - 1;
    const int input_samples = h.frame_count * hca::SPF - h.inserted_samples - h.appended_samples;
    st->post_samples = h.looping ? input_samples - h.sample_count : hca::SPSF;           // :70, :99
    st->buffer_pre = h.inserted_samples - hca::SPSF;                                        // :113
    (void)hipGetDevice(&st->device);
    const size_t pcm_bytes = (size_t)st->nch * st->chunks * hca::SPF * 2;
    VGA_HIP_TRY(st->d_pcm.alloc(pcm_bytes));
    VGA_HIP_TRY(hipMemset(st->d_pcm.p, 0, pcm_bytes));
    VGA_HIP_TRY(st->d_frames.alloc((size_t)round_up((int64_t)h.frame_count * h.frame_size + 8, 16)));
    VGA_HIP_TRY(st->d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(st->d_status.p, 0, sizeof(int)));
    VGA_HIP_TRY(hipStreamCreateWithFlags(&st->s, hipStreamNonBlocking));
    *out = st.release();
    return VGA_OK;
}

void vga_hca_stream_destroy(vga_hca_stream *st)
{
    if (!st) return;
    if (st->s) (void)hipStreamDestroy(st->s);
    delete st;
}

int vga_hca_stream_frame_size(const vga_hca_stream *st) { return st ? st->info.frame_size : 0; }
int vga_hca_stream_frames_processed(const vga_hca_stream *st) { return st ? st->frames_processed : 0; }
int vga_hca_stream_pending_frame_count(const vga_hca_stream *st) { return st ? (int)st->pending.size() : 0; }

int vga_hca_stream_get_pending_frame(vga_hca_stream *st, uint8_t *frame_out)
{
    if (!st || !frame_out) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (st->pending.empty()) { set_error("There are no pending frames"); return VGA_ERR_INVALID_OP; }    // :158
    std::memcpy(frame_out, st->pending.front().data(), st->pending.front().size());
    st->pending.pop_front();
    return VGA_OK;
}

int vga_hca_stream_encode(vga_hca_stream *st, const int16_t *const *pcm, uint8_t *hca_out, int *frames_output)
{
    if (!st || !pcm || !hca_out || !frames_output) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    *frames_output = 0;
    const vga_hca_info &h = st->info;
    if (st->frames_processed >= h.frame_count) {                                            // :128-131
        set_error("All audio frames have already been output by the encoder");
        return VGA_ERR_INVALID_OP;
    }
    for (int c = 0; c < st->nch; c++)
        if (!pcm[c]) { set_error("pcm[%d] is null", c); return VGA_ERR_ARGUMENT; }
    int device = -1;
    (void)hipGetDevice(&device);
    if (device != st->device) { set_error("the stream was created on device %d, the current one is %d", st->device, device); return VGA_ERR_ARGUMENT; }
    // the block joins the stream's PCM in HBM (blocks past the stream's end carry nothing the reference reads)
    if (st->chunks_fed < st->chunks) {
        const int64_t ch_pitch = (int64_t)st->chunks * hca::SPF;
        for (int c = 0; c < st->nch; c++)
            VGA_HIP_TRY(hipMemcpyAsync(st->d_pcm.as<int16_t>() + c * ch_pitch + (int64_t)st->chunks_fed * hca::SPF, pcm[c],
                                       hca::SPF * sizeof(int16_t), hipMemcpyHostToDevice, st->s));
        st->chunks_fed++;
        // the caller reuses its block buffer for the next call (CriHcaFormat.cs:50-56 does): the block is on the device before
        // this call returns, whether or not it completes a frame
        VGA_HIP_TRY(hipStreamSynchronize(st->s));
    }
    // ---- the reference's counters through this call (Encode :126-156 and what it calls): how many frames does it complete?
    // They are walked on the object and put back if the frames they promise cannot be delivered (launch, copy or status
    // failure): a caller that retries then gets the same frames instead of skipping them.
    const int saved_pre = st->buffer_pre, saved_pos = st->buffer_pos, saved_samples = st->samples_processed,
              saved_frames = st->frames_processed;
    auto roll_back = [&](int rc) {
        st->buffer_pre = saved_pre;
        st->buffer_pos = saved_pos;
        st->samples_processed = saved_samples;
        st->frames_processed = saved_frames;
        *frames_output = 0;
        return rc;
    };
    const int first = st->frames_processed;
    auto flush = [&]() {                                                                    // OutputFrame :256-269
        if (st->buffer_pos != hca::SPF) return;
        st->buffer_pos = 0;
        st->frames_processed++;
    };
    int pcm_pos = 0;
    if (st->buffer_pre > 0) {                                                               // EncodePreAudio :163-183
        while (st->buffer_pre > hca::SPF) {
            st->buffer_pos = hca::SPF;
            flush();
            st->buffer_pre -= hca::SPF;
        }
        st->buffer_pos = st->buffer_pre;
        st->buffer_pre = 0;
    }
    while (hca::SPF - pcm_pos > 0 && h.sample_count > st->samples_processed) {              // EncodeMainAudio :185-200
        int n = std::min(hca::SPF - st->buffer_pos, hca::SPF - pcm_pos);
        n = std::min(n, h.sample_count - st->samples_processed);
        st->buffer_pos += n;
        st->samples_processed += n;
        pcm_pos += n;
        flush();
    }
    if (h.sample_count == st->samples_processed) {                                          // EncodePostAudio :202-242
        int post_pos = 0;
        while (post_pos < st->post_samples) {
            const int n = std::min(hca::SPF - st->buffer_pos, st->post_samples - post_pos);
            st->buffer_pos += n;
            post_pos += n;
            flush();
        }
        while (st->frames_processed < h.frame_count) {
            st->buffer_pos = hca::SPF;
            flush();
        }
    }
    const int count = st->frames_processed - first;
    *frames_output = count;
    if (count == 0) return VGA_OK;
    const uint16_t *pow = nullptr;
    if (int rc = crc_pow_table(&pow)) return roll_back(rc);
    const int64_t ch_pitch = (int64_t)st->chunks * hca::SPF;
    const int64_t frames_pitch = round_up((int64_t)h.frame_count * h.frame_size + 8, 16);
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        set_error("%s failed: %s", what, hipGetErrorString(e));
        return false;
    };
    // the status word is this call's: an error of an earlier call was reported by that call
    if (!hip_ok(hipMemsetAsync(st->d_status.p, 0, sizeof(int), st->s), "hipMemsetAsync")) return roll_back(VGA_ERR_DEVICE);
    if (int rc = hca::launch_encode(st->d_pcm.as<int16_t>(), ch_pitch * st->nch, ch_pitch, 1, st->map, st->dev, st->d_frames.as<uint8_t>(),
                                    frames_pitch, pow, st->d_status.as<int>(), st->s, first, count))
        return roll_back(rc);
    st->host_frames.resize((size_t)count * h.frame_size);
    int status = 0;
    if (!hip_ok(hipMemcpyAsync(st->host_frames.data(), st->d_frames.as<uint8_t>() + (size_t)first * h.frame_size, st->host_frames.size(),
                               hipMemcpyDeviceToHost, st->s), "hipMemcpyAsync") ||
        !hip_ok(hipMemcpyAsync(&status, st->d_status.p, sizeof(int), hipMemcpyDeviceToHost, st->s), "hipMemcpyAsync") ||
        !hip_ok(hipStreamSynchronize(st->s), "hipStreamSynchronize"))
        return roll_back(VGA_ERR_DEVICE);
    if (int rc = status_to_error(status)) return roll_back(rc);
    std::memcpy(hca_out, st->host_frames.data(), (size_t)h.frame_size);
    for (int k = 1; k < count; k++)
        st->pending.emplace_back(st->host_frames.begin() + (size_t)k * h.frame_size, st->host_frames.begin() + (size_t)(k + 1) * h.frame_size);
    return VGA_OK;
}

// CriHcaEncoder.Initialize (CriHcaEncoder.cs:61-114)
int vga_hca_encoder_initialize(const vga_hca_params *c, vga_hca_info *h)
{
    if (!c || !h) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    memset(h, 0, sizeof *h);
    if (c->channel_count > 8 || c->channel_count < 1) {
        set_error("HCA channel count must be 8 or below");
        return VGA_ERR_OUT_OF_RANGE;
    }
    if (c->sample_rate <= 0 || c->sample_count < 0) { set_error("bad sample rate / count"); return VGA_ERR_ARGUMENT; }
    const int cutoff = c->sample_rate / 2;
    h->channel_count = c->channel_count;
    h->track_count = 1;
    h->sample_count = c->sample_count;
    h->sample_rate = c->sample_rate;
    h->min_resolution = 1;
    h->max_resolution = 15;
    h->inserted_samples = hca::SPSF;
    const int bitrate = calculate_bitrate(*h, c->quality, c->bitrate, c->limit_bitrate);
    if (bitrate <= 0) { set_error("bitrate resolves to %d", bitrate); return VGA_ERR_OUT_OF_RANGE; }
    calculate_band_counts(*h, bitrate, cutoff);
    if (h->bands_per_hfr_group > 0) {                           // HcaInfo.CalculateHfrValues :52-58
        h->hfr_band_count = h->total_band_count - h->base_band_count - h->stereo_band_count;
        h->hfr_group_count = divide_by_round_up(h->hfr_band_count, h->bands_per_hfr_group);
    }
    {                                                           // SetChannelConfiguration :370-381
        const int cpt = h->channel_count / h->track_count;
        const int cfg = hosttab::HCA_DefaultChannelMapping[cpt];
        if (hosttab::HCA_ValidChannelMappings[cpt - 1][cfg] != 1) {
            set_error("Channel mapping is not valid.");
            return VGA_ERR_OUT_OF_RANGE;
        }
        h->channel_config = cfg;
    }
    int input_sample_count = h->sample_count;
    if (c->looping) {
        h->looping = 1;
        h->sample_count = std::min(c->loop_end, c->sample_count);
        h->inserted_samples += get_next_multiple(c->loop_start, hca::SPF) - c->loop_start;
        {                                                       // CalculateLoopInfo :383-398
            const int ls = c->loop_start + h->inserted_samples, le = c->loop_end + h->inserted_samples;
            h->loop_start_frame = ls / hca::SPF;
            h->pre_loop_samples = ls % hca::SPF;
            h->loop_end_frame = le / hca::SPF;
            h->post_loop_samples = hca::SPF - le % hca::SPF;
            if (h->post_loop_samples == hca::SPF) { h->loop_end_frame--; h->post_loop_samples = 0; }
        }
        input_sample_count = std::min(get_next_multiple(h->sample_count, hca::SPSF), c->sample_count);
        input_sample_count += hca::SPSF * 2;
    }
    {                                                           // CalculateHeaderSize :400-418
        h->header_size = get_next_multiple(96 + h->comment_length, 32);
        if (h->looping) {
            if (h->frame_size <= 0) {
                // the reference divides by FrameSize here (CriHcaEncoder.cs:411: a catchable DivideByZeroException);
                // the non-looping path reports the same condition from the encoder ("Bitrate is set too low.")
                set_error("Bitrate is set too low.");
                return VGA_ERR_INVALID_DATA;
            }
            const int off = h->header_size + h->frame_size * h->loop_start_frame;
            const int padding_bytes = get_next_multiple(off, 2048) - off;
            const int padding_frames = padding_bytes / h->frame_size;
            h->inserted_samples += padding_frames * hca::SPF;
            h->loop_start_frame += padding_frames;
            h->loop_end_frame += padding_frames;
            h->header_size += padding_bytes % h->frame_size;
        }
    }
    const int total_samples = input_sample_count + h->inserted_samples;
    h->frame_count = divide_by_round_up(total_samples, hca::SPF);
    h->appended_samples = h->frame_count * hca::SPF - h->inserted_samples - input_sample_count;
    return VGA_OK;
}

size_t vga_hca_decode_workspace_bytes(const vga_hca_info *h, int nstreams)
{
    if (!h || nstreams <= 0 || h->frame_count <= 0 || h->channel_count < 1 || h->channel_count > 8) return 0;
    hca::DeviceInfo d;
    if (make_device_info(*h, d) != VGA_OK) return 0;
    return hca::decode_record_bytes(d) * (size_t)h->frame_count * (size_t)nstreams;
}

int vga_hca_encode_device(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, int pcm_length,
                          const vga_hca_info *h, uint8_t *d_frames, int64_t frames_pitch, int *d_status, void *stream)
{
    if (!h) { set_error("null HcaInfo"); return VGA_ERR_ARGUMENT; }
    // fewer bits than sync + noise level + checksum + one 3-bit channel header each: the reference's
    // CalculateNoiseLevel necessarily ends in InvalidDataException (CriHcaEncoder.cs:469-472)
    if (h->channel_count >= 1 && h->channel_count <= 8 && h->frame_size * 8 < 48 + 3 * h->channel_count + 16) {
        set_error("Bitrate is set too low.");
        return VGA_ERR_INVALID_DATA;
    }
    hca::DeviceInfo d;
    if (int rc = make_device_info(*h, d)) return rc;
    if (nstreams < 0 || pcm_length < 0 || (frames_pitch & 1) || frames_pitch < (int64_t)h->frame_count * h->frame_size ||
        ch_pitch < pcm_length || stream_pitch < ch_pitch * h->channel_count) {
        set_error("bad sizes / pitches for vga_hca_encode_device");
        return VGA_ERR_ARGUMENT;
    }
    hca::PcmMap m;
    if (int rc = make_pcm_map(*h, pcm_length, m)) return rc;
    const uint16_t *pow = nullptr;
    if (int rc = crc_pow_table(&pow)) return rc;
    return hca::launch_encode(d_pcm, stream_pitch, ch_pitch, nstreams, m, d, d_frames, frames_pitch, pow,
                              d_status, (hipStream_t)stream);
}

int vga_hca_decode_device(const vga_hca_info *h, const uint8_t *d_frames, int64_t frames_pitch, int nstreams,
                          int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, void *d_workspace,
                          size_t workspace_bytes, int *d_status, void *stream)
{
    if (!h) { set_error("null HcaInfo"); return VGA_ERR_ARGUMENT; }
    hca::DeviceInfo d;
    if (int rc = make_device_info(*h, d)) return rc;
    if (nstreams < 0 || (frames_pitch & 3) || ((uintptr_t)d_frames & 3) ||
        frames_pitch < (int64_t)h->frame_count * h->frame_size + 8 || ch_pitch < h->sample_count ||
        stream_pitch < ch_pitch * h->channel_count || workspace_bytes < vga_hca_decode_workspace_bytes(h, nstreams)) {
        set_error("bad sizes / pitches / workspace for vga_hca_decode_device (frames need 4-byte alignment and 8 bytes of slack)");
        return VGA_ERR_ARGUMENT;
    }
    return hca::launch_decode(d_frames, frames_pitch, nstreams, d, d_pcm, stream_pitch, ch_pitch, d_workspace, d_status,
                              (hipStream_t)stream);
}

// streams per chunk of the host pipeline (host_pipeline.hpp): 256 stereo streams x 2813 frames = 720 k workgroups
static constexpr int HCA_CHUNK_STREAMS = 256;

// CriHcaFormat.EncodeFromPcm16 (Formats/CriHca/CriHcaFormat.cs:34-84) for a batch of equally shaped streams.
// pcm: nstreams*channel_count planar pointers (stream-major); frames_out[s]: frame_count*frame_size bytes.
static constexpr int HCA_MIN_SHARE_STREAMS = 32;       // streams per share of a call spread over several GPUs (vga_set_devices)

static int hca_encode_batch_one(const int16_t *const *pcm, int nstreams, const vga_hca_params *p, vga_hca_info *info_out,
                                uint8_t *const *frames_out);
int vga_hca_encode_batch(const int16_t *const *pcm, int nstreams, const vga_hca_params *p, vga_hca_info *info_out,
                         uint8_t *const *frames_out)
{
    if (nstreams <= 0 || !pcm || !frames_out || !p || p->channel_count < 1 || p->channel_count > 8)
        return hca_encode_batch_one(pcm, nstreams, p, info_out, frames_out);
    const int nch = p->channel_count;
    return for_each_device_share(nstreams, HCA_MIN_SHARE_STREAMS, [&](int first, int count) {
        return hca_encode_batch_one(pcm + (size_t)first * nch, count, p, first == 0 ? info_out : nullptr, frames_out + first);
    });
}
static int hca_encode_batch_one(const int16_t *const *pcm, int nstreams, const vga_hca_params *p, vga_hca_info *info_out,
                                uint8_t *const *frames_out)
{
    vga_hca_info h;
    if (int rc = vga_hca_encoder_initialize(p, &h)) return rc;
    if (info_out) *info_out = h;
    if (nstreams < 0) { set_error("negative stream count"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    if (!pcm || !frames_out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    const int nch = h.channel_count, n = p->sample_count;
    for (int i = 0; i < nstreams * nch; i++)
        if (!pcm[i] && n > 0) { set_error("pcm[%d] is null", i); return VGA_ERR_ARGUMENT; }
    for (int i = 0; i < nstreams; i++)
        if (!frames_out[i]) { set_error("frames_out[%d] is null", i); return VGA_ERR_ARGUMENT; }
    if (int rc = require_device()) return rc;
    DevBuf d_pcm, d_frames, d_status;
    const int64_t ch_pitch = round_up(n > 0 ? n : 1, 8);
    const int64_t stream_pitch = ch_pitch * nch;
    const int64_t fbytes = (int64_t)h.frame_count * h.frame_size;
    const int64_t frames_pitch = round_up(fbytes + 8, 16);
    VGA_HIP_TRY(d_pcm.alloc((size_t)nstreams * stream_pitch * 2));
    VGA_HIP_TRY(d_frames.alloc((size_t)nstreams * frames_pitch));
    VGA_HIP_TRY(d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
    // a unit of the pipeline is a stream: channel_count input rows, one row of frames out (the reference encodes one
    // stream per task, CriHcaFormat.cs:53-81 under Cli/Batch.cs:24-25)
    pipe::Job job;
    job.units = nstreams;
    if (n > 0) {
        job.in_rows_per_unit = nch;
        job.in_rows = (const void *const *)pcm;
        job.in_row_bytes = (size_t)n * 2;
        job.d_in = d_pcm.as<char>();
        job.d_in_pitch = (size_t)ch_pitch * 2;
    }
    if (fbytes > 0) {
        job.out_rows = (void *const *)frames_out;
        job.out_row_bytes = (size_t)fbytes;
        job.d_out = d_frames.as<char>();
        job.d_out_pitch = (size_t)frames_pitch;
    }
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = vga_hca_encode_device(d_pcm.as<int16_t>() + (int64_t)first * stream_pitch, stream_pitch, ch_pitch, count, n, &h,
                                             d_frames.as<uint8_t>() + (int64_t)first * frames_pitch, frames_pitch, d_status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, HCA_CHUNK_STREAMS)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
    return status_to_error(status);
}

// CriHcaFormat.ToPcm16 (CriHcaFormat.cs:26-32 -> CriHcaDecoder.Decode, CriHcaDecoder.cs:11-29), batched.
static int hca_decode_batch_one(const vga_hca_info *h, const uint8_t *const *frames, int nstreams, int16_t *const *pcm_out);
int vga_hca_decode_batch(const vga_hca_info *h, const uint8_t *const *frames, int nstreams, int16_t *const *pcm_out)
{
    if (nstreams <= 0 || !h || !frames || !pcm_out || h->channel_count < 1 || h->channel_count > 8)
        return hca_decode_batch_one(h, frames, nstreams, pcm_out);
    const int nch = h->channel_count;
    return for_each_device_share(nstreams, HCA_MIN_SHARE_STREAMS, [&](int first, int count) {
        return hca_decode_batch_one(h, frames + first, count, pcm_out + (size_t)first * nch);
    });
}
static int hca_decode_batch_one(const vga_hca_info *h, const uint8_t *const *frames, int nstreams, int16_t *const *pcm_out)
{
    if (!h) { set_error("null HcaInfo"); return VGA_ERR_ARGUMENT; }
    hca::DeviceInfo d;
    if (int rc = make_device_info(*h, d)) return rc;
    if (nstreams < 0 || h->sample_count < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    if (!frames || !pcm_out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    const int nch = h->channel_count;
    for (int i = 0; i < nstreams; i++)
        if (!frames[i] && h->frame_count > 0) { set_error("frames[%d] is null", i); return VGA_ERR_ARGUMENT; }
    for (int i = 0; i < nstreams * nch; i++)
        if (!pcm_out[i] && h->sample_count > 0) { set_error("pcm_out[%d] is null", i); return VGA_ERR_ARGUMENT; }
    if (int rc = require_device()) return rc;
    DevBuf d_pcm, d_frames, d_status, d_ws;
    const int n = h->sample_count;
    const int64_t ch_pitch = round_up(n > 0 ? n : 1, 8);
    const int64_t stream_pitch = ch_pitch * nch;
    const int64_t fbytes = (int64_t)h->frame_count * h->frame_size;
    const int64_t frames_pitch = round_up(fbytes + 8, 16);
    VGA_HIP_TRY(d_pcm.alloc((size_t)nstreams * stream_pitch * 2));
    VGA_HIP_TRY(hipMemset(d_pcm.p, 0, (size_t)nstreams * stream_pitch * 2));
    VGA_HIP_TRY(d_frames.alloc((size_t)nstreams * frames_pitch));
    VGA_HIP_TRY(hipMemset(d_frames.p, 0, (size_t)nstreams * frames_pitch));     // the 8 bytes of slack behind every stream
    VGA_HIP_TRY(d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
    const size_t wsb = vga_hca_decode_workspace_bytes(h, nstreams);
    VGA_HIP_TRY(d_ws.alloc(wsb));
    const size_t ws_per_stream = nstreams > 0 ? wsb / (size_t)nstreams : 0;
    pipe::Job job;
    job.units = nstreams;
    if (fbytes > 0) {
        job.in_rows = (const void *const *)frames;
        job.in_row_bytes = (size_t)fbytes;
        job.d_in = d_frames.as<char>();
        job.d_in_pitch = (size_t)frames_pitch;
    }
    if (n > 0) {
        job.out_rows_per_unit = nch;
        job.out_rows = (void *const *)pcm_out;
        job.out_row_bytes = (size_t)n * 2;
        job.d_out = d_pcm.as<char>();
        job.d_out_pitch = (size_t)ch_pitch * 2;
    }
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int rc = vga_hca_decode_device(h, d_frames.as<uint8_t>() + (int64_t)first * frames_pitch, frames_pitch, count,
                                             d_pcm.as<int16_t>() + (int64_t)first * stream_pitch, stream_pitch, ch_pitch,
                                             d_ws.as<char>() + (size_t)first * ws_per_stream, (size_t)count * ws_per_stream,
                                             d_status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, HCA_CHUNK_STREAMS)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
    return status_to_error(status);
}

}  // extern "C"

// ---------------------------------------------------------------- ragged batches (VGAudio.Cli/Batch.cs:24-25: a worker per FILE)
// Every stream with its own CriHcaParameters (channel count, sample rate, length, quality, loop).  Streams that are not
// looping and differ in length only share their launches: sorted into buckets of similar length (host_batch.hpp,
// plan_buckets), zero-padded on the device to the bucket's longest and encoded with THAT stream's HcaInfo -- frame size
// and band counts come from the bitrate (CriHcaEncoder.cs:288-368), not from the length, every frame is encoded
// independently of the others (hca_encode_kernel.hip) and the encoder's own input past the end of the PCM is silence
// (:234-240), so a shorter stream's frames are the first frames of the padded one.  Looping streams (their loop audio is
// replayed behind the main audio, :209-232) only share a launch with streams of exactly their shape.
namespace {

struct HcaGroupKey {
    int quality, bitrate, limit_bitrate, channel_count, sample_rate, looping, loop_start, loop_end, exact_count;
};

int hca_group_of(std::vector<HcaGroupKey> &seen, const vga_hca_params &c)
{
    HcaGroupKey k = {c.quality, c.bitrate, c.limit_bitrate, c.channel_count, c.sample_rate, c.looping ? 1 : 0,
                     c.looping ? c.loop_start : 0, c.looping ? c.loop_end : 0, c.looping ? c.sample_count : -1};
    for (size_t i = 0; i < seen.size(); i++)
        if (memcmp(&seen[i], &k, sizeof k) == 0) return (int)i;
    seen.push_back(k);
    return (int)seen.size() - 1;
}

constexpr int64_t HCA_BUCKET_VOLUME = (int64_t)HCA_CHUNK_STREAMS * 2880000;      // padded samples per chunk and channel

// the streams of `units` (indices into the caller's arrays) that have `nch` channels: one pipelined job
int hca_encode_v_job(const std::vector<int> &units, int nch, const int16_t *const *pcm, const std::vector<size_t> &first_row,
                     const vga_hca_params *configs, const vga_hca_info *infos, uint8_t *const *frames_out)
{
    const int n = (int)units.size();
    std::vector<HcaGroupKey> keys;
    std::vector<int> group(n), length(n);
    for (int i = 0; i < n; i++) {
        group[i] = hca_group_of(keys, configs[units[i]]);
        length[i] = configs[units[i]].sample_count;
    }
    const BucketPlan plan = plan_buckets(group, length, HCA_CHUNK_STREAMS, HCA_BUCKET_VOLUME, true);
    const int chunks = (int)plan.chunk_begin.size() - 1;
    std::vector<vga_hca_info> chunk_info(chunks);
    std::vector<int64_t> pcm_base(chunks + 1, 0), fr_base(chunks + 1, 0), ch_pitch(chunks), fr_pitch(chunks);
    for (int k = 0; k < chunks; k++) {
        const int count = plan.chunk_begin[k + 1] - plan.chunk_begin[k];
        chunk_info[k] = infos[units[plan.order[plan.chunk_begin[k + 1] - 1]]];       // the bucket's longest stream
        ch_pitch[k] = round_up(std::max(plan.chunk_length[k], 1), 8);
        fr_pitch[k] = round_up((int64_t)chunk_info[k].frame_count * chunk_info[k].frame_size + 8, 16);
        pcm_base[k + 1] = pcm_base[k] + ch_pitch[k] * nch * count;
        fr_base[k + 1] = fr_base[k] + fr_pitch[k] * count;
    }
    std::vector<const void *> in_rows((size_t)n * nch);
    std::vector<void *> out_rows(n);
    std::vector<size_t> in_size((size_t)n * nch), in_off((size_t)n * nch), out_size(n), out_off(n);
    size_t max_in = 16, max_out = 16;
    for (int k = 0; k < chunks; k++)
        for (int i = plan.chunk_begin[k]; i < plan.chunk_begin[k + 1]; i++) {
            const int u = units[plan.order[i]], j = i - plan.chunk_begin[k];
            for (int c = 0; c < nch; c++) {
                in_rows[(size_t)i * nch + c] = pcm[first_row[u] + c];
                in_size[(size_t)i * nch + c] = (size_t)configs[u].sample_count * 2;
                in_off[(size_t)i * nch + c] = (size_t)(pcm_base[k] + ((int64_t)j * nch + c) * ch_pitch[k]) * 2;
            }
            out_rows[i] = frames_out[u];
            out_size[i] = (size_t)infos[u].frame_count * infos[u].frame_size;
            out_off[i] = (size_t)(fr_base[k] + j * fr_pitch[k]);
            max_in = std::max(max_in, (size_t)ch_pitch[k] * 2);
            max_out = std::max(max_out, (size_t)fr_pitch[k]);
        }
    DevBuf d_pcm, d_frames, d_status;
    VGA_HIP_TRY(d_pcm.alloc((size_t)pcm_base[chunks] * 2 + 64));
    VGA_HIP_TRY(hipMemset(d_pcm.p, 0, (size_t)pcm_base[chunks] * 2 + 64));           // silence behind every row
    VGA_HIP_TRY(d_frames.alloc((size_t)fr_base[chunks] + 64));
    VGA_HIP_TRY(d_status.alloc(sizeof(int)));
    VGA_HIP_TRY(hipMemset(d_status.p, 0, sizeof(int)));
    pipe::Job job;
    job.units = n;
    job.chunk_begin = plan.chunk_begin;
    job.in_rows_per_unit = nch;
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
    job.d_out = d_frames.as<char>();
    job.compute = [&](int first, int count, hipStream_t s, std::string &why) -> int {
        const int k = plan.chunk_of(first);
        int rc = VGA_OK;
        if (chunk_info[k].frame_count > 0)
            rc = vga_hca_encode_device(d_pcm.as<int16_t>() + pcm_base[k], ch_pitch[k] * nch, ch_pitch[k], count, plan.chunk_length[k],
                                       &chunk_info[k], d_frames.as<uint8_t>() + fr_base[k], fr_pitch[k], d_status.as<int>(), s);
        if (rc) why = vga_last_error();
        return rc;
    };
    if (int rc = run_batch_pipeline(job, HCA_CHUNK_STREAMS)) return rc;
    int status = 0;
    VGA_HIP_TRY(hipMemcpy(&status, d_status.p, sizeof(int), hipMemcpyDeviceToHost));
    return status_to_error(status);
}

}  // namespace

extern "C" {

int vga_hca_encode_batch_v(const int16_t *const *pcm, int nstreams, const vga_hca_params *configs, vga_hca_info *infos_out,
                           uint8_t *const *frames_out)
{
    if (nstreams < 0) { set_error("negative stream count"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    if (!pcm || !configs || !infos_out || !frames_out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    std::vector<size_t> first_row(nstreams);
    size_t rows = 0;
    for (int s = 0; s < nstreams; s++) {
        if (int rc = vga_hca_encoder_initialize(&configs[s], &infos_out[s])) return rc;
        first_row[s] = rows;
        rows += (size_t)configs[s].channel_count;
        if (!frames_out[s] && infos_out[s].frame_count > 0) { set_error("frames_out[%d] is null", s); return VGA_ERR_ARGUMENT; }
        for (int c = 0; c < configs[s].channel_count; c++)
            if (!pcm[first_row[s] + c] && configs[s].sample_count > 0) { set_error("stream %d channel %d is null", s, c); return VGA_ERR_ARGUMENT; }
    }
    if (int rc = require_device()) return rc;
    // a pipeline job has one row count per unit: one job per channel count
    for (int nch = 1; nch <= 8; nch++) {
        std::vector<int> units;
        for (int s = 0; s < nstreams; s++)
            if (configs[s].channel_count == nch) units.push_back(s);
        if (units.empty()) continue;
        if (int rc = hca_encode_v_job(units, nch, pcm, first_row, configs, infos_out, frames_out)) return rc;
    }
    return VGA_OK;
}

int vga_hca_decode_batch_v(const vga_hca_info *infos, const uint8_t *const *frames, int nstreams, int16_t *const *pcm_out)
{
    if (nstreams < 0) { set_error("negative stream count"); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0) return VGA_OK;
    if (!infos || !frames || !pcm_out) { set_error("null array"); return VGA_ERR_ARGUMENT; }
    // streams of one shape (the same HcaInfo) decode together; a batch of all-different lengths is one call per stream --
    // the decoder's launches carry one frame count (hca_decode_kernels.hip)
    std::vector<size_t> first_row(nstreams);
    size_t rows = 0;
    for (int s = 0; s < nstreams; s++) {
        if (infos[s].channel_count < 1 || infos[s].channel_count > 8) { set_error("stream %d: bad channel count", s); return VGA_ERR_ARGUMENT; }
        first_row[s] = rows;
        rows += (size_t)infos[s].channel_count;
    }
    std::vector<char> done(nstreams, 0);
    for (int s = 0; s < nstreams; s++) {
        if (done[s]) continue;
        std::vector<const uint8_t *> fr;
        std::vector<int16_t *> out;
        for (int t = s; t < nstreams; t++)
            if (!done[t] && memcmp(&infos[t], &infos[s], sizeof(vga_hca_info)) == 0) {
                done[t] = 1;
                fr.push_back(frames[t]);
                for (int c = 0; c < infos[t].channel_count; c++) out.push_back(pcm_out[first_row[t] + c]);
            }
        if (int rc = vga_hca_decode_batch(&infos[s], fr.data(), (int)fr.size(), out.data())) return rc;
    }
    return VGA_OK;
}

}  // extern "C"
