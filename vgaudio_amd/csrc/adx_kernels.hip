// adx_kernels.hip -- CRI ADX 4-bit ADPCM encode / decode for gfx950.
//
// Replaces VGAudio/Codecs/CriAdx/CriAdxCodec.cs:56-147 (Encode, EncodeFrame, CalculateScale,
// ScaleShortToNibble) and :9-54 (Decode), bit-exact (C# int32 wrap-around via -fwrapv, the one
// f64 multiply `(int)(rawDistance * gain)` kept literally, no FMA contraction).
//
// ADX is a serial recurrence per channel with a single predictor and no retry loop, so the
// decomposition is lane = channel (the reference's Parallel.For over channels,
// Formats/CriAdx/CriAdxFormat.cs:67 / :37).  Frame size is a run-time parameter.
#include "common.hpp"
#include "adx_kernels.hpp"

namespace vga {
namespace adx {

__device__ __forceinline__ int clamp16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int clamp4(int v) { return min(max(v, -8), 7); }

// Utilities/Helpers.cs:146-163 == floor(log2(v)) for v >= 1
__device__ __forceinline__ int log2_floor(int v) { return 31 - __builtin_clz((unsigned)v); }

// CriAdxCodec.cs:149-165
__device__ __forceinline__ int calculate_scale(int max_distance, bool exponential, double &gain, int &scale_to_write)
{
    int scale = (max_distance - 1) / 7 + 1;
    if (scale > 0x1000) scale = 0x1000;
    scale_to_write = scale - 1;
    if (exponential) {
        const int power = scale_to_write == 0 ? 0 : log2_floor(scale_to_write) + 1;
        scale = 1 << power;
        scale_to_write = 12 - power;
        max_distance = 8 * scale - 1;
    }
    gain = max_distance == 0 ? 0.0 : 32767.0 / (double)max_distance;
    return scale;
}

// CriAdxCodec.cs:167-171
__device__ __forceinline__ int scale_short_to_nibble(int sample)
{
    const int sign = (sample > 0) - (sample < 0);
    sample = (sample + 2340 * sign) / 4681;     // short.MaxValue/14, short.MaxValue/7
    return clamp4(sample);
}

// Encode (CriAdxCodec.cs:56-105): the stream the frames are cut from is `padding` untouched
// (zero) buffer slots followed by the PCM, zero padded at the end; frames lying entirely inside
// the padding are skipped (their bytes stay zero, :86).
__global__ __launch_bounds__(64) void adx_encode_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int pcm_length, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, int16_t *__restrict__ history_out)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint16_t *dst = reinterpret_cast<uint16_t *>(out + (int64_t)ch * out_pitch);

    const int spf = (p.frame_size - 2) * 2;
    const int sample_count = pcm_length + p.padding;
    const int frame_count = (sample_count + spf - 1) / spf;
    const int c0 = p.coef0, c1 = p.coef1;
    const int words_per_frame = p.frame_size / 2;

    int h0 = 0, h1 = 0;                       // pcmBuffer[0], pcmBuffer[1]
    int hist = p.history;
    if (p.version == 4 && p.padding == 0 && pcm_length > 0) {
        h0 = h1 = src[0];                     // :69-74
        hist = src[0];
    }
    if (history_out) history_out[ch] = (int16_t)hist;

    for (int i = 0; i < frame_count; i++) {
        uint16_t *frame = dst + (int64_t)i * words_per_frame;
        const int t0 = i * spf;                                   // first stream position of this frame
        if (min(t0 + spf, sample_count) <= p.padding) {           // whole frame is padding: skipped (:86)
            for (int w = 0; w < words_per_frame; w++) frame[w] = 0;
            continue;
        }
        // stream position t -> sample: 0 inside the padding and past the end
        auto sample_at = [&](int j) -> int {
            const int idx = t0 + j - p.padding;
            return (idx >= 0 && idx < pcm_length) ? (int)src[idx] : 0;
        };

        // pre-scan :112-118 (raw inputs, reconstructed history)
        int max_distance = 0;
        {
            int a = h0, b = h1;
            for (int j = 0; j < spf; j++) {
                const int x = sample_at(j);
                const int predicted = ((b * c0) >> 12) + ((a * c1) >> 12);
                int distance = clamp16(x - predicted);
                distance = distance < 0 ? -distance : distance;
                max_distance = max(max_distance, distance);
                a = b;
                b = x;
            }
        }
        double gain;
        int scale_out;
        const int scale = calculate_scale(max_distance, p.type == 4, gain, scale_out);

        // header :140-141, + filter bits for the Fixed type :95
        int b0 = (scale_out >> 8) & 0x1f;
        if (p.type == 2) b0 |= (p.filter << 5) & 0xff;
        frame[0] = (uint16_t)(b0 | ((scale_out & 0xff) << 8));

        // quantise :122-138
        int a = h0, b = h1;
        uint32_t word = 0;
        for (int j = 0; j < spf; j++) {
            const int x = sample_at(j);
            int predicted = ((b * c0) >> 12) + ((a * c1) >> 12);
            const int raw = x - predicted;
            const int scaled = clamp16((int)((double)raw * gain));
            const int q = scale_short_to_nibble(scaled);
            const int decoded_distance = clamp16(scale * q);
            if (p.version == 4) predicted = (b * c0 + a * c1) >> 12;
            const int rec = clamp16(decoded_distance + predicted);
            a = b;
            b = rec;
            // bytes are (even<<4 | odd&15); two bytes per little-endian u16
            const int sh = ((j & 2) ? 8 : 0) + ((j & 1) ? 0 : 4);
            word |= (uint32_t)(q & 0xF) << sh;
            if ((j & 3) == 3) {
                frame[1 + (j >> 2)] = (uint16_t)word;
                word = 0;
            }
        }
        h0 = a;                                                    // :98-99
        h1 = b;
    }
}

// Decode (CriAdxCodec.cs:9-54)
__global__ __launch_bounds__(64) void adx_decode_kernel(
    const uint8_t *__restrict__ adpcm, int64_t in_pitch, int nch, int sample_count, AdxDeviceParams p,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ status)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const uint8_t *src = adpcm + (int64_t)ch * in_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
    const int spf = (p.frame_size - 2) * 2;
    const int frame_count = (sample_count + spf - 1) / spf;

    int hist1 = p.history, hist2 = p.history;
    int current = 0;
    int start_sample = p.padding > 0 ? p.padding % spf : 0;
    int64_t in_index = (int64_t)(p.padding / spf) * p.frame_size;
    bool bad = false;

    for (int i = 0; i < frame_count; i++) {
        const int hb0 = src[in_index], hb1 = src[in_index + 1];
        int filter_num = ((hb0 >> 4) & 0xF) >> 1;
        int cf0, cf1;
        if (p.type == 2) {
            // CriAdxCodec.cs:186-191; an index past the table throws in the reference
            if (filter_num > 3) { bad = true; filter_num = 3; }
            cf0 = filter_num == 0 ? 0 : (filter_num == 1 ? 0x0F00 : (filter_num == 2 ? 0x1CC0 : 0x1880));
            cf1 = filter_num == 0 ? 0 : (filter_num == 1 ? 0 : (filter_num == 2 ? (int)(int16_t)0xF300 : (int)(int16_t)0xF240));
        } else {
            if (filter_num > 0) bad = true;
            cf0 = p.coef0;
            cf1 = p.coef1;
        }
        int scale = (int)(int16_t)(((hb0 << 8) | hb1) & 0x1FFF);
        scale = (int)(int16_t)(p.type == 4 ? (1 << ((12 - scale) & 31)) : scale + 1);
        in_index += 2 + start_sample / 2;

        const int to_read = min(spf, sample_count - current);
        for (int s = start_sample; s < to_read; s++) {
            const int byte = src[in_index];
            int sample = (s & 1) ? (byte & 0xF) : (byte >> 4);
            if (s & 1) in_index++;
            sample = (sample ^ 8) - 8;
            if (p.version == 4)
                sample = scale * sample + ((hist1 * cf0 + hist2 * cf1) >> 12);
            else
                sample = scale * sample + ((hist1 * cf0) >> 12) + ((hist2 * cf1) >> 12);
            const int fin = clamp16(sample);
            hist2 = hist1;
            hist1 = fin;
            dst[current++] = (int16_t)fin;
        }
        start_sample = 0;
    }
    for (; current < sample_count; current++) dst[current] = 0;     // `new short[sampleCount]` tail
    if (bad && status) atomicOr(status, 1);
}

int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length, const AdxDeviceParams &p,
                  uint8_t *d_out, int64_t out_pitch, int16_t *d_history_out, hipStream_t stream)
{
    if (nch <= 0) return VGA_OK;
    hipLaunchKernelGGL(adx_encode_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_pcm, pcm_pitch, nch, pcm_length,
                       p, d_out, out_pitch, d_history_out);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_decode(const uint8_t *d_adpcm, int64_t in_pitch, int nch, int sample_count, const AdxDeviceParams &p,
                  int16_t *d_pcm, int64_t pcm_pitch, int *d_status, hipStream_t stream)
{
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    hipLaunchKernelGGL(adx_decode_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_adpcm, in_pitch, nch,
                       sample_count, p, d_pcm, pcm_pitch, d_status);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace adx
}  // namespace vga
