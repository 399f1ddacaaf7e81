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

#include <cstdlib>
#include <type_traits>

#ifndef VGA_ADX_ABLATE
#define VGA_ADX_ABLATE 0
#endif

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

// (int)double the way RyuJIT x64 does it (cvttsd2si): out of range -> int.MinValue.  v_cvt_i32_f64 saturates,
// which differs for a POSITIVE overflow only (rawDistance * gain >= 2^31: needs max_distance <= 8 and a
// reconstruction far off the input -- not seen on audio, but the reference's answer is defined).
__device__ __forceinline__ int trunc_i32_ryujit(double v)
{
    const int i = (int)v;
    return v >= 2147483648.0 ? (int)0x80000000 : i;
}

// CriAdxCodec.cs:167-171
__device__ __forceinline__ int scale_short_to_nibble(int sample)
{
    const int sign = (sample > 0) - (sample < 0);
    sample = (sample + 2340 * sign) / 4681;     // short.MaxValue/14, short.MaxValue/7
    return clamp4(sample);
}

// One sample of the quantise recurrence (CriAdxCodec.cs:122-138) for the 18-byte-frame kernels, arranged so that the
// chain from the newest reconstructed sample `b` to the next one is ten instructions (the encoder wave's rate is what
// these kernels run at).  Returns u = q + 7 (0..14); (a, b) move on one sample.  Needs |c0|, |c1| <= 16384 (launch_encode
// checks; the reference's coefficients are at most 8192, CriAdxCodec.cs:173-191).
//  * rawDistance = (x - (a c1 >> 12)) - (b c0 >> 12) = (((x - (a c1 >> 12)) << 12) + 4095 - b c0) >> 12: subtracting a floor
//    is adding the ceiling of the negative, and ceil(n / 4096) = floor((n + 4095) / 4096); below 2^31 for such coefficients;
//  * ScaleShortToNibble (:167-171), truncating division of s + 2340 sign(s) by 4681, is floor((s + 2340) / 4681) for
//    either sign (-floor((|s| + 2340) / d) = ceil((s - 2340) / d) = floor((s - 2340 + d - 1) / d), d - 1 = 4680); with
//    t = s + 2340 + 7 * 4681 >= 2339 that is (t * 57346 >> 28) - 7: 57346 * 4681 = 2^28 + 1170, exact while
//    4680 * 57346 + 1170 k < 2^28 (k = t / 4681 <= 48; here <= 14), and t * 57346 < 2^32.  -7 .. 7: Clamp4 cannot bind, nor
//    the Clamp16 of scale * q (scale <= 4096);
//  * the sample is Clamp16(scale * q + predicted) = Clamp16(scale * u + (predicted - 7 scale)).
template <bool V4, bool GUARD>
__device__ __forceinline__ int adx_quantise_step(int x, int &a, int &b, int c0, int c1, double gain, int scale, int scale7)
{
    const int ac1 = __mul24(a, c1);                          // the older sample's share: ready a step early
    const int xb = x - (ac1 >> 12);
    const int k = (xb << 12) + 4095;
    const int raw = (__mul24(b, -c0) + k) >> 12;
    const double prod = (double)raw * gain;
    const int scaled = clamp16(GUARD ? trunc_i32_ryujit(prod) : (int)prod);
    const unsigned u = (unsigned)(__mul24(scaled, 57346) + 35107 * 57346) >> 28;
    const int predicted = V4 ? (__mul24(b, c0) + ac1) >> 12 : (__mul24(b, c0) >> 12) + (ac1 >> 12);
    const int rec = clamp16(__mul24(scale, (int)u) + (predicted - scale7));
    a = b;
    b = rec;
    return (int)u;
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
            const int scaled = clamp16(trunc_i32_ryujit((double)raw * gain));
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

// 16 bytes to / from any 2-byte boundary (the rows of a padded stream, see adx_encode_fs18_direct_kernel): one
// global_store_dwordx4 / global_load_dwordx4 either way, the type only tells hipcc not to assume more
typedef int adx_i32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ void adx_store16(int16_t *q, int4 v)
{
    adx_i32x4_a2 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<adx_i32x4_a2 *>(q) = t;
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

// ---------------------------------------------------------------- 18-byte frames: lane = channel, one wave per 64 channels and piece
// CriAdxCodec.Decode (CriAdxCodec.cs:9-54) for the common shape (18-byte frames, no padding), without helper waves.
// Time segments (blockIdx.y), as in gc_decode_kernel.hip: a channel's stream is cut into pieces of `seg_frames` frames (an
// even number: frame parity decides the load alignment) decoded side by side, every piece but the first from a guessed
// history -- (0, 0), 512 frames before the piece (see the warm-up below); adx_decode_fs18_fixup_kernel then closes the seams.
// A lane reads its own frames two at a time (36 contiguous bytes from a dword boundary, the next pair in flight during
// this one), takes the nibbles out of the loaded dwords with one v_bfe_i32 each, runs the recurrence (:36-45) and hands
// its samples to the wave's LDS block, from which they leave as whole lines (below).
// Rounds 1-2 had a serial wave + three helper waves here (the helpers unpacked scale * nibble into LDS tiles and wrote the
// samples out): 100 KB of LDS per 64 channels, so one workgroup per CU, four pieces per channel at 4096 channels and a
// quarter of the SIMDs busy -- 15.1 ms at configs[2].  This kernel: 8.0 ms, of which the recurrence is free: a build that
// skips it takes 7.7 ms: the kernel is bound by its stores -- 23.6 GB at 3.0 TB/s, 43 % of what a plain fill reaches on
// this box (tools/bench_fill.py): 65 536 slow sequential streams, one per channel and piece (LABNOTES.md 4.3).
constexpr int ADX_DECODE_WARM_FRAMES = 512;           // even: a piece's frames keep their alignment
constexpr int ADX_DECODE_SLOW_SEAM = 1024;            // frames a seam may stay open before it counts as slow (a multiple of 128)
constexpr int ADX_DECODE_TAIL_BUDGET = 2048;          // frames one lane of the tail kernel decodes again before it hands over
template <bool V4, bool REPAIR>                       // (REPAIR: a name of its own in profiles, as gc_decode_direct_kernel's)
__global__ __launch_bounds__(64) void adx_decode_fs18_direct_kernel(
    const uint8_t *__restrict__ adpcm, int64_t in_pitch, int nch, int total_samples, int seg_frames, AdxDeviceParams p,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ status, const int *__restrict__ first_open,
    const int *__restrict__ slow_seams)
{
    const int ch_raw = blockIdx.x * 64 + threadIdx.x;
    const bool live = ch_raw < nch;
    const int ch = live ? ch_raw : nch - 1;
    // REPAIR launch (round 5, as gc_decode_direct_kernel's): many seams of the batch would not close
    // (tones, clipped waves); the wave decodes its 64 channels again as one piece from the first piece any of them left
    // open, from the samples before it.
    constexpr bool repair = REPAIR;
    int repair_piece = 0;
    if (repair) {
        if (slow_seams[0] < slow_seams[1]) return;                        // few: adx_decode_fs18_tail_kernel has them
        int k = live ? first_open[ch] : 0x7f000000;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) k = min(k, __shfl_xor(k, o));
        if (k <= 0 || k >= 0x7f000000) return;
        repair_piece = k;
    }
    const int64_t first_frame = (int64_t)(repair ? repair_piece : (int)blockIdx.y) * seg_frames;   // even (seg_frames is)
    if (repair) seg_frames = 0x7fffff00 / 32;                              // ... to the end of the stream
    if (first_frame * 32 >= total_samples) return;
    const int sample_count = (int)((int64_t)total_samples - first_frame * 32 < (int64_t)seg_frames * 32
                                       ? (int64_t)total_samples - first_frame * 32 : (int64_t)seg_frames * 32);
    const int frame_count = (sample_count + 31) / 32;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(adpcm + (int64_t)ch * in_pitch + first_frame * 18);
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch + first_frame * 32;
    int hist1 = blockIdx.y > 0 ? 0 : p.history, hist2 = hist1;             // later pieces: the guess (0, 0)
    if (repair) {
        hist1 = dst[-1];
        hist2 = dst[-2];
    }
    bool bad = false;
    // Output: lane = channel holds one 64-byte line per frame; stored as it is, every store instruction would touch 64
    // rows, 16 bytes of each (measured: 20 of the kernel's 23 ms).  TURN frames of every channel (TURN x 64 contiguous
    // bytes of its row) are collected in the wave's own LDS and written out turned: LPR lanes per row, 64 / LPR rows per
    // store instruction.  One wave per workgroup and LDS operations of a wave complete in order: no barrier.
    constexpr int TURN = 2;                                                // (4 and 8 frames per block: no faster)
    constexpr int LPR = TURN * 4;                                          // lanes (16 bytes each) per row
    constexpr int RPI = 64 / LPR;                                          // rows per store instruction
    __shared__ int4 s_turn[64 * (LPR + 1)];                                // rows one int4 apart from a multiple of 8: no conflicts
    const int lane = threadIdx.x;
    // Rows past the last channel: those lanes decode channel nch - 1 again (`ch` above), so their lines ARE that channel's
    // and go to its row once more -- which keeps the stores unconditional: behind a branch hipcc can no longer count them
    // and waits for ALL outstanding stores before it touches the prefetched frames (s_waitcnt vmcnt counts both).
    auto turned_row = [&](int i) {
        const int c = blockIdx.x * 64 + lane / LPR + RPI * i;
        return pcm + (int64_t)(c < nch ? c : nch - 1) * pcm_pitch + first_frame * 32 + (lane % LPR) * 8;
    };
    // one frame whose 18 bytes are w[0 .. 4] (little-endian dwords, two bytes of slack)
    auto decode_frame = [&](const uint32_t (&w)[5], auto mode_tag, int frame, int valid, int skip = 0) {
        constexpr int MODE = decltype(mode_tag)::value;          // 0: whole frame, stored turned; 1: whole frame, stored by its lane; 2: partial; 3: not stored (warm-up)
                                                                 // 4: samples [skip, valid) only: the frame in which a padded stream's samples begin
        const int hb0 = w[0] & 0xff, hb1 = (w[0] >> 8) & 0xff;
        int filter_num = ((hb0 >> 4) & 0xF) >> 1;
        int cf0, cf1;
        if (p.type == 2) {
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
        int o[32];
#pragma unroll
        for (int s = 0; s < 32; s++) {
            const int b = 2 + (s >> 1);                                    // the byte that holds sample s: high nibble first
            const int nib = __builtin_amdgcn_sbfe((int)w[b >> 2], 8 * (b & 3) + ((s & 1) ? 0 : 4), 4);
            int sample;
            if (V4) {                                  // :38-39
                int rest = __mul24(hist2, cf1);
                asm("" : "+v"(rest));
                sample = __mul24(scale, nib) + ((__mul24(hist1, cf0) + rest) >> 12);
            } else {                                   // :41-42
                int rest = (__mul24(hist2, cf1) >> 12) + __mul24(scale, nib);
                asm("" : "+v"(rest));
                sample = (__mul24(hist1, cf0) >> 12) + rest;
            }
            const int fin = clamp16(sample);
            if (MODE != 4 || s >= skip) {              // (a padded stream's first samples are not decoded at all, :21-33)
                hist2 = hist1;                         // a partial last frame runs on: nothing reads the history after it
                hist1 = fin;
            }
            o[s] = fin;
        }
        if (MODE == 0) {
            int4 *mine = s_turn + lane * (LPR + 1) + (frame % TURN) * 4;
#pragma unroll
            for (int q = 0; q < 4; q++)
                mine[q] = make_int4((o[8 * q] & 0xFFFF) | (o[8 * q + 1] << 16), (o[8 * q + 2] & 0xFFFF) | (o[8 * q + 3] << 16),
                                    (o[8 * q + 4] & 0xFFFF) | (o[8 * q + 5] << 16), (o[8 * q + 6] & 0xFFFF) | (o[8 * q + 7] << 16));
            if (frame % TURN == TURN - 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < LPR; i++)            // read, wait, store -- one at a time: LPR stores back to back were
                    adx_store16(turned_row(i) + (int64_t)(frame - (TURN - 1)) * 32,                       // 5 ms slower
                                s_turn[(lane / LPR + RPI * i) * (LPR + 1) + lane % LPR]);
                asm volatile("" ::: "memory");
            }
        } else if (MODE == 1) {
            if (live) {
                int16_t *d = dst + (int64_t)frame * 32;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    adx_store16(d + 8 * q,
                                make_int4((o[8 * q] & 0xFFFF) | (o[8 * q + 1] << 16), (o[8 * q + 2] & 0xFFFF) | (o[8 * q + 3] << 16),
                                          (o[8 * q + 4] & 0xFFFF) | (o[8 * q + 5] << 16), (o[8 * q + 6] & 0xFFFF) | (o[8 * q + 7] << 16)));
            }
        } else if ((MODE == 2 || MODE == 4) && live) {
            int16_t *d = dst + (int64_t)frame * 32;
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++)
                if (s2 >= skip && s2 < valid) d[s2] = (int16_t)o[s2];
        }
    };
    // one frame of the row by its index (it starts on a dword for even i, two bytes after one for odd i)
    auto load_frame = [&](int i, uint32_t (&w)[5]) {
        const uint32_t *f = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint16_t *>(src) + (int64_t)i * 9 - (i & 1));
        uint32_t t[5];
#pragma unroll
        for (int q = 0; q < 5; q++) t[q] = f[q];
        if (i & 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) w[q] = (t[q] >> 16) | (t[q + 1] << 16);
            w[4] = t[4] >> 16;
        } else {
#pragma unroll
            for (int q = 0; q < 5; q++) w[q] = t[q];
        }
    };
    // ---- Round 6: the head of a PADDED stream (launch_decode: `pcm` arrives moved back by the padding and total_samples counts
    // it, as in the encoder): the frames before the one the padding ends in are not read at all, that frame is decoded from
    // its nibble p.padding % 32 on, from the stream's start history (CriAdxCodec.cs:21-33); a frame at a time up to the first
    // even frame behind it, where the pairs below take over.
    int head_frames = 0;
    if (!repair && blockIdx.y == 0 && p.padding > 0) {
        const int fp = p.padding / 32;
        head_frames = fp + 1 + ((fp + 1) & 1);
        if (head_frames > frame_count) head_frames = frame_count;
#pragma unroll 1
        for (int i = fp; i < head_frames; i++) {
            uint32_t w[5];
            load_frame(i, w);
            const int valid = min(32, sample_count - i * 32);
            decode_frame(w, std::integral_constant<int, 4>{}, i, valid, i == fp ? p.padding % 32 : 0);
        }
    }
    // ---- warm-up of a later piece: the WARM frames before it are decoded from the guess (0, 0) and not stored, so that the
    // piece itself starts from a history that has, as a rule, already fallen into step with the true run (the decoder
    // forgets a wrong history within 2000 samples on audio): its seam then closes on the first frame the fix-up launch
    // checks (15 seams per channel at configs[2]: 3.7 ms of fix-up without this, against 7.9 ms for the decode itself)
    if (!repair && blockIdx.y > 0) {
        const int warm = (int)(first_frame < ADX_DECODE_WARM_FRAMES ? first_frame : ADX_DECODE_WARM_FRAMES);      // even
        const uint32_t *wsrc = src - (int64_t)warm / 2 * 9;
#pragma unroll 1
        for (int k = 0; k < warm / 2; k++) {
            uint32_t c9[9];
#pragma unroll
            for (int q = 0; q < 9; q++) c9[q] = wsrc[(int64_t)k * 9 + q];
            const uint32_t a[5] = {c9[0], c9[1], c9[2], c9[3], c9[4]};
            const uint32_t b[5] = {(c9[4] >> 16) | (c9[5] << 16), (c9[5] >> 16) | (c9[6] << 16), (c9[6] >> 16) | (c9[7] << 16),
                                   (c9[7] >> 16) | (c9[8] << 16), c9[8] >> 16};
            decode_frame(a, std::integral_constant<int, 3>{}, 0, 32);
            decode_frame(b, std::integral_constant<int, 3>{}, 0, 32);
        }
    }
    // ---- whole pairs of full frames: 36 bytes from a dword boundary, the next pair's loads in flight meanwhile
    const int full_pairs = (sample_count / 32) / TURN * (TURN / 2);          // whole blocks of TURN full frames
    uint32_t cur[9], nxt[9];
    const int first_pair = head_frames / 2;                                // (0 but for a padded stream's first piece)
    {
        const uint32_t *f = src + (int64_t)min(first_pair, max(full_pairs - 1, 0)) * 9;   // the first pair (or, with no pair at all,
        const int nq = full_pairs > 0 ? 9 : 5;                             // five dwords of the row's first frame: unused)
#pragma unroll
        for (int q = 0; q < 9; q++) cur[q] = q < nq ? f[q] : 0u;
        // the first pair is waited for HERE: left to the loop header, the wait would also sit on the back edge, where it
        // means "every store of the pair before has completed"
#pragma unroll
        for (int q = 0; q < 9; q++) asm volatile("" : "+v"(cur[q]));
    }
#pragma unroll 1
    for (int k = first_pair; k < full_pairs; k++) {
        const uint32_t *f = src + (int64_t)min(k + 1, full_pairs - 1) * 9;
#pragma unroll
        for (int q = 0; q < 9; q++) nxt[q] = f[q];
        const uint32_t a[5] = {cur[0], cur[1], cur[2], cur[3], cur[4]};
        // the second frame starts two bytes into cur[4]
        const uint32_t b[5] = {(cur[4] >> 16) | (cur[5] << 16), (cur[5] >> 16) | (cur[6] << 16), (cur[6] >> 16) | (cur[7] << 16),
                               (cur[7] >> 16) | (cur[8] << 16), cur[8] >> 16};
        decode_frame(a, std::integral_constant<int, 0>{}, 2 * k, 32);
        decode_frame(b, std::integral_constant<int, 0>{}, 2 * k + 1, 32);
#pragma unroll
        for (int q = 0; q < 9; q++) cur[q] = nxt[q];
    }
    // ---- what is left of the piece: fewer than TURN full frames and a partial one
#pragma unroll 1
    for (int i = max(2 * full_pairs, head_frames); i < frame_count; i++) {
        uint32_t w[5];
        load_frame(i, w);
        const int valid = min(32, sample_count - i * 32);
        if (valid == 32) decode_frame(w, std::integral_constant<int, 1>{}, i, 32);
        else decode_frame(w, std::integral_constant<int, 2>{}, i, valid);
    }
    if (bad && live && status) atomicOr(status, 1);
}

// One frame of CriAdxCodec.Decode (:23-45) from the history (hist1, hist2) into o[0 .. valid).  `fr` = the frame's first byte:
// 2 bytes past a dword boundary for odd frames (rows are dword-aligned in these kernels); the 18 bytes arrive as five dword
// loads from the boundary at or before them and a whole frame leaves as four 16-byte stores (round 5: a byte load per two
// samples and a 2-byte store per sample until then -- 4 us a frame on a path that can walk a whole channel).
template <bool V4>
__device__ __forceinline__ void adx_decode_frame_serial(const uint8_t *fr, const AdxDeviceParams &p, int valid, int &hist1,
                                                        int &hist2, int16_t *o)
{
    const bool odd = (reinterpret_cast<uintptr_t>(fr) & 2) != 0;
    const uint32_t *f32 = reinterpret_cast<const uint32_t *>(fr - (odd ? 2 : 0));
    uint32_t t[5], w[5];
#pragma unroll
    for (int q = 0; q < 5; q++) t[q] = f32[q];
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = odd ? (t[q] >> 16) | (t[q + 1] << 16) : t[q];
    w[4] = odd ? t[4] >> 16 : t[4];
    const int hb0 = w[0] & 0xff, hb1 = (w[0] >> 8) & 0xff;
    int filter_num = ((hb0 >> 4) & 0xF) >> 1;
    int cf0, cf1;
    if (p.type == 2) {                                  // the fixed filters (CriAdxCodec.cs:186-191)
        if (filter_num > 3) filter_num = 3;
        cf0 = filter_num == 0 ? 0 : (filter_num == 1 ? 0x0F00 : (filter_num == 2 ? 0x1CC0 : 0x1880));
        cf1 = filter_num == 0 ? 0 : (filter_num == 1 ? 0 : (filter_num == 2 ? (int)(int16_t)0xF300 : (int)(int16_t)0xF240));
    } else {
        cf0 = p.coef0;
        cf1 = p.coef1;
    }
    int scale = (int)(int16_t)(((hb0 << 8) | hb1) & 0x1FFF);
    scale = (int)(int16_t)(p.type == 4 ? (1 << ((12 - scale) & 31)) : scale + 1);
    int out[32];
#pragma unroll
    for (int s2 = 0; s2 < 32; s2++) {
        const int b = 2 + (s2 >> 1);                    // the byte that holds sample s2: high nibble first
        int sample = __builtin_amdgcn_sbfe((int)w[b >> 2], 8 * (b & 3) + ((s2 & 1) ? 0 : 4), 4);
        if (V4) sample = scale * sample + ((hist1 * cf0 + hist2 * cf1) >> 12);
        else sample = scale * sample + ((hist1 * cf0) >> 12) + ((hist2 * cf1) >> 12);
        const int fin = clamp16(sample);
        if (s2 < valid) {
            hist2 = hist1;
            hist1 = fin;
        }
        out[s2] = fin;
    }
    if (valid == 32) {                                  // (a padded stream's frames start at any 2-byte boundary: adx_store16)
#pragma unroll
        for (int q = 0; q < 4; q++)
            adx_store16(o + 8 * q,
                        make_int4((out[8 * q] & 0xFFFF) | (out[8 * q + 1] << 16), (out[8 * q + 2] & 0xFFFF) | (out[8 * q + 3] << 16),
                                  (out[8 * q + 4] & 0xFFFF) | (out[8 * q + 5] << 16), (out[8 * q + 6] & 0xFFFF) | (out[8 * q + 7] << 16)));
    } else {
        for (int s2 = 0; s2 < valid; s2++) o[s2] = (int16_t)out[s2];
    }
}

// Closes the seams between time segments: one lane per (channel, seam), all seams at once.  From the history the piece
// before ended on (its last two samples: final provided THAT piece's own seam closes) decode again frame by frame
// over the guessed run's samples until both histories coincide at a frame end -- from there on the guessed run is
// what the serial decoder produces.  A seam that does not close inside its piece records its index in
// first_open[channel]; adx_decode_fs18_tail_kernel then decodes that channel serially from there.  Exact in every case.
template <bool V4>
__global__ __launch_bounds__(64) void adx_decode_fs18_fixup_kernel(
    const uint8_t *__restrict__ adpcm, int64_t in_pitch, int nch, int total_samples, int seg_frames, AdxDeviceParams p,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ first_open, int *__restrict__ seam_open, int force_open,
    int *__restrict__ slow_seams, const int *__restrict__ own_samples)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const int k = blockIdx.y + 1;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (ch >= nch || f0 * 32 >= total_samples) return;
    // own_samples (the ragged entry point's length buckets, capi_adx.hip): the channel is a shorter stream padded to
    // total_samples; a seam in its padding is nobody's output, and neither is what a run does once it has passed the
    // channel's own samples (two runs through zero frames need never meet: -1 is a fixed point of the predictor's floor)
    const int64_t own = own_samples ? (int64_t)own_samples[ch] : (int64_t)total_samples;
    if (f0 * 32 >= own) return;
    const uint8_t *src = adpcm + (int64_t)ch * in_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
    // Seed read concurrently with seam k-1's lane rewriting piece k-1 -- same invariant as gc_decode_fixup_kernel
    // (gc_decode_kernel.hip): a seam that closes leaves the piece's last samples with the values they already hold,
    // one that stays open hands pieces k.. to the tail kernel, which redoes them from the final samples.
    int hist1 = dst[f0 * 32 - 1], hist2 = dst[f0 * 32 - 2];
    // (round 5, as gc_decode_fixup_kernel: a seam still open after ADX_DECODE_SLOW_SEAM frames counts as slow; once the batch
    // holds slow_seams[1] of them the lanes give up and the REPAIR launch of the direct kernel decodes from their pieces on)
    int walked = 0;
    bool counted = false, gave_up = false;
    const bool countable = !seam_forced_open(force_open, ch, k) || force_open == 3;
    for (int64_t f = f0; f < f0 + seg_frames && f * 32 < total_samples; f++) {
        const int valid = (int)((int64_t)total_samples - f * 32 < 32 ? (int64_t)total_samples - f * 32 : 32);
        int16_t *o = dst + f * 32;
        int g1 = 0, g2 = 0;                             // the guessed run's history at this frame's end
        if (valid == 32) { g1 = o[31]; g2 = o[30]; }
        adx_decode_frame_serial<V4>(src + f * 18, p, valid, hist1, hist2, o);
        if (valid == 32 && hist1 == g1 && hist2 == g2 && !seam_forced_open(force_open, ch, k)) return;
        if (valid < 32) return;                         // the stream's last, partial frame: nothing follows
        if ((f + 1) * 32 >= own) return;                // the channel's own samples are all final
        if (++walked == ADX_DECODE_SLOW_SEAM && countable) {
            atomicAdd(&slow_seams[0], 1);
            counted = true;
        }
        if (walked >= ADX_DECODE_SLOW_SEAM && (walked & 127) == 0 &&
            __hip_atomic_load(&slow_seams[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slow_seams[1]) {
            gave_up = true;
            break;
        }
    }
    const bool piece_follows = f0 + seg_frames < ((int64_t)total_samples + 31) / 32;
    if (!counted && piece_follows && countable) atomicAdd(&slow_seams[0], 1);
    if (piece_follows || gave_up) {                     // open (and a piece follows), or this piece itself is left unfinished
        if (piece_follows) seam_open[(int64_t)(k - 1) * nch + ch] = 1;
        atomicMin(&first_open[ch], k);
    }
}

// Channels with an open seam (practically none): decode serially from the piece after it to the end of the stream.
// The channels with an open seam, piece after piece -- as gc_decode_tail_kernel (gc_decode_kernel.hip): the piece after
// an open seam is decoded again from the final samples until it agrees with what the piece holds at a frame end; a run
// that does not meet carries on into the next piece; a later open seam starts the same again.
template <bool V4>
__global__ __launch_bounds__(64) void adx_decode_fs18_tail_kernel(
    const uint8_t *__restrict__ adpcm, int64_t in_pitch, int nch, int total_samples, int seg_frames, int segments, AdxDeviceParams p,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ first_open, const int *__restrict__ seam_open,
    int force_open, int *__restrict__ slow_seams, const int *__restrict__ own_samples)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const int64_t own = own_samples ? (int64_t)own_samples[ch] : (int64_t)total_samples;       // (see the fix-up kernel)
    // many seams that would not close -- or a lane of this launch has handed a channel over (below): the REPAIR launch runs,
    // and it takes every channel whose first_open is still set, this one included
    if (__hip_atomic_load(&slow_seams[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slow_seams[1]) return;
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    int walked_total = 0;                               // frames this lane has decoded again (see ADX_DECODE_TAIL_BUDGET)
    const uint8_t *src = adpcm + (int64_t)ch * in_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
    bool carry = false;
    int hist1 = 0, hist2 = 0;
    for (int k = k0; k < segments; k++) {
        const int64_t f0 = (int64_t)k * seg_frames;
        if (f0 * 32 >= total_samples || f0 * 32 >= own) break;
        const bool flagged = seam_open[(int64_t)(k - 1) * nch + ch] != 0;
        bool apart = false;
        if (carry) {
            apart = true;
            for (int64_t f = f0; f < f0 + seg_frames && f * 32 < total_samples; f++) {
                const int valid = (int)((int64_t)total_samples - f * 32 < 32 ? (int64_t)total_samples - f * 32 : 32);
                int16_t *o = dst + f * 32;
                int g1 = 0, g2 = 0;
                if (valid == 32) { g1 = o[31]; g2 = o[30]; }
                adx_decode_frame_serial<V4>(src + f * 18, p, valid, hist1, hist2, o);
                walked_total++;
                if (valid == 32 && hist1 == g1 && hist2 == g2 && !seam_forced_open(force_open, ch, k)) { apart = false; break; }
                if ((f + 1) * 32 >= own) { apart = false; break; }          // past the channel's own samples: as good as met
            }
        }
        const int64_t f1 = f0 + seg_frames;
        if (apart) {
            carry = true;
            // A run that has not met after ADX_DECODE_TAIL_BUDGET frames (a tone, a clipped wave: it never will) is not
            // walked to the end of the stream by ONE lane: pieces up to this one are final now, the REPAIR launch decodes
            // the channel's wave from the next piece on at the direct kernel's speed (bench.py signal_sensitivity: 43
            // such channels in 4096 cost this kernel 347 ms).  Seams the test hook holds open do not count.
            if (walked_total >= ADX_DECODE_TAIL_BUDGET && f1 * 32 < total_samples && (force_open == 0 || force_open == 3)) {
                first_open[ch] = k + 1;
                atomicMax(&slow_seams[0], slow_seams[1]);
                return;
            }
        } else if (flagged && f1 * 32 < total_samples) {
            carry = true;
            hist1 = dst[f1 * 32 - 1];
            hist2 = dst[f1 * 32 - 2];
        } else
            carry = false;
    }
    first_open[ch] = 0x7f7f7f7f;                        // done: nothing of this channel is left for the REPAIR launch
}

// A frame's 32 input samples as the 16 dwords they are loaded as; sample j sign-extended (one v_bfe_i32 / v_ashrrev, or an
// SDWA operand, where it is used: 32 unpacked samples are 32 live registers)
__device__ __forceinline__ int adx_sample(const uint32_t (&xw)[16], int j)
{
    return (j & 1) ? (int)xw[j >> 1] >> 16 : (int)(int16_t)(xw[j >> 1] & 0xFFFFu);
}

// The pre-scan (CriAdxCodec.cs:112-118) of the 30 distances whose history is input only (samples 2..31): max |Clamp16(d)|
// from the two signed extremes (clamp and magnitude are monotone on either side of zero: two instructions per sample
// less than clamping each).  The encoder's pieces leave it in their crumbs, the seam runs take it from there.
__device__ __forceinline__ int adx_prescan30(const uint32_t (&xw)[16], int c0, int c1)
{
    int hi = 0, lo = 0;
#pragma unroll
    for (int j = 2; j < 32; j++) {
        // 16-bit x 16-bit: exact in 24 bits
        const int d = (adx_sample(xw, j) - (__mul24(adx_sample(xw, j - 1), c0) >> 12)) - (__mul24(adx_sample(xw, j - 2), c1) >> 12);
        hi = max(hi, d);
        lo = min(lo, d);
    }
    return max(clamp16(hi), -clamp16(lo));
}

// One frame of CriAdxCodec.EncodeFrame (:107-147) from the history (a, b); maths as adx_encode_kernel.  pm30 = adx_prescan30(xw).
// The frame leaves as its 16 header bits (low byte = the frame's first byte) and four dwords of nibbles in memory order
// (frame bytes 2..17: eight samples a dword, the first sample in the high nibble of the lowest byte).
template <bool V4, bool EXPONENTIAL>
__device__ __forceinline__ void adx_encode_frame_packed(const uint32_t (&xw)[16], int &a, int &b, int c0, int c1, int filter_bits, int pm30,
                                                        uint32_t &hdr, uint32_t (&nib)[4])
{
    int max_distance;
    {                                                    // the two distances that see the reconstructed history
        const int x0 = adx_sample(xw, 0), x1 = adx_sample(xw, 1);
        const int d0 = (x0 - (__mul24(b, c0) >> 12)) - (__mul24(a, c1) >> 12);
        const int d1 = (x1 - (__mul24(x0, c0) >> 12)) - (__mul24(b, c1) >> 12);
        max_distance = max(max(clamp16(max(d0, d1)), -clamp16(min(d0, d1))), pm30);
    }
    double gain;
    int scale_out;
    const int scale = calculate_scale(max_distance, EXPONENTIAL, gain, scale_out);
    hdr = (uint32_t)((((scale_out >> 8) & 0x1f) | filter_bits) & 0xff) | ((uint32_t)(scale_out & 0xff) << 8);
    // the quantise recurrence (:122-138) (adx_quantise_step), and the RyuJIT overflow semantics of the cast only for a frame
    // whose gain can push rawDistance past 2^31 (a wave-uniform, practically never taken branch).  Eight samples' u = q + 7
    // are gathered into a dword a nibble at a time (u <= 14: no carries), first sample on top; the nibble of q is
    // (u + 9) mod 16 = (u + 1) ^ 8 -- one add and one xor for all eight -- and memory order wants the bytes reversed.
    const int scale7 = 7 * scale;
    auto quantise = [&](auto guard_c) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_c)::value;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t acc = 0;
#pragma unroll
            for (int j = 8 * w; j < 8 * w + 8; j++)
                acc = (acc << 4) + (uint32_t)adx_quantise_step<V4, GUARD>(adx_sample(xw, j), a, b, c0, c1, gain, scale, scale7);
            nib[w] = __builtin_bswap32((acc + 0x11111111u) ^ 0x88888888u);
        }
    };
    const double raw_bound = 32770.0 + 8.0 * (double)((c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1));
    if (__any(gain * raw_bound >= 2147483648.0)) quantise(std::true_type{});
    else quantise(std::false_type{});
}

// The same as nine 16-bit words (low byte = the earlier byte of the frame): for the seam runs' 16-bit stores.
template <bool V4, bool EXPONENTIAL>
__device__ __forceinline__ void adx_encode_frame_words(const uint32_t (&xw)[16], int &a, int &b, int c0, int c1, int filter_bits, int pm30,
                                                       uint32_t (&fw)[9])
{
    uint32_t nib[4];
    adx_encode_frame_packed<V4, EXPONENTIAL>(xw, a, b, c0, c1, filter_bits, pm30, fw[0], nib);
#pragma unroll
    for (int w = 0; w < 4; w++) {
        fw[1 + 2 * w] = nib[w] & 0xFFFFu;
        fw[2 + 2 * w] = nib[w] >> 16;
    }
}

// A frame with fewer than 32 samples left (zero padded, :86-91), or any frame a sample at a time
// (first: the stream's first real sample -- the positions before it are the reference's untouched, zero, buffer slots of a
// padded stream, CriAdxCodec.cs:78-91, and `src` must not be read there)
__device__ __forceinline__ void adx_load_frame_slow(const int16_t *src, int64_t f, int total_length, uint32_t (&xw)[16], int first = 0)
{
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int64_t i0 = f * 32 + 2 * j;
        const uint32_t lo = (i0 >= first && i0 < total_length) ? (uint16_t)src[i0] : 0u;
        const uint32_t hi = (i0 + 1 >= first && i0 + 1 < total_length) ? (uint16_t)src[i0 + 1] : 0u;
        xw[j] = lo | (hi << 16);
    }
}

// ---------------------------------------------------------------- 18-byte frames: lane = channel, one wave per 64 channels and piece
// CriAdxCodec.Encode / EncodeFrame (:56-147) for the common shape, one wave on its own (no helper waves, no LDS, no barrier).
// Time segments (blockIdx.y), as the decoder above has them: every piece but the first starts from a guessed history -- the
// two INPUT samples before it -- and adx_encode_fs18_fixup_kernel closes the seams afterwards; seg_state[piece][channel] gets
// each piece's final history.  A lane reads its own frames (two at a time: a 128-byte line, the next pair's loads in flight
// during this one) and writes eight frames at a time (144 bytes from a dword boundary).
// Every frame costs the same, so a plain grid with as many waves as the chip holds is balanced: the wave's ~900
// instructions per frame (pre-scan 8 per sample, adx_quantise_step 18) are what it runs at -- configs[2]: 12.1 ms with 32
// pieces per channel, 84 % of the VALU issue slots, 23.6 GB fetched and 11.0 GB written (23.6 and 9.6 GB algorithmic, the
// crumbs included; tools/pmc_adx_encode.sh).  The tiled kernel of rounds 2-4 -- one encoder wave and three helper waves per 64 channels, the
// tiles in LDS, hence two encoder waves per CU -- took 19.7 ms, 17.5 ms with this file's arithmetic.
// Crumbs: 8 bytes per frame and channel in a scratch array [frame][channel] -- what a seam run needs to know about the
// guessed run it replaces (see adx_encode_fs18_fixup_kernel).
// Round 6: PADDED streams (CriAdxFormat.cs:59-62: every looping file whose loop start is not a multiple of the alignment gets
// Padding = alignmentSamples, up to 63 zero samples in front) take these kernels too.  They work in STREAM positions: `pcm`
// arrives moved back by the padding (launch_encode), total_length counts the padding, and a position below p.padding is
// never read: the frames of piece 0 that reach into the padding are loaded a sample at a time -- those lying wholly inside
// it are skipped, their bytes zero (:84-86) -- and nothing else comes near it.  A frame of a padded stream starts at any
// 2-byte boundary, hence the 2-byte alignment of the 16-byte loads (the same global_load_dwordx4 either way).
typedef uint32_t adx_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t adx_u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
__device__ __forceinline__ uint4 adx_load16(const int16_t *q)
{
    const adx_u32x4_a2 v = *reinterpret_cast<const adx_u32x4_a2 *>(q);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// REPAIR (round 6, the decoders' scheme): a launch of ONE piece row after the fix-up.  When many seams of the batch stayed open
// to the end of their pieces (a batch of tones or clipped waves: the run from the true history and the guessed one stay one
// LSB apart for good, LABNOTES 8.7) the chained tail kernel would walk every such channel piece after piece with one lane
// (254 ms for 60 s of a 440 Hz tone in every channel).  Here a wave that holds such a channel encodes its 64 channels again as
// ONE piece, from the first piece any of them left open to the end of the stream, from seg_state of the piece before --
// the true history for all 64 (every seam before that piece closed) -- at this kernel's own rate: the serial floor.
template <bool V4, bool EXPONENTIAL, bool REPAIR = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void adx_encode_fs18_direct_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_length, int seg_frames, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, int16_t *__restrict__ history_out, int16_t *__restrict__ seg_state,
    uint2 *__restrict__ crumbs, const int *__restrict__ first_open, const int *__restrict__ open_seams, int many)
{
    const int ch_raw = blockIdx.x * 64 + threadIdx.x;
    int k = blockIdx.y;
    if (REPAIR) {
        if (open_seams[0] < many) return;              // few: adx_encode_fs18_tail_kernel has chained them
        int ko = ch_raw < nch ? first_open[ch_raw] : 0x7f000000;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ko = min(ko, __shfl_xor(ko, o));
        if (ko <= 0 || ko >= 0x7f000000) return;       // no open seam among this wave's channels
        k = ko;
    }
    const int ch = ch_raw;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (ch >= nch || (k > 0 && f0 * 32 >= total_length)) return;
    if (REPAIR) {
        seg_frames = 0x7fffff00 / 32 - (int)f0;        // ... to the end of the stream
        crumbs = nullptr;
    }
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = out + (int64_t)ch * out_pitch;
    const int c0 = p.coef0, c1 = p.coef1;
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    int a = 0, b = 0;
    if (REPAIR) {                                      // the true history at the start of piece k
        a = seg_state[((int64_t)(k - 1) * nch + ch) * 2];
        b = seg_state[((int64_t)(k - 1) * nch + ch) * 2 + 1];
    } else if (k > 0) {                                // the guess: the input just before this piece
        a = src[f0 * 32 - 2];
        b = src[f0 * 32 - 1];
    } else {
        int hist = p.history;
        if (V4 && total_length > 0 && p.padding == 0) { a = b = src[0]; hist = a; }    // :69-74
        if (history_out) history_out[ch] = (int16_t)hist;
    }
    const int64_t frames = ((int64_t)total_length + 31) / 32, full_frames = total_length / 32;
    const int64_t fe = f0 + seg_frames < frames ? f0 + seg_frames : frames;
    // Two frames (one 128-byte line of the lane's row) are loaded together, the pair after them in flight meanwhile: the
    // halves of a line loaded a frame apart did not survive in the L1 between the two (30.5 GB fetched for 23.6).
    auto fetch2 = [&](int64_t f, uint4 (&px)[8]) {     // unconditional, clamped to the last full frames
        const int64_t fc = f + 1 < full_frames ? f : (full_frames >= 2 ? full_frames - 2 : 0);
        if (full_frames >= 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) px[i] = adx_load16(src + fc * 32 + 8 * i);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) px[i] = make_uint4(0, 0, 0, 0);
        }
    };
    auto encode_x = [&](int64_t f, const uint32_t (&xw)[16], uint32_t &hdr, uint32_t (&nib)[4]) {
#if VGA_ADX_ABLATE & 4                                  // (timing-only builds: tools/build_variants.sh)
        const int pm30 = (int)(xw[5] & 0x7FFFu);
#else
        const int pm30 = adx_prescan30(xw, c0, c1);
#endif
        adx_encode_frame_packed<V4, EXPONENTIAL>(xw, a, b, c0, c1, filter_bits, pm30, hdr, nib);
        // the crumb of this frame, for the seam that may run over it: the history this run leaves it with and the part
        // of the pre-scan that does not depend on any history (a wave's 64 crumbs are 512 contiguous bytes)
#if !(VGA_ADX_ABLATE & 3)
        if (crumbs && k > 0) crumbs[f * nch + ch] = make_uint2(((uint32_t)a & 0xFFFFu) | ((uint32_t)b << 16), (uint32_t)pm30);
#endif
    };
    auto encode = [&](int64_t f, const uint4 (&px)[8], auto half_c, uint32_t &hdr, uint32_t (&nib)[4]) {   // the pair's first or second frame
        constexpr int H = decltype(half_c)::value * 4;
        uint32_t xw[16];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            xw[4 * i] = px[H + i].x; xw[4 * i + 1] = px[H + i].y; xw[4 * i + 2] = px[H + i].z; xw[4 * i + 3] = px[H + i].w;
        }
        encode_x(f, xw, hdr, nib);
    };
    // Eight frames at a time: their 144 bytes leave as nine 16-byte stores (whole lines for the L2 to write back, where 36
    // bytes per pair of frames left partial ones: 17.0 GB written for 9.6)
    const int64_t fe8 = fe < full_frames ? fe : full_frames;                           // groups of eight need full frames
    auto encode_slow = [&](int64_t f) {                 // a frame loaded a sample at a time (zero outside the stream's samples)
        uint32_t xw[16], hdr, nib[4];
        adx_load_frame_slow(src, f, total_length, xw, p.padding);
        encode_x(f, xw, hdr, nib);
        uint16_t *d = reinterpret_cast<uint16_t *>(dst + f * 18);
        d[0] = (uint16_t)hdr;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            d[1 + 2 * i] = (uint16_t)(nib[i] & 0xFFFFu);
            d[2 + 2 * i] = (uint16_t)(nib[i] >> 16);
        }
    };
    uint4 cur[8], nxt[8];
    int64_t f = f0;
    if (!REPAIR && k == 0 && p.padding > 0) {
        // the head of a padded stream: frames up to the first even one that lies wholly behind the padding
        int64_t fh = ((int64_t)p.padding + 31) / 32;
        fh += fh & 1;
        for (; f < fh && f < fe; f++) {
            const int64_t end = (f + 1) * 32 < total_length ? (f + 1) * 32 : total_length;
            if (end <= p.padding) {                     // wholly inside the padding: skipped, its bytes stay zero (:84-86)
                uint16_t *d = reinterpret_cast<uint16_t *>(dst + f * 18);
#pragma unroll
                for (int i = 0; i < 9; i++) d[i] = 0;
            } else
                encode_slow(f);
        }
    }
    if (f + 8 <= fe8) fetch2(f, cur);
    for (; f + 8 <= fe8; f += 8) {
        uint32_t w[36];
        auto pair = [&](auto pr_c) __attribute__((always_inline)) {                    // (a lambda per pair: constant indices into w)
            constexpr int pr = decltype(pr_c)::value;
            uint32_t he, ho, ne[4], no[4];
#if VGA_ADX_ABLATE & 8                                  // no loads after the first: the same frames over and over
            if (f == f0 && pr == 0) fetch2(f + 2, nxt);
#else
            fetch2(f + 2 * pr + 2, nxt);
#endif
            encode(f + 2 * pr, cur, std::integral_constant<int, 0>{}, he, ne);
            encode(f + 2 * pr + 1, cur, std::integral_constant<int, 1>{}, ho, no);
            // 36 bytes: header, 16 bytes of nibbles, header, 16 bytes of nibbles
            uint32_t *d = w + 9 * pr;
            d[0] = he | (ne[0] << 16);
            d[1] = (ne[0] >> 16) | (ne[1] << 16);
            d[2] = (ne[1] >> 16) | (ne[2] << 16);
            d[3] = (ne[2] >> 16) | (ne[3] << 16);
            d[4] = (ne[3] >> 16) | (ho << 16);
            d[5] = no[0]; d[6] = no[1]; d[7] = no[2]; d[8] = no[3];
#if !(VGA_ADX_ABLATE & 8)
#pragma unroll
            for (int i = 0; i < 8; i++) cur[i] = nxt[i];
#endif
            __builtin_amdgcn_sched_barrier(0);          // (the next pair's work stays behind this one: registers)
        };
        pair(std::integral_constant<int, 0>{});
        pair(std::integral_constant<int, 1>{});
        pair(std::integral_constant<int, 2>{});
        pair(std::integral_constant<int, 3>{});
#if VGA_ADX_ABLATE & 2
        if (w[0] == 0x12345678u && w[35] == 0x9abcdef0u && w[17] == 77u) dst[f * 18] = 1;
#else
        adx_u32x4_a4 *d = reinterpret_cast<adx_u32x4_a4 *>(dst + f * 18);              // f - f0 is a multiple of 8, f0 even
#pragma unroll
        for (int i = 0; i < 9; i++) {
            adx_u32x4_a4 v;
            v.x = w[4 * i]; v.y = w[4 * i + 1]; v.z = w[4 * i + 2]; v.w = w[4 * i + 3];
            d[i] = v;
        }
#endif
    }
    for (; f < fe; f++) encode_slow(f);                 // what is left of the piece, the zero-padded last frame included
    if (seg_state && !REPAIR) {
        int16_t *st = seg_state + ((int64_t)k * nch + ch) * 2;
        st[0] = (int16_t)a;
        st[1] = (int16_t)b;
    }
}

// From the TRUE history (ta, tb) at the start of piece k: encode again frame by frame next to a replay of the run whose
// bytes the piece holds (decoded from ITS start history (sa, sb), before they are overwritten), until both histories
// coincide at a frame end.  Returns true when the piece ended first; (ta, tb) is then the true history at its end.
template <bool V4, bool EXPONENTIAL>
__device__ __forceinline__ bool adx_encode_seam_run(const int16_t *__restrict__ src, uint8_t *__restrict__ dst, int64_t f0, int seg_frames,
                                                    int total_length, int c0, int c1, int filter_bits, int &ta, int &tb, int sa, int sb,
                                                    int ch, int k, int force_open)
{
    // lane = channel, so every load of the wave touches 64 different rows: a frame is fetched as four 16-byte loads of PCM
    // and nine 16-bit loads of the old frame (instead of 32 + 18 scalar loads), one frame ahead of its use (clamped,
    // unconditional), and leaves as nine 16-bit stores.  All lanes of a wave are at the same frame (same seam index).
    const int64_t full_frames = total_length / 32;     // frames with all 32 samples (>= 64 here: pieces are that long at least)
    auto fetch = [&](int64_t f, uint4 (&px)[4], uint32_t (&fw)[9]) {
        const int64_t fc = f < full_frames ? f : full_frames - 1;
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = adx_load16(src + fc * 32 + 8 * i);
        const uint16_t *q = reinterpret_cast<const uint16_t *>(dst + fc * 18);
#pragma unroll
        for (int i = 0; i < 9; i++) fw[i] = q[i];
    };
    uint4 px[4], nx[4];
    uint32_t ow[9], nw[9];
    fetch(f0, px, ow);
    for (int64_t f = f0; f < f0 + seg_frames && f * 32 < total_length; f++) {
        fetch(f + 1, nx, nw);                           // in flight during this frame
        uint32_t xw[16];
        if (f < full_frames) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xw[4 * i] = px[i].x; xw[4 * i + 1] = px[i].y; xw[4 * i + 2] = px[i].z; xw[4 * i + 3] = px[i].w;
            }
        } else {                                        // the zero-padded last frame: its own loads
            adx_load_frame_slow(src, f, total_length, xw);
            const uint16_t *q = reinterpret_cast<const uint16_t *>(dst + f * 18);
#pragma unroll
            for (int i = 0; i < 9; i++) ow[i] = q[i];
        }
        // the guessed run's reconstruction of this frame (CriAdxCodec.Decode :23-45)
        int scale = (int)(int16_t)((((ow[0] & 0xff) << 8) | (ow[0] >> 8)) & 0x1FFF);
        scale = (int)(int16_t)(EXPONENTIAL ? (1 << ((12 - scale) & 31)) : scale + 1);
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int bi = 2 + (j >> 1);                // frame byte
            const int byte = (int)((ow[bi >> 1] >> (8 * (bi & 1))) & 0xff);
            int v = (j & 1) ? (byte & 0xF) : (byte >> 4);
            v = (v ^ 8) - 8;
            if (V4) v = __mul24(scale, v) + ((__mul24(sb, c0) + __mul24(sa, c1)) >> 12);
            else v = __mul24(scale, v) + (__mul24(sb, c0) >> 12) + (__mul24(sa, c1) >> 12);
            sa = sb;
            sb = clamp16(v);
        }
        uint32_t fw[9];
        adx_encode_frame_words<V4, EXPONENTIAL>(xw, ta, tb, c0, c1, filter_bits, adx_prescan30(xw, c0, c1), fw);
        uint16_t *o = reinterpret_cast<uint16_t *>(dst + f * 18);
#pragma unroll
        for (int i = 0; i < 9; i++) o[i] = (uint16_t)fw[i];
        if (ta == sa && tb == sb && !seam_forced_open(force_open, ch, k)) return false;   // closed: the rest of the piece stands
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = nx[i];
#pragma unroll
        for (int i = 0; i < 9; i++) ow[i] = nw[i];
    }
    return true;
}

// The fix-up launch: every seam of the batch -- (channel, piece) pairs, `items` of them -- from a queue, a LANE at a time.
// A seam is a serial run of unknown length (at configs[2]: 200 frames on average, 2500 for the longest of 127 000; the
// lengths are close to exponentially distributed, tests/host/analysis/adx_seam_stats.c), so a wave that kept 64 seams
// until the last of them closed would run 1000 frames for 200 frames of work per lane.  Here a lane whose seam has closed
// takes the next one (the wave asks the queue when ADX_FIXUP_REFILL lanes are idle), and the launch lasts about as long as
// its longest seam run by a wave that has its SIMD to itself.
// The guessed run's crumbs stand in for a replay of its bytes (adx_encode_seam_run, which the tail kernel keeps): per frame
// one 8-byte load replaces nine 16-bit loads and the 32-sample decode, and the pre-scan is down to the two distances
// that see the history -- 650 instructions per frame instead of 1100.
// A lane's frame is loaded an iteration ahead (a lane that has just taken a seam sits its first iteration out).
constexpr int ADX_FIXUP_REFILL = 8;
template <bool V4, bool EXPONENTIAL>
__global__ __launch_bounds__(64) void adx_encode_fs18_fixup_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_length, int seg_frames, int segments, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, const int16_t *__restrict__ seg_state, const uint2 *__restrict__ crumbs,
    int *__restrict__ first_open, int *__restrict__ seam_open, int *__restrict__ seam_end, int force_open, int *__restrict__ queue,
    const int *__restrict__ own_frames, int *__restrict__ open_seams)
{
    const int lane = threadIdx.x;
    const int c0 = p.coef0, c1 = p.coef1;
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    const int64_t frames = ((int64_t)total_length + 31) / 32;
    const int64_t full_frames = total_length / 32;     // frames with all 32 samples (>= 64 here: pieces are that long at least)
    // own_frames (the ragged entry points' length buckets, capi_adx.hip): channel ch is a shorter stream zero-padded to
    // total_length and only its first own_frames[ch] frames are anybody's output.  What a seam does past them is nobody's
    // business -- and in digital silence the two runs need never meet (the run from the true history settles on a small
    // non-zero fixed point of the predictor's floors, the guessed run on zero): round 5's ragged call of 10 008 files spent
    // 100 ms per bucket chaining such seams through the padding of the bucket's shortest file.
    // the pieces that exist: seam k (1 .. pieces - 1) starts piece k
    const int pieces = (int)((frames + seg_frames - 1) / seg_frames) < segments ? (int)((frames + seg_frames - 1) / seg_frames) : segments;
    const int items = nch * (pieces - 1);
    bool active = false, have = false, drained = false;
    int ch = 0, k = 0, ta = 0, tb = 0;
    int64_t f = 0, fend = 0, own_end = 0;
    const int16_t *src = pcm;
    uint8_t *dst = out;
    uint4 cur[4], nxt[4];
    uint2 ccr = make_uint2(0, 0), ncr = make_uint2(0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) cur[i] = nxt[i] = make_uint4(0, 0, 0, 0);
    for (;;) {
        const uint64_t idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (!drained && (n_idle >= ADX_FIXUP_REFILL || n_idle == 64)) {
            int base = 0;
            if (lane == __ffsll((long long)idle) - 1) base = atomicAdd(queue, n_idle);
            base = __shfl(base, __ffsll((long long)idle) - 1);
            if (!active) {
                const int idx = base + (int)__popcll(idle & ((1ull << lane) - 1ull));
                if (idx < items) {                      // channel-fastest: neighbouring lanes start on neighbouring crumbs
                    k = 1 + idx / nch;
                    ch = idx - (k - 1) * nch;
                    f = (int64_t)k * seg_frames;
                    fend = f + seg_frames < frames ? f + seg_frames : frames;
                    own_end = own_frames ? (int64_t)own_frames[ch] : frames;
                    src = pcm + (int64_t)ch * pcm_pitch;
                    dst = out + (int64_t)ch * out_pitch;
                    ta = seg_state[((int64_t)(k - 1) * nch + ch) * 2];
                    tb = seg_state[((int64_t)(k - 1) * nch + ch) * 2 + 1];
                    active = f < own_end;               // a seam in the channel's padding: nothing to do
                    have = false;
                }
            }
            if (base + n_idle >= items) drained = true;
        }
        if (!__any(active)) {
            if (drained) return;
            continue;
        }
        if (active) {                                   // the frame after this one (a new seam: its first), clamped
            const int64_t fl = have ? f + 1 : f;
            const int64_t fc = fl < full_frames ? fl : full_frames - 1;
#pragma unroll
            for (int i = 0; i < 4; i++) nxt[i] = adx_load16(src + fc * 32 + 8 * i);
            ncr = crumbs[fc * nch + ch];
        }
        if (active && have) {
            uint32_t xw[16];
            if (f < full_frames) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    xw[4 * i] = cur[i].x; xw[4 * i + 1] = cur[i].y; xw[4 * i + 2] = cur[i].z; xw[4 * i + 3] = cur[i].w;
                }
            } else {                                    // the zero-padded last frame: its own loads
                adx_load_frame_slow(src, f, total_length, xw);
                ccr = crumbs[f * nch + ch];
            }
            uint32_t fw[9];
            adx_encode_frame_words<V4, EXPONENTIAL>(xw, ta, tb, c0, c1, filter_bits, (int)ccr.y, fw);
            uint16_t *o = reinterpret_cast<uint16_t *>(dst + f * 18);
#pragma unroll
            for (int i = 0; i < 9; i++) o[i] = (uint16_t)fw[i];
            const int sa = (int)(int16_t)(ccr.x & 0xFFFFu), sb = (int)ccr.x >> 16;       // the guessed run's history after this frame
            f++;
            if (ta == sa && tb == sb && !seam_forced_open(force_open, ch, k)) {
                active = false;                         // closed: the rest of the piece stands
            } else if (f >= own_end) {
                active = false;                         // the channel's own frames are all written: the rest is padding
            } else if (f >= fend) {
                // still open at the end of its piece: the chain launch carries on from the history reached here
                seam_open[(int64_t)(k - 1) * nch + ch] = 1;
                seam_end[(int64_t)(k - 1) * nch + ch] = (int)(((unsigned)tb << 16) | ((unsigned)ta & 0xFFFFu));
                atomicMin(&first_open[ch], k);
                // (seams the test hook holds open count only in its REPAIR mode, 3: the chained tail has tests of its own)
                if (!seam_forced_open(force_open, ch, k) || force_open == 3) atomicAdd(open_seams, 1);
                active = false;
            }
        }
        if (active) {
#pragma unroll
            for (int i = 0; i < 4; i++) cur[i] = nxt[i];
            ccr = ncr;
            have = true;
        }
    }
}

// The channels with an open seam, piece after piece (one lane per channel; lanes without one leave at once) -- the
// encoder's counterpart of the decoders' chained tail kernels and of gc_encode_chain_kernel: where the true history at
// the start of piece k is not seg_state[k - 1], which the fix-up launch assumed, the piece holds the run from
// seg_state[k - 1]; the same seam run from the true history finds where the two meet.  A run that does not meet by the
// end of the piece carries on; a later open seam of the channel starts again from its recorded end.  (Round 1 encoded
// the rest of the channel serially: 0.7 s for a 60 s channel.)
template <bool V4, bool EXPONENTIAL>
__global__ __launch_bounds__(64) void adx_encode_fs18_tail_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_length, int seg_frames, int segments, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, const int16_t *__restrict__ seg_state, const int *__restrict__ first_open,
    const int *__restrict__ seam_open, const int *__restrict__ seam_end, int force_open, const int *__restrict__ own_frames,
    const int *__restrict__ open_seams, int many)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    if (open_seams[0] >= many) return;                 // many seams that would not close: the REPAIR launch takes them all
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    const int64_t own_end = own_frames ? (int64_t)own_frames[ch] : ((int64_t)total_length + 31) / 32;   // (see the fix-up kernel)
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = out + (int64_t)ch * out_pitch;
    const int c0 = p.coef0, c1 = p.coef1;
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    bool carry = false;
    int ta = 0, tb = 0;
    for (int k = k0; k < segments; k++) {
        const int64_t f0 = (int64_t)k * seg_frames;
        if (f0 * 32 >= total_length || f0 >= own_end) break;
        const int64_t idx = (int64_t)(k - 1) * nch + ch;
        bool apart = false;
        if (carry)
            apart = adx_encode_seam_run<V4, EXPONENTIAL>(src, dst, f0, seg_frames, total_length, c0, c1, filter_bits, ta, tb,
                                                         seg_state[idx * 2], seg_state[idx * 2 + 1], ch, k, force_open);
        if (apart) {
            carry = true;                              // (ta, tb): the true history at the end of this piece
        } else if (seam_open[idx] != 0) {
            carry = true;                              // this piece's own seam ran out of frames: its recorded end is the truth
            const int e = seam_end[idx];
            ta = (int)(int16_t)(e & 0xFFFF);
            tb = e >> 16;
        } else
            carry = false;
    }
}


// The one-wave encoder's pieces: two waves on every SIMD, each piece at least this long (a seam takes 200 frames to close on
// average and 2500 for the longest of configs[2]'s 127 000, tests/host/analysis/adx_seam_stats.c; one still open at the end
// of its piece carries on in the tail kernel).  The fix-up's lanes take seams from a queue: one persistent wave per SIMD
// (profiles/r04_e_adx_parts.log: 512 / 1024 / 2048 waves 5.8 / 5.4 / 6.4 ms).
constexpr int ADX_DIRECT_WAVES_PER_SIMD = 2;
constexpr int ADX_DIRECT_MIN_PIECE_FRAMES = 2560;
constexpr int ADX_FIXUP_WAVES_PER_SIMD = 1;

int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length, const AdxDeviceParams &p,
                  uint8_t *d_out, int64_t out_pitch, int16_t *d_history_out, hipStream_t stream, const int *d_own_frames)
{
    if (nch <= 0) return VGA_OK;
    const dim3 grid((nch + 63) / 64), block(64);
    // pcm rows must be 16-byte aligned for the vector loads, output rows 4-byte aligned for the dword stores
    // (and coefficients of the size the reference can produce: adx_quantise_step's 32-bit bound)
    // (padding: at most two frames of it -- CriAdxFormat.cs:59-62 never asks for more than 63 samples; the kernels see the
    // stream, i.e. the rows moved back by the padding and a length that counts it)
    const bool fast = p.frame_size == 18 && p.padding >= 0 && p.padding <= 64 && (pcm_pitch % 8) == 0 && ((uintptr_t)d_pcm % 16) == 0 &&
                      (out_pitch % 4) == 0 && ((uintptr_t)d_out % 4) == 0 && std::abs((int)p.coef0) <= 16384 &&
                      std::abs((int)p.coef1) <= 16384;
    if (fast) {
        d_pcm -= p.padding;                            // from here on d_pcm / pcm_length are the STREAM's
        pcm_length += p.padding;
        const bool v4 = p.version == 4, ex = p.type == 4;
        // as many time segments as put ADX_DIRECT_WAVES_PER_SIMD waves on every SIMD, each an even number of frames and at
        // least ADX_DIRECT_MIN_PIECE_FRAMES long
        const int groups64 = (nch + 63) / 64;
        const int cus = device_cu_count();
        const int frames = (pcm_length + 31) / 32;
        int segments = cus * 4 * ADX_DIRECT_WAVES_PER_SIMD / groups64;
        if (segments > frames / ADX_DIRECT_MIN_PIECE_FRAMES) segments = frames / ADX_DIRECT_MIN_PIECE_FRAMES;
        if (segments < 1) segments = 1;
        if (segments > 64) segments = 64;
        if (encoder_segments_override() > 0) segments = std::min(std::max(frames / 64, 1), encoder_segments_override());   // test hook
        int seg_frames = (frames + segments - 1) / segments;
        seg_frames += seg_frames & 1;
        AsyncBuf scratch;                              // freed (stream-ordered) on every exit path
        int16_t *seg_state = nullptr;                  // [segments][nch][2] final histories, then [nch] first open seam
        int *first_open = nullptr, *seam_open = nullptr, *seam_end = nullptr, *queue = nullptr;
        uint2 *crumbs = nullptr;                       // [frames][nch]: 8 bytes per frame and channel (2.9 GB at configs[2])
        if (segments > 1) {
            const size_t state_bytes = round_up((size_t)segments * nch * 2 * sizeof(int16_t), 16);
            const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
            const size_t small_bytes = round_up(state_bytes + (size_t)nch * sizeof(int) + 2 * flag_bytes + 16, 16);   // (+ the fix-up's queue, the open seams' count)
            VGA_HIP_TRY(scratch.alloc(small_bytes + (size_t)frames * nch * sizeof(uint2), stream));
            seg_state = scratch.as<int16_t>();
            first_open = reinterpret_cast<int *>(scratch.as<unsigned char>() + state_bytes);
            seam_open = first_open + nch;
            seam_end = seam_open + (size_t)(segments - 1) * nch;
            queue = seam_end + (size_t)(segments - 1) * nch;
            crumbs = reinterpret_cast<uint2 *>(scratch.as<unsigned char>() + small_bytes);
            VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
            VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes, stream));
            VGA_HIP_TRY(hipMemsetAsync(queue, 0, 2 * sizeof(int), stream));
        }
        int *open_seams = queue ? queue + 1 : nullptr; // seams still open at the end of their pieces
        // "many": one seam in 64, and at least 8 -- as the decoders' threshold (gc_decode_kernel.hip); test hook mode 3 = any
        const int many = force_open_seams() == 3 ? 1
                         : (int)std::min<int64_t>(0x7fffffff, std::max<int64_t>(8, (int64_t)nch * (segments - 1) / 64));
        // the fix-up's persistent waves: one per SIMD, fewer when there are not that many seams
        int fixup_waves = cus * 4 * ADX_FIXUP_WAVES_PER_SIMD;
#ifdef VGA_TUNING   // tools/build_variants.sh only
        if (const char *e = std::getenv("VGA_HIP_ADX_FIXUP_WAVES")) fixup_waves = std::max(1, std::atoi(e));
#endif
        fixup_waves = (int)std::min<int64_t>(fixup_waves, ((int64_t)nch * (segments - 1) + 63) / 64);
        if (fixup_waves < 1) fixup_waves = 1;
#define VGA_ADX_ENC_T(V, E)                                                                                              \
        {                                                                                                                \
            hipLaunchKernelGGL((adx_encode_fs18_direct_kernel<V, E>), dim3(groups64, segments), dim3(64), 0, stream, d_pcm, \
                               pcm_pitch, nch, pcm_length, seg_frames, p, d_out, out_pitch, d_history_out, seg_state, crumbs, \
                               (const int *)nullptr, (const int *)nullptr, 0);                                          \
            VGA_HIP_TRY(hipGetLastError());                                                                              \
            if (segments > 1) {                                                                                          \
                hipLaunchKernelGGL((adx_encode_fs18_fixup_kernel<V, E>), dim3(fixup_waves), dim3(64), 0, stream,         \
                                   d_pcm, pcm_pitch, nch, pcm_length, seg_frames, segments, p, d_out, out_pitch, seg_state, crumbs, \
                                   first_open, seam_open, seam_end, force_open_seams(), queue, d_own_frames, open_seams); \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL((adx_encode_fs18_tail_kernel<V, E>), dim3(groups64), dim3(64), 0, stream, d_pcm,     \
                                   pcm_pitch, nch, pcm_length, seg_frames, segments, p, d_out, out_pitch, seg_state,    \
                                   first_open, seam_open, seam_end, force_open_seams(), d_own_frames, open_seams, many); \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL((adx_encode_fs18_direct_kernel<V, E, true>), dim3(groups64, 1), dim3(64), 0, stream, d_pcm, \
                                   pcm_pitch, nch, pcm_length, seg_frames, p, d_out, out_pitch, (int16_t *)nullptr, seg_state, \
                                   (uint2 *)nullptr, (const int *)first_open, (const int *)open_seams, many);           \
            }                                                                                                            \
        }
        if (v4 && ex) VGA_ADX_ENC_T(true, true)
        else if (v4) VGA_ADX_ENC_T(true, false)
        else if (ex) VGA_ADX_ENC_T(false, true)
        else VGA_ADX_ENC_T(false, false)
#undef VGA_ADX_ENC_T
    } else {                                           // other frame sizes, padded (looping) streams, odd alignments
        hipLaunchKernelGGL(adx_encode_kernel, grid, block, 0, stream, d_pcm, pcm_pitch, nch, pcm_length, p, d_out, out_pitch,
                           d_history_out);
    }
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_decode(const uint8_t *d_adpcm, int64_t in_pitch, int nch, int sample_count, const AdxDeviceParams &p,
                  int16_t *d_pcm, int64_t pcm_pitch, int *d_status, hipStream_t stream, const int *d_own_samples)
{
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    // (padded streams: up to two frames of padding, equal-length batches, at least two frames of output)
    const bool fast = p.frame_size == 18 && (p.padding == 0 || (p.padding > 0 && p.padding <= 64 && !d_own_samples && sample_count >= 64)) &&
                      (pcm_pitch % 8) == 0 && ((uintptr_t)d_pcm % 16) == 0 && (in_pitch % 4) == 0 && ((uintptr_t)d_adpcm % 4) == 0;
    if (fast) {
        if (p.padding > 0) {
            // The reference reads ceil(sampleCount / 32) frames from the frame the padding ends in and takes 32 - padding % 32
            // samples from the first of them (CriAdxCodec.cs:18-34): when that is not enough for sampleCount, the last samples
            // stay zero.  In stream positions: samples [padding, padding + decoded) are decoded, `pcm` moves back by the padding.
            const int out_count = sample_count;
            const int decoded = std::min(out_count, (out_count + 31) / 32 * 32 - p.padding % 32);
            if (decoded < out_count)
                VGA_HIP_TRY(hipMemset2DAsync(d_pcm + decoded, (size_t)pcm_pitch * sizeof(int16_t), 0,
                                             (size_t)(out_count - decoded) * sizeof(int16_t), (size_t)nch, stream));
            d_pcm -= p.padding;
            sample_count = decoded + p.padding;
        }
        // as many time pieces as put ONE wave on every SIMD (a wave = 64 channels of one piece), each at least 512 frames long
        // and an even number of frames.  The kernel is bound by its stores, and what the memory system holds open is one
        // row position per (channel, piece): at configs[2] 8 / 16 / 32 / 64 pieces take 8.5 / 8.0 / 13.1 / 12.1 ms (and 12 or
        // 24, which leave some SIMDs with two waves and some with one, 11 ms)
        const int groups = (nch + 63) / 64;
        const int cus = device_cu_count();
        const int frames = (sample_count + 31) / 32;
        int segments = cus * 4 / groups;
        if (segments > frames / 512) segments = frames / 512;
        if (segments < 1) segments = 1;
        // every piece boundary is a seam that may still be open at the end of its piece (an integer IIR can keep two runs
        // one LSB apart for good; the tail kernel then decodes the next piece again from the true history, piece after
        // piece while the runs stay apart): at most 64 pieces
        if (segments > 64) segments = 64;
        if (encoder_segments_override() > 0) segments = std::min(std::max(frames / 8, 1), encoder_segments_override());   // test hook
        int seg_frames = (frames + segments - 1) / segments;
        seg_frames += seg_frames & 1;
        AsyncBuf scratch;                              // freed (stream-ordered) on every exit path
        int *first_open = nullptr, *seam_open = nullptr, *slow_seams = nullptr;
        if (segments > 1) {
            const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
            VGA_HIP_TRY(scratch.alloc((size_t)nch * sizeof(int) + flag_bytes + 16, stream));
            first_open = scratch.as<int>();
            seam_open = first_open + nch;
            slow_seams = seam_open + (size_t)(segments - 1) * nch;         // [0] seams that stayed open, [1] how many make "many"
            VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
            VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes + 16, stream));
            const int many = (int)std::min<int64_t>(0x7fffffff, std::max<int64_t>(8, (int64_t)nch * (segments - 1) / 64));   // (gc_decode_kernel.hip)
            // (a fill, not a copy from this stack frame: a pageable host-to-device copy makes the call wait for the stream)
            VGA_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slow_seams + 1), many, 1, stream));
        }
#define VGA_ADX_DEC_T(V)                                                                                                 \
        {                                                                                                                \
            hipLaunchKernelGGL((adx_decode_fs18_direct_kernel<V, false>), dim3(groups, segments), dim3(64), 0, stream,   \
                               d_adpcm, in_pitch, nch, sample_count, seg_frames, p, d_pcm, pcm_pitch, d_status,          \
                               (const int *)nullptr, (const int *)nullptr);                                              \
            VGA_HIP_TRY(hipGetLastError());                                                                              \
            if (segments > 1) {                                                                                          \
                hipLaunchKernelGGL(adx_decode_fs18_fixup_kernel<V>, dim3(groups, segments - 1), dim3(64), 0, stream,    \
                                   d_adpcm, in_pitch, nch, sample_count, seg_frames, p, d_pcm, pcm_pitch, first_open,   \
                                   seam_open, force_open_seams(), slow_seams, d_own_samples);                           \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL(adx_decode_fs18_tail_kernel<V>, dim3(groups), dim3(64), 0, stream, d_adpcm, in_pitch, \
                                   nch, sample_count, seg_frames, segments, p, d_pcm, pcm_pitch, first_open, seam_open,  \
                                   force_open_seams(), slow_seams, d_own_samples);                                      \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL((adx_decode_fs18_direct_kernel<V, true>), dim3(groups, 1), dim3(64), 0, stream,       \
                                   d_adpcm, in_pitch, nch, sample_count, seg_frames, p, d_pcm, pcm_pitch, d_status,      \
                                   (const int *)first_open, (const int *)slow_seams);                                   \
            }                                                                                                            \
        }
        if (p.version == 4) VGA_ADX_DEC_T(true)
        else VGA_ADX_DEC_T(false)
#undef VGA_ADX_DEC_T
    } else {
        hipLaunchKernelGGL(adx_decode_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_adpcm, in_pitch, nch, sample_count, p,
                           d_pcm, pcm_pitch, d_status);
    }
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace adx
}  // namespace vga
