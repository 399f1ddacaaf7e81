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

// ---- FrameSize 18 (32 samples), no padding -- BASELINE config 3 and the reference's defaults: one frame's
// 32 samples as 16 dwords (the tiled encoder's loader)
__device__ __forceinline__ void adx_load32(const int16_t *src, int64_t first, int pcm_length, uint32_t (&w)[16])
{
    if (first + 32 <= pcm_length) {
        const uint4 *p = reinterpret_cast<const uint4 *>(src + first);       // 64-byte aligned: first % 32 == 0
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 v = p[k];
            w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t i0 = first + 2 * k, i1 = i0 + 1;
            const uint32_t lo = i0 < pcm_length ? (uint16_t)src[i0] : 0u;
            const uint32_t hi = i1 < pcm_length ? (uint16_t)src[i1] : 0u;
            w[k] = lo | (hi << 16);
        }
    }
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
template <bool V4>
__global__ __launch_bounds__(64) void adx_decode_fs18_direct_kernel(
    const uint8_t *__restrict__ adpcm, int64_t in_pitch, int nch, int total_samples, int seg_frames, AdxDeviceParams p,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ status)
{
    const int64_t first_frame = (int64_t)blockIdx.y * seg_frames;          // even (seg_frames is)
    if (first_frame * 32 >= total_samples) return;
    const int sample_count = (int)((int64_t)total_samples - first_frame * 32 < (int64_t)seg_frames * 32
                                       ? (int64_t)total_samples - first_frame * 32 : (int64_t)seg_frames * 32);
    const int frame_count = (sample_count + 31) / 32;
    const int ch_raw = blockIdx.x * 64 + threadIdx.x;
    const bool live = ch_raw < nch;
    const int ch = live ? ch_raw : nch - 1;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(adpcm + (int64_t)ch * in_pitch + first_frame * 18);
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch + first_frame * 32;
    int hist1 = blockIdx.y > 0 ? 0 : p.history, hist2 = hist1;             // later pieces: the guess (0, 0)
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
    auto decode_frame = [&](const uint32_t (&w)[5], auto mode_tag, int frame, int valid) {
        constexpr int MODE = decltype(mode_tag)::value;          // 0: whole frame, stored turned; 1: whole frame, stored by its lane; 2: partial; 3: not stored (warm-up)
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
            hist2 = hist1;                             // a partial last frame runs on: nothing reads the history after it
            hist1 = fin;
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
                    *reinterpret_cast<int4 *>(turned_row(i) + (int64_t)(frame - (TURN - 1)) * 32) =       // 5 ms slower
                        s_turn[(lane / LPR + RPI * i) * (LPR + 1) + lane % LPR];
                asm volatile("" ::: "memory");
            }
        } else if (MODE == 1) {
            if (live) {
                int16_t *d = dst + (int64_t)frame * 32;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    reinterpret_cast<int4 *>(d)[q] =
                        make_int4((o[8 * q] & 0xFFFF) | (o[8 * q + 1] << 16), (o[8 * q + 2] & 0xFFFF) | (o[8 * q + 3] << 16),
                                  (o[8 * q + 4] & 0xFFFF) | (o[8 * q + 5] << 16), (o[8 * q + 6] & 0xFFFF) | (o[8 * q + 7] << 16));
            }
        } else if (MODE == 2 && live) {
            int16_t *d = dst + (int64_t)frame * 32;
#pragma unroll
            for (int s2 = 0; s2 < 32; s2++)
                if (s2 < valid) d[s2] = (int16_t)o[s2];
        }
    };
    // ---- warm-up of a later piece: the WARM frames before it are decoded from the guess (0, 0) and not stored, so that the
    // piece itself starts from a history that has, as a rule, already fallen into step with the true run (the decoder
    // forgets a wrong history within 2000 samples on audio): its seam then closes on the first frame the fix-up launch
    // checks (15 seams per channel at configs[2]: 3.7 ms of fix-up without this, against 7.9 ms for the decode itself)
    if (blockIdx.y > 0) {
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
    {
        const uint32_t *f = src;                                           // pair 0 (or, with no pair at all, five dwords of
        const int nq = full_pairs > 0 ? 9 : 5;                             // the row's first frame: unused)
#pragma unroll
        for (int q = 0; q < 9; q++) cur[q] = q < nq ? f[q] : 0u;
        // the first pair is waited for HERE: left to the loop header, the wait would also sit on the back edge, where it
        // means "every store of the pair before has completed"
#pragma unroll
        for (int q = 0; q < 9; q++) asm volatile("" : "+v"(cur[q]));
    }
#pragma unroll 1
    for (int k = 0; k < full_pairs; k++) {
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
    for (int i = 2 * full_pairs; i < frame_count; i++) {
        // frame i starts at byte 18 i: on a dword for even i, two bytes after one for odd i
        const uint32_t *f = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint16_t *>(src) + (int64_t)i * 9 - (i & 1));
        uint32_t t[5], w[5];
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
        const int valid = min(32, sample_count - i * 32);
        if (valid == 32) decode_frame(w, std::integral_constant<int, 1>{}, i, 32);
        else decode_frame(w, std::integral_constant<int, 2>{}, i, valid);
    }
    if (bad && live && status) atomicOr(status, 1);
}

// One frame of CriAdxCodec.Decode (:23-45) from the history (hist1, hist2) into o[0 .. valid).
template <bool V4>
__device__ __forceinline__ void adx_decode_frame_serial(const uint8_t *fr, const AdxDeviceParams &p, int valid, int &hist1,
                                                        int &hist2, int16_t *o)
{
    const int hb0 = fr[0], hb1 = fr[1];
    int filter_num = ((hb0 >> 4) & 0xF) >> 1;
    int cf0, cf1;
    if (p.type == 2) {                                  // as the tiled kernel's `prepare`
        if (filter_num > 3) filter_num = 3;
        cf0 = filter_num == 0 ? 0 : (filter_num == 1 ? 0x0F00 : (filter_num == 2 ? 0x1CC0 : 0x1880));
        cf1 = filter_num == 0 ? 0 : (filter_num == 1 ? 0 : (filter_num == 2 ? (int)(int16_t)0xF300 : (int)(int16_t)0xF240));
    } else {
        cf0 = p.coef0;
        cf1 = p.coef1;
    }
    int scale = (int)(int16_t)(((hb0 << 8) | hb1) & 0x1FFF);
    scale = (int)(int16_t)(p.type == 4 ? (1 << ((12 - scale) & 31)) : scale + 1);
    for (int s = 0; s < valid; s++) {
        const int byte = fr[2 + (s >> 1)];
        int sample = (s & 1) ? (byte & 0xF) : (byte >> 4);
        sample = (sample ^ 8) - 8;
        if (V4) sample = scale * sample + ((hist1 * cf0 + hist2 * cf1) >> 12);
        else sample = scale * sample + ((hist1 * cf0) >> 12) + ((hist2 * cf1) >> 12);
        const int fin = clamp16(sample);
        hist2 = hist1;
        hist1 = fin;
        o[s] = (int16_t)fin;
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
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ first_open, int *__restrict__ seam_open, int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const int k = blockIdx.y + 1;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (ch >= nch || f0 * 32 >= total_samples) return;
    const uint8_t *src = adpcm + (int64_t)ch * in_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
    // Seed read concurrently with seam k-1's lane rewriting piece k-1 -- same invariant as gc_decode_fixup_kernel
    // (gc_decode_kernel.hip): a seam that closes leaves the piece's last samples with the values they already hold,
    // one that stays open hands pieces k.. to the tail kernel, which redoes them from the final samples.
    int hist1 = dst[f0 * 32 - 1], hist2 = dst[f0 * 32 - 2];
    for (int64_t f = f0; f < f0 + seg_frames && f * 32 < total_samples; f++) {
        const int valid = (int)((int64_t)total_samples - f * 32 < 32 ? (int64_t)total_samples - f * 32 : 32);
        int16_t *o = dst + f * 32;
        int g1 = 0, g2 = 0;                             // the guessed run's history at this frame's end
        if (valid == 32) { g1 = o[31]; g2 = o[30]; }
        adx_decode_frame_serial<V4>(src + f * 18, p, valid, hist1, hist2, o);
        if (valid == 32 && hist1 == g1 && hist2 == g2 && !seam_forced_open(force_open, ch, k)) return;
        if (valid < 32) return;                         // the stream's last, partial frame: nothing follows
    }
    if (f0 + seg_frames < ((int64_t)total_samples + 31) / 32) {   // open, and a piece follows
        seam_open[(int64_t)(k - 1) * nch + ch] = 1;
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
    int16_t *__restrict__ pcm, int64_t pcm_pitch, const int *__restrict__ first_open, const int *__restrict__ seam_open,
    int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    const uint8_t *src = adpcm + (int64_t)ch * in_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
    bool carry = false;
    int hist1 = 0, hist2 = 0;
    for (int k = k0; k < segments; k++) {
        const int64_t f0 = (int64_t)k * seg_frames;
        if (f0 * 32 >= total_samples) break;
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
                if (valid == 32 && hist1 == g1 && hist2 == g2 && !seam_forced_open(force_open, ch, k)) { apart = false; break; }
            }
        }
        const int64_t f1 = f0 + seg_frames;
        if (apart) {
            carry = true;
        } else if (flagged && f1 * 32 < total_samples) {
            carry = true;
            hist1 = dst[f1 * 32 - 1];
            hist2 = dst[f1 * 32 - 2];
        } else
            carry = false;
    }
}

// ---------------------------------------------------------------- 18-byte frames, encoder wave + helper waves
// CriAdxCodec.EncodeFrame (:107-147) with the same division of labour as the decoders: three helper waves load
// and unpack the 32 samples of every frame one tile ahead, pre-scan the 30 distances that involve input
// samples only (:112-118; the first two use the reconstructed history) and pack / store the previous tile's
// nibbles; the encoder wave (lane = channel) keeps the scale computation and the quantise recurrence
// (:120-138).  scale_short_to_nibble (:167-171) is done on the magnitude: |q| = ((|v| + 2340) * 114692) >> 29 is
// floor((|v| + 2340) / 4681) for |v| <= 32768 (2^29 / 4681 = 114691.5.., error term 2340 per unit: exact below
// 229 432), the result is at most 7 so Clamp4 and the Clamp16 of scale * q (scale <= 4096) cannot bind.
constexpr int EATF = 2;                                // frames per tile (1: 35.5 instead of 27.0 ms at configs[2])
constexpr int ECW = 128;                               // channels per workgroup: TWO encoder waves (on different SIMDs of the
                                                       // CU) and six helper waves; the double-buffered tiles fill one CU's LDS
constexpr int ETHREADS = ECW * 4, EHELPERS = ETHREADS - ECW;
struct AdxEncodeTile {
    int4 x[EATF][8][ECW];                                // [frame][eighth][channel]: the 32 input samples
    int pmax[EATF][ECW];                                 // max |Clamp16(distance)| over samples 2..31
    int4 q[EATF][8][ECW];                                // encoder output: 32 nibbles (-7..7)
    int hdr[EATF][ECW];                                  // encoder output: the 16 header bits (scale, filter)
};

template <bool V4, bool EXPONENTIAL>
// Time segments (blockIdx.y): every piece of `seg_frames` frames (an even number) but the first is encoded from a
// guessed history -- the two INPUT samples before it -- and adx_encode_fs18_fixup_kernel closes the seams afterwards.
// seg_state[segment][channel] receives each piece's final history (two int16).
__global__ __launch_bounds__(ETHREADS) void adx_encode_fs18_tiled_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_length, int seg_frames, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, int16_t *__restrict__ history_out, int16_t *__restrict__ seg_state)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    AdxEncodeTile *s_tile = reinterpret_cast<AdxEncodeTile *>(s_raw);          // [2]
    const int tid = threadIdx.x;
    const int ch0 = blockIdx.x * ECW;
    const int64_t first_frame = (int64_t)blockIdx.y * seg_frames;
    if (first_frame > 0 && first_frame * 32 >= total_length) return;
    const int pcm_length = (int)((int64_t)total_length - first_frame * 32 < (int64_t)seg_frames * 32
                                     ? (int64_t)total_length - first_frame * 32 : (int64_t)seg_frames * 32);
    pcm += first_frame * 32;
    out += first_frame * 18;
    const int frame_count = (pcm_length + 31) / 32;
    const int tiles = (frame_count + EATF - 1) / EATF;
    const int c0 = p.coef0, c1 = p.coef1;

    if (tid >= ECW) {
        // ------------------------------------------------------------ helper waves (192 lanes)
        const int hl = tid - ECW;
        constexpr int ITEMS = (ECW * EATF + EHELPERS - 1) / EHELPERS;
        struct Raw { uint4 v[4]; };
        auto load_tile = [&](int tile, Raw (&raw)[ITEMS]) {          // unconditional loads, clamped frame index
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const int item = min(hl + EHELPERS * k, ECW * EATF - 1);
                const int c = item / EATF, j = item - c * EATF;
                // the last frame may be partial: clamp to the last FULL frame (re-read below if needed)
                const int i = min(tile * EATF + j, max(pcm_length / 32 - 1, 0));
                const int ch = min(ch0 + c, nch - 1);
                const uint4 *src = reinterpret_cast<const uint4 *>(pcm + (int64_t)ch * pcm_pitch + (int64_t)i * 32);
                if (pcm_length < 32) {                  // no full frame at all: nothing to prefetch (uniform)
#pragma unroll
                    for (int q = 0; q < 4; q++) raw[k].v[q] = make_uint4(0, 0, 0, 0);
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 4; q++) raw[k].v[q] = src[q];
            }
        };
        auto prepare = [&](int tile, const Raw (&raw)[ITEMS]) {
            AdxEncodeTile &T = s_tile[tile & 1];
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const int item = hl + EHELPERS * k;
                if (item >= ECW * EATF) continue;
                const int c = item / EATF, j = item - c * EATF;
                const int i = tile * EATF + j;
                if (i >= frame_count) continue;
                uint32_t w[16];
                if ((int64_t)i * 32 + 32 <= pcm_length) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        w[4 * q] = raw[k].v[q].x; w[4 * q + 1] = raw[k].v[q].y; w[4 * q + 2] = raw[k].v[q].z; w[4 * q + 3] = raw[k].v[q].w;
                    }
                } else {                               // zero-padded last frame
                    const int ch = min(ch0 + c, nch - 1);
                    adx_load32(pcm + (int64_t)ch * pcm_pitch, (int64_t)i * 32, pcm_length, w);
                }
                int x[32];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    x[2 * q] = (int)(int16_t)(w[q] & 0xFFFF);
                    x[2 * q + 1] = (int)w[q] >> 16;
                }
                int pm = 0;
#pragma unroll
                for (int s = 2; s < 32; s++) {         // history = the two input samples before (:112-118)
                    const int predicted = ((x[s - 1] * c0) >> 12) + ((x[s - 2] * c1) >> 12);
                    int distance = clamp16(x[s] - predicted);
                    distance = distance < 0 ? -distance : distance;
                    pm = max(pm, distance);
                }
                T.pmax[j][c] = pm;
#pragma unroll
                for (int q = 0; q < 8; q++) T.x[j][q][c] = make_int4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
            }
        };
        auto flush = [&](int tile) {
            const AdxEncodeTile &T = s_tile[tile & 1];
            for (int item = hl; item < ECW * EATF; item += EHELPERS) {
                const int c = item / EATF, j = item - c * EATF;
                const int i = tile * EATF + j;
                if (i >= frame_count || ch0 + c >= nch) continue;
                uint32_t bytes[5] = {(uint32_t)T.hdr[j][c], 0, 0, 0, 0};       // 18 bytes + 2 spare
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int4 v = T.q[j][q][c];
                    const int qq[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int s = 4 * q + e;
                        const int byte_index = 2 + (s >> 1);
                        bytes[byte_index >> 2] |= (uint32_t)(qq[e] & 0xF) << (8 * (byte_index & 3) + ((s & 1) ? 0 : 4));
                    }
                }
                uint16_t *f = reinterpret_cast<uint16_t *>(out + (int64_t)(ch0 + c) * out_pitch) + (int64_t)i * 9;
                if ((i & 1) == 0) {                    // 36*k bytes: dword aligned
                    uint32_t *f32 = reinterpret_cast<uint32_t *>(f);
#pragma unroll
                    for (int q = 0; q < 4; q++) f32[q] = bytes[q];
                    f[8] = (uint16_t)bytes[4];
                } else {                               // 2 bytes past a dword boundary
                    f[0] = (uint16_t)bytes[0];
                    uint32_t *f32 = reinterpret_cast<uint32_t *>(f + 1);
#pragma unroll
                    for (int q = 0; q < 4; q++) f32[q] = (bytes[q] >> 16) | (bytes[q + 1] << 16);
                }
            }
        };
        Raw ra[ITEMS], rb[ITEMS];
        load_tile(0, ra);
        load_tile(1, rb);
        if (tiles > 0) prepare(0, ra);
        lds_barrier();
        for (int tile = 0; tile < tiles; tile += 2) {
            load_tile(tile + 2, ra);
            if (tile + 1 < tiles) prepare(tile + 1, rb);
            if (tile > 0) flush(tile - 1);
            lds_barrier();
            if (tile + 1 < tiles) {
                load_tile(tile + 3, rb);
                if (tile + 2 < tiles) prepare(tile + 2, ra);
                flush(tile);
                lds_barrier();
            }
        }
        if (tiles > 0) flush(tiles - 1);
        return;
    }

    // ---------------------------------------------------------------- encoder wave: lane = channel
    __builtin_amdgcn_s_setprio(3);
    const int ch = min(ch0 + tid, nch - 1);
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    const double raw_bound = 32770.0 + 8.0 * (double)((c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1));
    int h0 = 0, h1 = 0, hist = p.history;             // h1 = the newer sample
    if (blockIdx.y > 0) {                              // the guess: the input just before this piece
        h0 = pcm[(int64_t)ch * pcm_pitch - 2];
        h1 = pcm[(int64_t)ch * pcm_pitch - 1];
    } else {
        if (V4 && pcm_length > 0) { h0 = h1 = pcm[(int64_t)ch * pcm_pitch]; hist = h0; }      // :69-74
        if (history_out && ch0 + tid < nch) history_out[ch] = (int16_t)hist;
    }

    lds_barrier();                                     // tile 0 prepared
    for (int tile = 0; tile < tiles; tile++) {
        AdxEncodeTile &T = s_tile[tile & 1];
        const int nf = min(EATF, frame_count - tile * EATF);
#pragma unroll 1
        for (int j = 0; j < nf; j++) {
            int x[32];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int4 v = T.x[j][q][tid];
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
            // pre-scan (:112-118): the helpers' maximum + the two distances that see the reconstructed history
            int max_distance = T.pmax[j][tid];
            {
                int d0 = clamp16(x[0] - (((h1 * c0) >> 12) + ((h0 * c1) >> 12)));
                int d1 = clamp16(x[1] - (((x[0] * c0) >> 12) + ((h1 * c1) >> 12)));
                d0 = d0 < 0 ? -d0 : d0;
                d1 = d1 < 0 ? -d1 : d1;
                max_distance = max(max_distance, max(d0, d1));
            }
            double gain;
            int scale_out;
            const int scale = calculate_scale(max_distance, EXPONENTIAL, gain, scale_out);
            T.hdr[j][tid] = (((scale_out >> 8) & 0x1f) | filter_bits) | ((scale_out & 0xff) << 8);     // :140-141, :95
            int a = h0, b = h1, qv[32];
            auto quantise = [&](auto guard_c) __attribute__((always_inline)) {
                constexpr bool GUARD = decltype(guard_c)::value;
#pragma unroll
                for (int s = 0; s < 32; s++) {         // :122-138
                    const int pb = (__mul24(a, c1)) >> 12;           // older sample: ready a step early
                    const int pa = (__mul24(b, c0)) >> 12;
                    const int raw = (x[s] - pb) - pa;
                    const double prod = (double)raw * gain;
                    const int scaled = clamp16(GUARD ? trunc_i32_ryujit(prod) : (int)prod);
                    const int sm = scaled >> 31;                      // scale_short_to_nibble on the magnitude
                    const unsigned mag = (unsigned)((scaled ^ sm) - sm);
                    const int aq = (int)((mag * 114692u + 2340u * 114692u) >> 29);
                    const int q = (aq ^ sm) - sm;
                    const int predicted = V4 ? (__mul24(b, c0) + __mul24(a, c1)) >> 12 : pa + pb;
                    const int rec = clamp16(__mul24(scale, q) + predicted);
                    a = b;
                    b = rec;
                    qv[s] = q;
                }
            };
            // |rawDistance| <= 32768 + 8 (|c0| + |c1|): only a frame whose gain can push that past 2^31 needs the
            // RyuJIT overflow semantics of the cast (a wave-uniform, practically never taken branch)
            if (__any(gain * raw_bound >= 2147483648.0)) quantise(std::true_type{});
            else quantise(std::false_type{});
            h0 = a;
            h1 = b;
#pragma unroll
            for (int q = 0; q < 8; q++) T.q[j][q][tid] = make_int4(qv[4 * q], qv[4 * q + 1], qv[4 * q + 2], qv[4 * q + 3]);
        }
        lds_barrier();
    }
    if (seg_state && ch0 + tid < nch) {
        int16_t *st = seg_state + ((int64_t)blockIdx.y * nch + ch) * 2;
        st[0] = (int16_t)h0;
        st[1] = (int16_t)h1;
    }
}

// One frame of CriAdxCodec.EncodeFrame (:107-147) from the history (a, b); maths as adx_encode_kernel.
// The frame leaves as nine 16-bit words (low byte = the earlier byte of the frame).
template <bool V4, bool EXPONENTIAL>
__device__ __forceinline__ void adx_encode_frame_words(const int (&x)[32], int &a, int &b, int c0, int c1, int filter_bits,
                                                       uint32_t (&fw)[9])
{
    int max_distance = 0;
    {
        int pa = a, pb = b;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int predicted = (__mul24(pb, c0) >> 12) + (__mul24(pa, c1) >> 12);     // 16-bit x 16-bit: exact in 24 bits
            int distance = clamp16(x[j] - predicted);
            distance = distance < 0 ? -distance : distance;
            max_distance = max(max_distance, distance);
            pa = pb;
            pb = x[j];
        }
    }
    double gain;
    int scale_out;
    const int scale = calculate_scale(max_distance, EXPONENTIAL, gain, scale_out);
    fw[0] = (uint32_t)((((scale_out >> 8) & 0x1f) | filter_bits) & 0xff) | ((uint32_t)(scale_out & 0xff) << 8);
    // the quantise recurrence (:122-138) as the tiled kernel has it: 24-bit multiplies, ScaleShortToNibble on the magnitude
    // (see there), and the RyuJIT overflow semantics of the cast only for a frame whose gain can push rawDistance past 2^31
    // (a wave-uniform, practically never taken branch)
    int qv[32];
    auto quantise = [&](auto guard_c) __attribute__((always_inline)) {
        constexpr bool GUARD = decltype(guard_c)::value;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const int pb = __mul24(a, c1) >> 12;
            const int pa = __mul24(b, c0) >> 12;
            const int raw = (x[j] - pb) - pa;
            const double prod = (double)raw * gain;
            const int scaled = clamp16(GUARD ? trunc_i32_ryujit(prod) : (int)prod);
            const int sm = scaled >> 31;
            const unsigned mag = (unsigned)((scaled ^ sm) - sm);
            const int aq = (int)((mag * 114692u + 2340u * 114692u) >> 29);
            const int q = (aq ^ sm) - sm;
            const int predicted = V4 ? (__mul24(b, c0) + __mul24(a, c1)) >> 12 : pa + pb;
            const int rec = clamp16(__mul24(scale, q) + predicted);
            a = b;
            b = rec;
            qv[j] = q;
        }
    };
    const double raw_bound = 32770.0 + 8.0 * (double)((c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1));
    if (__any(gain * raw_bound >= 2147483648.0)) quantise(std::true_type{});
    else quantise(std::false_type{});
#pragma unroll
    for (int w = 0; w < 8; w++)                  // frame bytes 2 + 2w, 3 + 2w: four nibbles, high first
        fw[1 + w] = (uint32_t)(((qv[4 * w] & 0xF) << 4) | (qv[4 * w + 1] & 0xF)) |
                    ((uint32_t)(((qv[4 * w + 2] & 0xF) << 4) | (qv[4 * w + 3] & 0xF)) << 8);
}

template <bool V4, bool EXPONENTIAL>
__device__ __forceinline__ void adx_encode_frame_serial(const int (&x)[32], int &a, int &b, int c0, int c1, int filter_bits,
                                                        uint8_t *fr)
{
    uint32_t fw[9];
    adx_encode_frame_words<V4, EXPONENTIAL>(x, a, b, c0, c1, filter_bits, fw);
#pragma unroll
    for (int i = 0; i < 9; i++) {
        fr[2 * i] = (uint8_t)(fw[i] & 0xff);
        fr[2 * i + 1] = (uint8_t)(fw[i] >> 8);
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
        const uint4 *p = reinterpret_cast<const uint4 *>(src + fc * 32);
#pragma unroll
        for (int i = 0; i < 4; i++) px[i] = p[i];
        const uint16_t *q = reinterpret_cast<const uint16_t *>(dst + fc * 18);
#pragma unroll
        for (int i = 0; i < 9; i++) fw[i] = q[i];
    };
    uint4 px[4], nx[4];
    uint32_t ow[9], nw[9];
    fetch(f0, px, ow);
    for (int64_t f = f0; f < f0 + seg_frames && f * 32 < total_length; f++) {
        fetch(f + 1, nx, nw);                           // in flight during this frame
        int x[32];
        if (f < full_frames) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t wv[4] = {px[i].x, px[i].y, px[i].z, px[i].w};
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    x[8 * i + 2 * t] = (int)(int16_t)(wv[t] & 0xFFFF);
                    x[8 * i + 2 * t + 1] = (int)wv[t] >> 16;
                }
            }
        } else {                                        // the zero-padded last frame: its own loads
#pragma unroll
            for (int j = 0; j < 32; j++) x[j] = f * 32 + j < total_length ? (int)src[f * 32 + j] : 0;
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
        adx_encode_frame_words<V4, EXPONENTIAL>(x, ta, tb, c0, c1, filter_bits, fw);
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

template <bool V4, bool EXPONENTIAL>
__global__ __launch_bounds__(64) void adx_encode_fs18_fixup_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_length, int seg_frames, AdxDeviceParams p,
    uint8_t *__restrict__ out, int64_t out_pitch, const int16_t *__restrict__ seg_state, int *__restrict__ first_open,
    int *__restrict__ seam_open, int *__restrict__ seam_end, int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const int k = blockIdx.y + 1;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (ch >= nch || f0 * 32 >= total_length) return;
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = out + (int64_t)ch * out_pitch;
    const int c0 = p.coef0, c1 = p.coef1;
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    int ta = seg_state[((int64_t)(k - 1) * nch + ch) * 2], tb = seg_state[((int64_t)(k - 1) * nch + ch) * 2 + 1];
    const int sa = src[f0 * 32 - 2], sb = src[f0 * 32 - 1];                       // the guessed run's start
    if (adx_encode_seam_run<V4, EXPONENTIAL>(src, dst, f0, seg_frames, total_length, c0, c1, filter_bits, ta, tb, sa, sb, ch, k,
                                             force_open)) {
        // still open at the end of its piece: the chain launch carries on from the history reached here
        seam_open[(int64_t)(k - 1) * nch + ch] = 1;
        seam_end[(int64_t)(k - 1) * nch + ch] = (int)(((unsigned)tb << 16) | ((unsigned)ta & 0xFFFFu));
        atomicMin(&first_open[ch], k);
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
    const int *__restrict__ seam_open, const int *__restrict__ seam_end, int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = out + (int64_t)ch * out_pitch;
    const int c0 = p.coef0, c1 = p.coef1;
    const int filter_bits = p.type == 2 ? ((p.filter << 5) & 0xff) : 0;
    bool carry = false;
    int ta = 0, tb = 0;
    for (int k = k0; k < segments; k++) {
        const int64_t f0 = (int64_t)k * seg_frames;
        if (f0 * 32 >= total_length) break;
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


int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length, const AdxDeviceParams &p,
                  uint8_t *d_out, int64_t out_pitch, int16_t *d_history_out, hipStream_t stream)
{
    if (nch <= 0) return VGA_OK;
    const dim3 grid((nch + 63) / 64), block(64);
    // pcm rows must be 16-byte aligned for the vector loads, output rows 4-byte aligned for the dword stores
    const bool fast = p.frame_size == 18 && p.padding == 0 && (pcm_pitch % 8) == 0 && ((uintptr_t)d_pcm % 16) == 0 &&
                      (out_pitch % 4) == 0 && ((uintptr_t)d_out % 4) == 0;
    if (fast) {
        const bool v4 = p.version == 4, ex = p.type == 4;
        const size_t lds = 2 * sizeof(AdxEncodeTile);
        // as many time segments as fill the device once, each an even number of frames and at least 8192 frames long: a
        // seam still open at the end of its piece sends its channel to the serial tail kernel (0.7 s for a 60 s
        // channel), and with 2048-frame pieces a 1024-channel launch (44 pieces, 44 000 seams) had such seams -- the host
        // pipeline's chunks took 250 ms each instead of 10.  The longest seam seen at configs[2] is ~1100 frames.
        // (the seams re-encode some hundred frames each)
        const int groups = (nch + ECW - 1) / ECW, groups64 = (nch + 63) / 64;
        const int cus = device_cu_count();
        const int frames = (pcm_length + 31) / 32;
        const int per_cu = (int)((160 * 1024) / lds) > 0 ? (int)((160 * 1024) / lds) : 1;
        int segments = cus * per_cu / groups;
        if (segments > frames / 8192) segments = frames / 8192;
        if (segments < 1) segments = 1;
        if (segments > 64) segments = 64;
        if (encoder_segments_override() > 0) segments = std::min(std::max(frames / 64, 1), encoder_segments_override());   // test hook
        int seg_frames = (frames + segments - 1) / segments;
        seg_frames += seg_frames & 1;
        AsyncBuf scratch;                              // freed (stream-ordered) on every exit path
        int16_t *seg_state = nullptr;                  // [segments][nch][2] final histories, then [nch] first open seam
        int *first_open = nullptr, *seam_open = nullptr, *seam_end = nullptr;
        if (segments > 1) {
            const size_t state_bytes = round_up((size_t)segments * nch * 2 * sizeof(int16_t), 16);
            const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
            VGA_HIP_TRY(scratch.alloc(state_bytes + (size_t)nch * sizeof(int) + 2 * flag_bytes, stream));
            seg_state = scratch.as<int16_t>();
            first_open = reinterpret_cast<int *>(scratch.as<unsigned char>() + state_bytes);
            seam_open = first_open + nch;
            seam_end = seam_open + (size_t)(segments - 1) * nch;
            VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
            VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes, stream));
        }
#define VGA_ADX_ENC_T(V, E)                                                                                              \
        {                                                                                                                \
            VGA_HIP_TRY(allow_dynamic_lds(adx_encode_fs18_tiled_kernel<V, E>, lds));                                     \
            hipLaunchKernelGGL((adx_encode_fs18_tiled_kernel<V, E>), dim3(groups, segments), dim3(ETHREADS), lds, stream, d_pcm, \
                               pcm_pitch, nch, pcm_length, seg_frames, p, d_out, out_pitch, d_history_out, seg_state);  \
            VGA_HIP_TRY(hipGetLastError());                                                                              \
            if (segments > 1) {                                                                                          \
                hipLaunchKernelGGL((adx_encode_fs18_fixup_kernel<V, E>), dim3(groups64, segments - 1), dim3(64), 0, stream, \
                                   d_pcm, pcm_pitch, nch, pcm_length, seg_frames, p, d_out, out_pitch, seg_state,       \
                                   first_open, seam_open, seam_end, force_open_seams());                                \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL((adx_encode_fs18_tail_kernel<V, E>), dim3(groups64), dim3(64), 0, stream, d_pcm,     \
                                   pcm_pitch, nch, pcm_length, seg_frames, segments, p, d_out, out_pitch, seg_state,    \
                                   first_open, seam_open, seam_end, force_open_seams());                                \
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
                  int16_t *d_pcm, int64_t pcm_pitch, int *d_status, hipStream_t stream)
{
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    const bool fast = p.frame_size == 18 && p.padding == 0 && (pcm_pitch % 8) == 0 && ((uintptr_t)d_pcm % 16) == 0 &&
                      (in_pitch % 4) == 0 && ((uintptr_t)d_adpcm % 4) == 0;
    if (fast) {
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
        int *first_open = nullptr, *seam_open = nullptr;
        if (segments > 1) {
            const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
            VGA_HIP_TRY(scratch.alloc((size_t)nch * sizeof(int) + flag_bytes, stream));
            first_open = scratch.as<int>();
            seam_open = first_open + nch;
            VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
            VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes, stream));
        }
#define VGA_ADX_DEC_T(V)                                                                                                 \
        {                                                                                                                \
            hipLaunchKernelGGL(adx_decode_fs18_direct_kernel<V>, dim3(groups, segments), dim3(64), 0, stream, d_adpcm,   \
                               in_pitch, nch, sample_count, seg_frames, p, d_pcm, pcm_pitch, d_status);                  \
            VGA_HIP_TRY(hipGetLastError());                                                                              \
            if (segments > 1) {                                                                                          \
                hipLaunchKernelGGL(adx_decode_fs18_fixup_kernel<V>, dim3(groups, segments - 1), dim3(64), 0, stream,    \
                                   d_adpcm, in_pitch, nch, sample_count, seg_frames, p, d_pcm, pcm_pitch, first_open,   \
                                   seam_open, force_open_seams());                                                      \
                VGA_HIP_TRY(hipGetLastError());                                                                          \
                hipLaunchKernelGGL(adx_decode_fs18_tail_kernel<V>, dim3(groups), dim3(64), 0, stream, d_adpcm, in_pitch, \
                                   nch, sample_count, seg_frames, segments, p, d_pcm, pcm_pitch, first_open, seam_open,  \
                                   force_open_seams());                                                                  \
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
