// gc_encode_kernel.hip -- GC-ADPCM frame encoder for gfx950 (the headline kernel).
//
// Replaces VGAudio/Codecs/GcAdpcm/GcAdpcmEncoder.cs:14-171 (Encode, DspEncodeFrame,
// DspEncodeCoef), bit-exact.
//
// ADPCM is a serial recurrence inside a channel (the reconstructed samples 12,13 of frame
// k are the history of frame k+1, :40-41; sample s+1 needs reconstructed sample s, :138,160),
// so time = frames x (length of the dependent chain per frame).  The design shortens that
// chain instead of chasing bandwidth:
//
//  * lane = (channel, predictor, scale candidate): 16 lanes per channel, 4 channels per
//    wave64, one wave per workgroup -> 1024 workgroups for 4096 channels, one per SIMD.
//  * Speculation on the retry loop (:127-170): the reference's first quantise pass is one
//    scale too small 93 % of the time and exactly right 6 %, so candidate A runs the pass
//    at scale s1 and candidate B at s1+1 IN PARALLEL LANES; resolve_candidates() decides
//    from the two overflow values which one the reference ends on.  The residual ~0.3 %
//    (overflow bumps, third pass) re-enters the literal loop under a wave-uniform branch.
//  * The quantise pass is integer-only (7 dependent VALU ops per sample instead of the
//    float/double detour); exactness is proven a posteriori per frame (gc_encode_core.hpp
//    S2/S3), otherwise the literal pass is re-run.
//  * The 14-sample pre-scan (:107-115) needs only input samples: the two candidate lanes
//    split it (7 samples each) and merge max/min with one DPP op each.
//  * 8-predictor argmin + winner-history broadcast: v_min_u32 / v_or_b32 with DPP operands
//    (quad_perm, row_half_mirror, row_mirror) -- 8 VALU ops, no LDS.
//  * PCM is read straight from the planar layout with 7 dword loads per frame, prefetched
//    one frame ahead (the loop is latency-bound at ~0.3 TB/s aggregate, far below HBM).
#include "common.hpp"
#include "gc_encode_core.hpp"
#include "gcadpcm_kernels.hpp"

#include <cstdlib>

namespace vga {
namespace gc {

constexpr int DPP_QUAD_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;  // i <-> 7-i inside each 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;       // i <-> 15-i inside each 16 lanes

template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// all-reduce over each 16-lane row
template <class Op>
__device__ __forceinline__ unsigned row16_reduce(unsigned v, Op op)
{
    v = op(v, (unsigned)dpp<DPP_QUAD_XOR1>((int)v));
    v = op(v, (unsigned)dpp<DPP_QUAD_XOR2>((int)v));
    v = op(v, (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)v));
    v = op(v, (unsigned)dpp<DPP_ROW_MIRROR>((int)v));
    return v;
}

__global__ __launch_bounds__(64) void gc_encode_kernel_v2(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int sample_count,
    const int16_t *__restrict__ coefs, const int16_t *__restrict__ hist1,
    const int16_t *__restrict__ hist2, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch)
{
    __shared__ uint32_t s_in[2][4][16][7];    // two 16-frame input tiles per channel group
    __shared__ uint2 s_out[4][16];            // one 16-frame output tile per channel group
    const int lane = threadIdx.x;
    const int grp = lane >> 4;
    const int l16 = lane & 15;
    const int p = l16 >> 1;
    const bool cand_b = (l16 & 1) != 0;
    const int ch_raw = blockIdx.x * 4 + grp;
    const bool live = ch_raw < nch;
    const int ch = live ? ch_raw : nch - 1;

    const int c0 = coefs[ch * 16 + 2 * p];
    const int c1 = coefs[ch * 16 + 2 * p + 1];
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = adpcm + (int64_t)ch * adpcm_pitch;

    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;

    int x[16];
    x[0] = hist2 ? hist2[ch] : 0;   // pcmBuffer[0] = History2 (GcAdpcmEncoder.cs:24)
    x[1] = hist1 ? hist1[ch] : 0;   // pcmBuffer[1] = History1 (:25)

    // One frame: x[2..15] already hold the 14 input samples (zero padded), x[0..1] the history.
    auto encode_frame = [&](int f, int slot, bool full) __attribute__((always_inline)) {
        // ---- pre-scan (:107-124), split over the two candidate lanes
        int s1;
        {
            int y[9];
#pragma unroll
            for (int t = 0; t < 9; t++) y[t] = cand_b ? x[7 + t] : x[t];
            int dmax = 0, dmin = 0;
#pragma unroll
            for (int s = 0; s < 7; s++) {
                const int predicted = (y[s] * c1 + y[s + 1] * c0) / 2048;
                const int d = y[s + 2] - predicted;
                dmax = imax(dmax, d);
                dmin = imin(dmin, d);
            }
            dmax = imax(dmax, dpp<DPP_QUAD_XOR1>(dmax));
            dmin = imin(dmin, dpp<DPP_QUAD_XOR1>(dmin));
            s1 = first_scale_power_from_range(dmax, dmin);
            if (__any(s1 == -100)) {
                if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
            }
        }

        // ---- speculative quantise pass: A at s1, B at s1+1
        int final_sp = imin(s1 + (cand_b ? 1 : 0), 12);
        PassOut r = pass_fast(x, c0, c1, final_sp);
        if (__any(!r.exact)) {
            if (!r.exact) r = pass_literal(x, c0, c1, final_sp);
        }
        const int ov_other = dpp<DPP_QUAD_XOR1>(r.max_overflow);
        const int ov_a = cand_b ? ov_other : r.max_overflow;
        const int ov_b = cand_b ? r.max_overflow : ov_other;
        Resolve z = resolve_candidates_nobump(s1, ov_a, ov_b);
        if (__any(imax(ov_a, ov_b) > 248)) z = resolve_candidates(s1, ov_a, ov_b);   // bump loop: rare
        bool fin = cand_b ? z.final_b : z.final_a;
        const bool resume = !cand_b && !z.final_a && !z.final_b;
        if (__any(resume)) {
            if (resume) {
                r = resume_passes(x, c0, c1, z.resume_sp, final_sp);
                fin = true;
            }
        }

        // ---- argmin over the 8 predictors, first index wins ties (:66-76)
        int winner;
        const bool wide = __any(fin && r.total >= (1ull << 28));
        if (!wide) {
            const unsigned key = fin ? (((unsigned)r.total << 4) | (unsigned)l16) : 0xFFFFFFFFu;
            const unsigned best = row16_reduce(key, [](unsigned a, unsigned b) { return a < b ? a : b; });
            winner = (int)(best & 15u);
        } else {
            uint64_t key = fin ? ((r.total << 4) | (uint64_t)l16) : ~0ull;
#define VGA_MIN64_STAGE(CTRL)                                                              \
            {                                                                              \
                const unsigned olo = (unsigned)dpp<CTRL>((int)(uint32_t)key);              \
                const unsigned ohi = (unsigned)dpp<CTRL>((int)(uint32_t)(key >> 32));      \
                const uint64_t okey = ((uint64_t)ohi << 32) | olo;                         \
                key = okey < key ? okey : key;                                             \
            }
            VGA_MIN64_STAGE(DPP_QUAD_XOR1)
            VGA_MIN64_STAGE(DPP_QUAD_XOR2)
            VGA_MIN64_STAGE(DPP_ROW_HALF_MIRROR)
            VGA_MIN64_STAGE(DPP_ROW_MIRROR)
#undef VGA_MIN64_STAGE
            winner = (int)(key & 15u);
        }
        const bool won = l16 == winner;
        const unsigned pay = row16_reduce(won ? ((unsigned)(r.o12 & 0xFFFF) | ((unsigned)r.o13 << 16)) : 0u,
                                          [](unsigned a, unsigned b) { return a | b; });

        if (won) {
            uint32_t d0, d1;
            frame_words(r, p, final_sp, d0, d1);
            if (full) {
                s_out[grp][slot] = make_uint2(d0, d1);          // flushed 16 frames at a time
            } else if (live) {
                // partial last frame: SampleCountToByteCount(tail) bytes (:38)
                const int nbytes = (tail + 2 + 1) / 2;
                const uint64_t both = ((uint64_t)d1 << 32) | d0;
                for (int b = 0; b < nbytes; b++) dst[(int64_t)f * 8 + b] = (uint8_t)(both >> (8 * b));
            }
        }
        x[0] = (int)(int16_t)(pay & 0xFFFF);   // pcmBuffer[0] = pcmBuffer[14] (:40)
        x[1] = (int)pay >> 16;                 // pcmBuffer[1] = pcmBuffer[15] (:41)
    };

    auto unpack = [&](const uint32_t (&w)[7]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 7; i++) {
            x[2 + 2 * i] = (int)(int16_t)(w[i] & 0xFFFF);
            x[3 + 2 * i] = (int)w[i] >> 16;
        }
    };
    // Block pipeline over 16-frame tiles (all global traffic is coalesced and off the per-frame path):
    //   * lane l fetches frame (l & 15) of its channel's NEXT tile: 7 dwords, 16 lanes = 448
    //     contiguous bytes per channel, issued a whole tile (~16 x 0.4 us) before they are needed;
    //   * the tile being encoded is read from LDS (same address for the 16 lanes of a channel ->
    //     broadcast), ping-ponged one frame ahead in registers;
    //   * winners drop their 8-byte frames into LDS; every 16 frames lane l stores frame (l & 15)
    //     of its channel: 128 contiguous bytes per channel.
    auto tile_load = [&](uint32_t (&w)[7], int tile) __attribute__((always_inline)) {
        const int fr = imin(tile * 16 + l16, full_frames - 1);       // clamped: stays inside the channel
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + (int64_t)fr * 14);
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = p32[i];
    };
    auto tile_to_lds = [&](const uint32_t (&w)[7], int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 7; i++) s_in[buf][grp][l16][i] = w[i];
    };
    auto lds_frame = [&](uint32_t (&w)[7], int buf, int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = s_in[buf][grp][j][i];
    };

    if (full_frames > 0) {
        const int tiles = (full_frames + 15) / 16;
        uint32_t stage[7];
        tile_load(stage, 0);
        tile_to_lds(stage, 0);
        if (tiles > 1) tile_load(stage, 1);
        __syncthreads();
        for (int tile = 0; tile < tiles; tile++) {
            const int cur = tile & 1;
            const int fb = tile * 16;
            const int nf = imin(16, full_frames - fb);
            uint32_t wa[7], wb[7];
            lds_frame(wa, cur, 0);
            int j = 0;
            for (; j + 1 < nf; j += 2) {
                lds_frame(wb, cur, j + 1);
                unpack(wa);
                encode_frame(fb + j, j, true);
                lds_frame(wa, cur, imin(j + 2, 15));
                unpack(wb);
                encode_frame(fb + j + 1, j + 1, true);
            }
            if (j < nf) {
                unpack(wa);
                encode_frame(fb + j, j, true);
            }
            // next tile into the idle LDS buffer first (its loads are a whole tile old), the flush
            // store last, so no vmcnt wait ever sits behind a store that was just issued
            if (tile + 1 < tiles) {
                tile_to_lds(stage, cur ^ 1);
                if (tile + 2 < tiles) tile_load(stage, tile + 2);
            }
            __syncthreads();
            if (live && fb + l16 < full_frames)
                *reinterpret_cast<uint2 *>(dst + (int64_t)(fb + l16) * 8) = s_out[grp][l16];
        }
    }
    if (tail) {
#pragma unroll
        for (int s = 0; s < 14; s++) x[2 + s] = (s < tail) ? (int)src[(int64_t)full_frames * 14 + s] : 0;
        encode_frame(full_frames, 0, false);
    }
}


// =====================================================================================================
// gc_encode_kernel -- serial wave + helper wave per workgroup (128 threads, 4 channels).
// Everything that does not depend on the reconstructed history is taken off the serial wave:
//   helper wave (wave 1), one 16-frame tile AHEAD of the encoder:
//     * coalesced global loads of the tile (lane = frame j of channel g: 28 contiguous bytes),
//     * unpack to int32, x*2048, and for each of the 8 predictors the max/min pre-scan distance over the
//       twelve samples s = 2..13 that involve input samples only (GcAdpcmEncoder.cs:107-115),
//     * all of it into LDS (double-buffered), plus the coalesced flush of the previous tile's frames
//       (the zero-padded partial last frame travels through the same path);
//   serial wave (wave 0): per frame 9 LDS reads, the two history-dependent pre-scan distances, the
//     speculative quantise pass, candidate resolution, DPP argmin, one LDS write.
// The hot loop is ONE copy of the frame body (a few KB of code): the reference's third-and-later
// quantise passes re-enter the same pass code through a wave-uniform loop, and the never-on-audio
// fallbacks (literal f32/f64 pass, sequential pre-scan tie-break) are out-of-line functions.  Measured:
// compiling the rare paths inline cost 80 ms of 257 ms (code size, register pressure, branches).
// The SIMDs have idle issue slots next to a lone latency-bound wave (tools/ubench_valu.hip: two waves
// per SIMD do not slow each other's dependent chains), so the helper costs the encoder nothing.
#ifndef VGA_ENC_TILE
#define VGA_ENC_TILE 16
#endif
constexpr int TF = VGA_ENC_TILE;   // frames per tile (<= 16: one helper lane per frame and channel)
struct GcTile {
    int x[4][TF][16];          // [channel group][frame][sample]  (14 used)
    int in2048[4][TF][16];     // x * 2048
    uint32_t pre[4][TF][8];    // per predictor: clamp16(max d) & 0xFFFF | clamp16(min d) << 16, over s = 2..13
};
#ifdef VGA_ENC_MARKS   // analysis builds: region markers in the assembly listing
#define VGA_MARK(name) asm volatile("; MARK " name)
#else
#define VGA_MARK(name)
#endif
struct X16 { int v[16]; };
struct ResumeOut { PassOut r; int final_sp; };

#ifdef VGA_ENC_COLD_OUTLINE      // experiment switch: measured 267 ms out of line vs 260 ms inline at configs[1]
#define VGA_COLD __device__ __noinline__
#else
#define VGA_COLD __device__ __forceinline__
#endif

// Rare paths, out of line so the per-frame loop stays small (arguments by value: the hot copy of the
// frame stays in registers, the cold copy may live wherever the callee likes).
VGA_COLD PassOut pass_literal_cold(X16 xs, int c0, int c1, int scale_power)
{
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = xs.v[i];
    return pass_literal(x, c0, c1, scale_power);
}
VGA_COLD int prescan_sequential_cold(X16 xs, int c0, int c1)
{
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = xs.v[i];
    return prescan_sequential(x, c0, c1);
}
VGA_COLD ResumeOut resume_cold(X16 xs, int c0, int c1, int scale_power)
{
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = xs.v[i];
    ResumeOut o;
    o.r = resume_passes(x, c0, c1, scale_power, o.final_sp);
    return o;
}
VGA_COLD Resolve resolve_cold(int s1, int ov_a, int ov_b) { return resolve_candidates(s1, ov_a, ov_b); }

__global__ __launch_bounds__(128) void gc_encode_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int sample_count,
    const int16_t *__restrict__ coefs, const int16_t *__restrict__ hist1,
    const int16_t *__restrict__ hist2, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch)
{
    __shared__ GcTile s_tile[2];
    __shared__ uint2 s_out[2][4][TF];
    const int tid = threadIdx.x;
    const bool helper = tid >= 64;
    const int lane = tid & 63;
    const int grp = lane >> 4;
    const int l16 = lane & 15;
    const int ch_raw = blockIdx.x * 4 + grp;
    const bool live = ch_raw < nch;
    const int ch = live ? ch_raw : nch - 1;
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    uint8_t *dst = adpcm + (int64_t)ch * adpcm_pitch;

    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    const int frames = full_frames + (tail ? 1 : 0);
    const int tiles = (frames + TF - 1) / TF;

    if (helper) {
        // ---------------------------------------------------------------- helper wave
        int cf[16];
#pragma unroll
        for (int i = 0; i < 16; i++) cf[i] = coefs[ch * 16 + i];
        auto prepare = [&](int tile) {
            if (l16 >= TF) return;
            const int fr = imin(tile * TF + l16, frames - 1);
            int in[14];
            if (fr < full_frames) {
                const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + (int64_t)fr * 14);
                uint32_t w[7];
#pragma unroll
                for (int i = 0; i < 7; i++) w[i] = p32[i];
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    in[2 * i] = (int)(int16_t)(w[i] & 0xFFFF);
                    in[2 * i + 1] = (int)w[i] >> 16;
                }
            } else {                                   // zero-padded partial last frame (:32-33)
#pragma unroll
                for (int s = 0; s < 14; s++) in[s] = (s < tail) ? (int)src[(int64_t)fr * 14 + s] : 0;
            }
            GcTile &T = s_tile[tile & 1];
            int4 *xr = reinterpret_cast<int4 *>(&T.x[grp][l16][0]);
            int4 *mr = reinterpret_cast<int4 *>(&T.in2048[grp][l16][0]);
            xr[0] = make_int4(in[0], in[1], in[2], in[3]);
            xr[1] = make_int4(in[4], in[5], in[6], in[7]);
            xr[2] = make_int4(in[8], in[9], in[10], in[11]);
            xr[3] = make_int4(in[12], in[13], 0, 0);
            mr[0] = make_int4(in[0] * 2048, in[1] * 2048, in[2] * 2048, in[3] * 2048);
            mr[1] = make_int4(in[4] * 2048, in[5] * 2048, in[6] * 2048, in[7] * 2048);
            mr[2] = make_int4(in[8] * 2048, in[9] * 2048, in[10] * 2048, in[11] * 2048);
            mr[3] = make_int4(in[12] * 2048, in[13] * 2048, 0, 0);
            uint32_t pre[8];
#pragma unroll
            for (int p = 0; p < 8; p++) {
                const int c0 = cf[2 * p], c1 = cf[2 * p + 1];
                int dmax = 0, dmin = 0;
#pragma unroll
                for (int s = 2; s < 14; s++) {          // x[s] = in[s-2], x[s+1] = in[s-1], x[s+2] = in[s]
                    const int predicted = (in[s - 2] * c1 + in[s - 1] * c0) / 2048;
                    const int d = in[s] - predicted;
                    dmax = imax(dmax, d);
                    dmin = imin(dmin, d);
                }
                pre[p] = (uint32_t)(clamp16i(dmax) & 0xFFFF) | ((uint32_t)clamp16i(dmin) << 16);
            }
            uint4 *pr = reinterpret_cast<uint4 *>(&T.pre[grp][l16][0]);
            pr[0] = make_uint4(pre[0], pre[1], pre[2], pre[3]);
            pr[1] = make_uint4(pre[4], pre[5], pre[6], pre[7]);
        };
        auto flush = [&](int tile) {
            const int fr = tile * TF + l16;
            if (!live || l16 >= TF || fr >= frames) return;
            const uint2 v = s_out[tile & 1][grp][l16];
            if (fr < full_frames) {
                *reinterpret_cast<uint2 *>(dst + (int64_t)fr * 8) = v;
            } else {
                // partial last frame: SampleCountToByteCount(tail) bytes (:38)
                const int nbytes = (tail + 2 + 1) / 2;
                const uint64_t both = ((uint64_t)v.y << 32) | v.x;
                for (int b = 0; b < nbytes; b++) dst[(int64_t)fr * 8 + b] = (uint8_t)(both >> (8 * b));
            }
        };
        if (tiles > 0) prepare(0);
        __syncthreads();
        for (int tile = 0; tile < tiles; tile++) {
            if (tile + 1 < tiles) prepare(tile + 1);
            if (tile > 0) flush(tile - 1);
            __syncthreads();
        }
        if (tiles > 0) flush(tiles - 1);
        return;
    }

    // -------------------------------------------------------------------- serial (encoder) wave
    const int p = l16 >> 1;
    const bool cand_b = (l16 & 1) != 0;
    const int c0 = coefs[ch * 16 + 2 * p];
    const int c1 = coefs[ch * 16 + 2 * p + 1];
    int h0 = hist2 ? hist2[ch] : 0;   // pcmBuffer[0] = History2 (GcAdpcmEncoder.cs:24)
    int h1 = hist1 ? hist1[ch] : 0;   // pcmBuffer[1] = History1 (:25)

    struct Row { int x[16]; int m[14]; uint32_t pre; };
    auto read_row = [&](const GcTile &T, int j, Row &R) {
        const int4 *xr = reinterpret_cast<const int4 *>(&T.x[grp][j][0]);
        const int4 *mr = reinterpret_cast<const int4 *>(&T.in2048[grp][j][0]);
        const int4 a0 = xr[0], a1 = xr[1], a2 = xr[2], a3 = xr[3];
        const int4 b0 = mr[0], b1 = mr[1], b2 = mr[2], b3 = mr[3];
        int *x = R.x, *m = R.m;
        x[2] = a0.x; x[3] = a0.y; x[4] = a0.z; x[5] = a0.w; x[6] = a1.x; x[7] = a1.y; x[8] = a1.z; x[9] = a1.w;
        x[10] = a2.x; x[11] = a2.y; x[12] = a2.z; x[13] = a2.w; x[14] = a3.x; x[15] = a3.y;
        m[0] = b0.x; m[1] = b0.y; m[2] = b0.z; m[3] = b0.w; m[4] = b1.x; m[5] = b1.y; m[6] = b1.z; m[7] = b1.w;
        m[8] = b2.x; m[9] = b2.y; m[10] = b2.z; m[11] = b2.w; m[12] = b3.x; m[13] = b3.y;
        R.pre = T.pre[grp][j][p];
    };
    auto pack = [](const int (&x)[16]) {
        X16 xs;
#pragma unroll
        for (int i = 0; i < 16; i++) xs.v[i] = x[i];
        return xs;
    };

    auto encode_frame = [&](Row &R, int buf, int j) {
        int (&x)[16] = R.x;
        VGA_MARK("frame_begin");
        x[0] = h0;
        x[1] = h1;
        // ---- pre-scan (:107-124): two history-dependent distances + the helper's range for s = 2..13
        int s1;
        {
            const int d0 = x[2] - (VGA_MUL24(x[0], c1) + VGA_MUL24(x[1], c0)) / 2048;
            const int d1 = x[3] - (VGA_MUL24(x[1], c1) + VGA_MUL24(x[2], c0)) / 2048;
            const int dmax = imax(imax((int)(int16_t)(R.pre & 0xFFFF), d0), d1);
            const int dmin = imin(imin((int)R.pre >> 16, d0), d1);
            s1 = first_scale_power_from_range(dmax, dmin);
#if !defined(VGA_EXPERIMENT_NO_COLD) && !defined(VGA_X_NO_TIE)
            if (__any(s1 == -100)) {                   // +M and -M both present: first occurrence decides
                if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential_cold(pack(x), c0, c1));
            }
#endif
        }
        VGA_MARK("prescan_end");
        // ---- first trip: candidate A at s1, B at s1+1 (speculation on the loop of :127-170)
        int final_sp = imin(s1 + (cand_b ? 1 : 0), 12);
        PassOut r = pass_fast_core(x, R.m, c0, c1, final_sp);
#if !defined(VGA_EXPERIMENT_NO_COLD) && !defined(VGA_X_NO_LITERAL)
        if (__any(!r.exact)) {                         // 32-bit error sum not provably exact: literal pass
            if (!r.exact) r = pass_literal_cold(pack(x), c0, c1, final_sp);
        }
#endif
        VGA_MARK("pass_end");
        const int ov_other = dpp<DPP_QUAD_XOR1>(r.max_overflow);
        const int ov_a = cand_b ? ov_other : r.max_overflow;
        const int ov_b = cand_b ? r.max_overflow : ov_other;
        Resolve z = resolve_candidates_nobump(s1, ov_a, ov_b);
#if !defined(VGA_EXPERIMENT_NO_COLD) && !defined(VGA_X_NO_BUMP)
        if (__any(imax(ov_a, ov_b) > 248)) z = resolve_cold(s1, ov_a, ov_b);   // scale bumps (:160-168): rare
#endif
        bool fin = cand_b ? z.final_b : z.final_a;
#if !defined(VGA_EXPERIMENT_NO_COLD) && !defined(VGA_X_NO_RESUME)
        // ---- third and later trips (about 10 % of wave-frames): the A lane of the pair carries on
        const bool resume = !cand_b && !z.final_a && !z.final_b;
        if (__any(resume)) {
            if (resume) {
                const ResumeOut o = resume_cold(pack(x), c0, c1, z.resume_sp);
                r = o.r;
                final_sp = o.final_sp;
                fin = true;
            }
        }
#endif
        VGA_MARK("resolve_end");
        // ---- argmin over the 8 predictors, first index wins ties (:66-76)
        int winner;
        if (!__any(fin && r.total >= (1ull << 28))) {
            const unsigned key = fin ? (((unsigned)r.total << 4) | (unsigned)l16) : 0xFFFFFFFFu;
            const unsigned best = row16_reduce(key, [](unsigned a, unsigned b) { return a < b ? a : b; });
            winner = (int)(best & 15u);
        } else {
            uint64_t key = fin ? ((r.total << 4) | (uint64_t)l16) : ~0ull;
#define VGA_MIN64_STAGE(CTRL)                                                          \
            {                                                                          \
                const unsigned olo = (unsigned)dpp<CTRL>((int)(uint32_t)key);          \
                const unsigned ohi = (unsigned)dpp<CTRL>((int)(uint32_t)(key >> 32));  \
                const uint64_t okey = ((uint64_t)ohi << 32) | olo;                     \
                key = okey < key ? okey : key;                                         \
            }
            VGA_MIN64_STAGE(DPP_QUAD_XOR1)
            VGA_MIN64_STAGE(DPP_QUAD_XOR2)
            VGA_MIN64_STAGE(DPP_ROW_HALF_MIRROR)
            VGA_MIN64_STAGE(DPP_ROW_MIRROR)
#undef VGA_MIN64_STAGE
            winner = (int)(key & 15u);
        }
        const bool won = l16 == winner;
        const unsigned pay = row16_reduce(won ? ((unsigned)(r.o12 & 0xFFFF) | ((unsigned)r.o13 << 16)) : 0u,
                                          [](unsigned a, unsigned b) { return a | b; });
        if (won) {
            uint32_t d0, d1;
            frame_words(r, p, final_sp, d0, d1);
            s_out[buf][grp][j] = make_uint2(d0, d1);          // flushed by the helper, 16 frames at a time
        }
        h0 = (int)(int16_t)(pay & 0xFFFF);   // pcmBuffer[0] = pcmBuffer[14] (:40)
        h1 = (int)pay >> 16;                 // pcmBuffer[1] = pcmBuffer[15] (:41)
        VGA_MARK("frame_end");
    };

    __syncthreads();                                   // tile 0 prepared
    for (int tile = 0; tile < tiles; tile++) {
        const int buf = tile & 1;
        const int nf = imin(TF, frames - tile * TF);
        const GcTile &T = s_tile[buf];
#ifdef VGA_ENC_SINGLE
#pragma unroll 1
        for (int j = 0; j < nf; j++) {
            Row R;
            read_row(T, j, R);
            encode_frame(R, buf, j);
        }
#else
        // two row register sets, ping-pong: the LDS reads of frame j+1 are in flight during frame j
        Row RA, RB;
        read_row(T, 0, RA);
#pragma unroll 1
        for (int j = 0; j < nf; j += 2) {
            read_row(T, imin(j + 1, TF - 1), RB);
            encode_frame(RA, buf, j);
            if (j + 1 < nf) {
                read_row(T, imin(j + 2, TF - 1), RA);
                encode_frame(RB, buf, j + 1);
            }
        }
#endif
        __syncthreads();                               // tile done: helper may flush it and refill this buffer later
    }
}

int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                  const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                  hipStream_t stream)
{
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    // A/B switch for measurements only: VGA_GC_ENCODE_IMPL=v1 selects the first (literal) kernel
    static const bool use_v1 = [] {
        const char *e = getenv("VGA_GC_ENCODE_IMPL");
        return e && e[0] == 'v' && e[1] == '1';
    }();
    if (use_v1)
        return launch_encode_v1(d_pcm, pcm_pitch, nch, sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch,
                                stream);
    static const bool use_v2 = [] {
        const char *e = getenv("VGA_GC_ENCODE_IMPL");
        return e && e[0] == 'v' && e[1] == '2';
    }();
    if (use_v2)
        hipLaunchKernelGGL(gc_encode_kernel_v2, dim3((nch + 3) / 4), dim3(64), 0, stream, d_pcm, pcm_pitch, nch,
                           sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch);
    else
        hipLaunchKernelGGL(gc_encode_kernel, dim3((nch + 3) / 4), dim3(128), 0, stream, d_pcm, pcm_pitch, nch,
                           sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace gc
}  // namespace vga
