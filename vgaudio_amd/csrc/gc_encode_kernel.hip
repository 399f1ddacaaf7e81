// gc_encode_kernel.hip -- GC-ADPCM frame encoder for gfx950 (the headline kernel).
//
// Replaces VGAudio/Codecs/GcAdpcm/GcAdpcmEncoder.cs:14-171 (Encode, DspEncodeFrame,
// DspEncodeCoef), bit-exact.
//
// ADPCM is a serial recurrence inside a channel (the reconstructed samples 12,13 of frame
// k are the history of frame k+1, :40-41; sample s+1 needs reconstructed sample s, :138,160),
// so time = frames x (wave-instructions per frame): the kernel is bound by VALU issue (one
// instruction per 4 cycles per SIMD for this mix; LABNOTES.md 4.0), not by HBM.
//
//  * lane = (channel, predictor, scale candidate): 16 lanes per channel, 4 channels per
//    encoder wave; a helper wave per workgroup prepares 16-frame tiles in LDS (below).
//  * Speculation on the retry loop (:127-170): the reference's first quantise pass is one
//    scale too small 93 % of the time and exactly right 6 %, so candidate A runs the pass
//    at scale s1 and candidate B at s1+1 IN PARALLEL LANES; two compares on the exchanged
//    overflow values decide which one the reference ends on.  Third trips, overflow bumps
//    and everything else that is rare sit behind ONE wave-uniform branch per frame.
//  * The quantise pass is integer-only (16 VALU ops per sample, 8 on the dependent chain,
//    instead of the float/double detour); exactness conditions in gc_encode_core.hpp S2/S3.
//  * 8-predictor argmin + winner-history broadcast: v_min_u32 / v_or_b32 with DPP operands
//    (quad_perm, row_half_mirror, row_mirror) -- 8 VALU ops, no LDS.
#include "common.hpp"
#include "gc_encode_core.hpp"
#include "gcadpcm_kernels.hpp"

#include <cstdio>
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

__device__ __forceinline__ unsigned umin32(unsigned a, unsigned b) { return a < b ? a : b; }

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


// =====================================================================================================
// gc_encode_kernel -- SW encoder (serial) waves + one helper wave per workgroup (4 channels per encoder wave).
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
// measured at configs[1]: 1 encoder wave -> 209.5 ms, 2 -> 200.1 ms (two pieces per channel)
#ifdef VGA_DEBUG_TIMESTAMPS
// tools/time_wave_ends.py (a -DVGA_DEBUG_TIMESTAMPS build under tools/variants/ only): when every workgroup's encoder wave
// 0 started and ended, in ticks of the 100 MHz wall clock
__device__ unsigned long long g_vga_enc_ts[1 << 16];
extern "C" int vga_debug_encode_timestamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vga_enc_ts), (size_t)n * sizeof(unsigned long long));
}
#endif
// Diagnostics of the data-dependent parts (vga_testing_gc_encode_stats, include/vgaudio_hip_testing.h; bench.py's
// signal_sensitivity block): counted with a handful of atomics per piece and per seam -- nothing in the frame loop but one
// scalar add inside the cold block.  Summed over every launch of the process since the last reset.
//   [0] seams closed inside their piece   [1] seams left open for gc_encode_chain_kernel   [2] frames re-encoded by seam runs
//   [3] wave-frames encoded by the piece kernels   [4] ... of which took the cold block (third trips, bump loop, inexact sums;
//   counted by -DVGA_GC_STATS builds only -- tools/build_variants.sh stats:"-DVGA_GC_STATS" -- 0 otherwise, [7] says which)
//   [5] channels gc_encode_chain_kernel had to walk   [6] pieces encoded   [7] 1 = this build counts [4]
__device__ unsigned long long g_vga_gc_stats[8];
extern "C" int vga_testing_gc_encode_stats(unsigned long long *out8, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_vga_gc_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
#ifdef VGA_GC_STATS
    if (out8) out8[7] = 1;
#else
    if (out8) out8[7] = 0;
#endif
    if (reset) {
        const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_vga_gc_stats), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
constexpr int SW = 2;              // encoder (serial) waves per workgroup; one helper wave serves them all
constexpr int ENC_THREADS = 64 * (SW + 1);
// issue priorities (s_setprio) of the encoder waves and of the helper wave.  Round 6 (profiles/r06_t_encode_priority*.log):
// helper at 1 or 2, or at priorities that differ between the helpers of a SIMD (by wave slot, rotating per piece or per
// tile): 143.7-144.9 ms against 144.0-144.4; helper at 3: 148.5; encoder waves below the helper: 150 ms.
#ifndef VGA_GC_ENCODER_PRIO
#define VGA_GC_ENCODER_PRIO 3
#endif
#ifndef VGA_GC_HELPER_PRIO
#define VGA_GC_HELPER_PRIO 0
#endif
// Two lane layouts of the encoder wave (template parameter CPW = channels per encoder wave):
//   CPW = 4: lane = (channel, predictor, scale candidate A/B) -- the candidates of the retry loop run side by side;
//   CPW = 8: lane = (channel, predictor) -- candidate B (s1 + 1) and candidate A (s1) run one after the other in the
//            same lane.  Twice the channels share every per-frame fixed cost (row reads, pre-scan, resolution,
//            argmin, the cold block's pass), which is worth more than the second pass costs: LABNOTES.md 4.1.
template <int CPW>
struct Lay {
    static constexpr int CS = CPW * SW;    // channel slots per workgroup
    static constexpr int TF = 64 / CS;     // frames per tile: one helper lane per (channel slot, frame)
};
template <int CS, int TF>
struct GcTileT {
    int x[CS][TF][16];         // [channel slot][frame][sample]  (14 used)
    int in2048[CS][TF][16];    // x * 2048
    int in2048p[CS][TF][16];   // x * 2048 + 1024
    uint32_t pre[CS][TF][8];   // per predictor: clamp16(max d) & 0xFFFF | clamp16(min d) << 16, over s = 2..13
};
typedef short short2v __attribute__((ext_vector_type(2)));
struct X16 { int v[16]; };

// inline: measured 267 ms out of line vs 260 ms inline at configs[1] (round 1)
#define VGA_COLD __device__ __forceinline__

// Rare +M/-M tie of the pre-scan (argument by value: the hot copy of the frame stays in registers).
VGA_COLD int prescan_sequential_cold(X16 xs, int c0, int c1)
{
    int x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = xs.v[i];
    return prescan_sequential(x, c0, c1);
}

#ifdef VGA_GC_R05_COLD
#define VGA_COLD_OPAQUE(v) ((void)0)
#else
#define VGA_COLD_OPAQUE(v) VGA_OPAQUE(v)
#endif
// Third and later trips / the generic redo for one frame.  Its mere presence in the frame loop costs the hot
// path (a build with the block compiled in but never run: 172 ms vs 140 ms without it), which is why the
// frame's tail is instantiated once per branch below instead of merging the two branches' results.
struct ColdState {
    int x[16], m[14], mp[14];
    int c0, c1, s1;
    // what this lane needs (any of them in any lane sends the wave here):
    //   generic: the reference's loop as written from `start` (the scalePower its loop continues from): the bump loop
    //            (:166-168) was entered, or the coefficients can wrap int32
    //   drop:    (lane-per-candidate layout) the pair's other lane redoes the whole loop: this lane is out
    //   wide:    the final pass ran at the cap with an overflow above 3: same pass again with a 64-bit error sum
    //   resume:  third and later trips
    // start: -100 = not known yet: the loop stands behind the bumps of the pass that started them (bump_a: the pass at s1
    // with overflow ov_a, else the pass at s1 + 1 with ov_b) -- worked out only by a lane that goes that way (round 6: the
    // bump loops ran, under an empty exec mask as a rule, on every visit of the cold block)
    int generic, start, drop, wide, resume;
    int bump_a, ov_a, ov_b;
};
struct ColdOut { PassOut r; int final_sp; int fin; };
// inline: 370 ms out of line (the by-value state goes through scratch) vs 209 inline (round 1)
__device__ __forceinline__
ColdOut encode_frame_cold(ColdState st, PassOut r, int final_sp, int fin)
{
    int x[16], m[14], mp[14];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = st.x[i];
#pragma unroll
    for (int i = 0; i < 14; i++) { m[i] = st.m[i]; mp[i] = st.mp[i]; }
    if (st.drop) fin = 0;
    if (__any((st.generic | st.wide) != 0)) {      // (one test in front of both: the common visitor is a third trip)
    if (__any(st.generic != 0)) {                  // hostile input, and tones the first scale misjudges by 2^5 and more
        if (st.generic) {
            // (the frame is made opaque here: hipcc otherwise hoists the literal pass's set-up -- fourteen shifts, 64-bit
            // mads -- out of this branch into every visit of the cold block)
            int xg[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { xg[i] = x[i]; VGA_COLD_OPAQUE(xg[i]); }
            const int start = st.start != -100 ? st.start
                                               : (st.bump_a ? apply_bumps(st.s1, st.ov_a) : apply_bumps(st.s1 + 1, st.ov_b));
            r = resume_passes(xg, st.c0, st.c1, start, final_sp);
            fin = 1;
        }
    }
    // Round 5: until then any lane whose pass at the cap overflowed by more than 3 sent the whole wave through the
    // reference's loop from its start (three to four passes, the literal f32 / f64 one among them) -- 96 % of the wave-frames of
    // full-scale white noise and of a clipped square (bench.py signal_sensitivity: 316 and 343 ms against 146).  The pass
    // itself is right (gc_encode_core.hpp S2 holds for every int32 distance); only its 32-bit error sum may have wrapped.
    if (__any(st.wide != 0)) {
        if (st.wide) {
            const PassOut w = pass_fast_core_wide(x, m, mp, st.c0, st.c1, final_sp);
            r.total = w.total;
            fin = 1;
        }
    }
    }
    if (__any(st.resume != 0)) {
        if (st.resume) {
            // third and later trips, same straight-line tests as the first trip
            int sp = st.s1 + 1;                    // < 12: neither candidate was at the cap
            for (;;) {
                sp++;
#if defined(VGA_GC_NO_FAST_PASSES) || defined(VGA_GC_NO_FAST_THIRD)  // (timing-only switches, tools/build_variants.sh)
                bool short_pass = false;
#else
                bool short_pass = !__any(sp > 9);  // (over the lanes still in this loop) without the f32 detour, as the first two passes
#endif
                if (short_pass) {
                    r = pass_fast_core_no_round(x, m, mp, st.c0, st.c1, sp);
                    short_pass = !__any(!pass_no_round_is_exact(sp, r.max_overflow));
                }
                if (!short_pass) r = pass_fast_core(x, m, mp, st.c0, st.c1, sp);
                const bool cap = sp >= 12;
                if ((unsigned)r.max_overflow > (cap ? 3u : 248u)) {      // bump loop / inexact sum: generic
                    int xg[16];                                          // (opaque: see above)
#pragma unroll
                    for (int i = 0; i < 16; i++) { xg[i] = x[i]; VGA_COLD_OPAQUE(xg[i]); }
                    r = resume_passes(xg, st.c0, st.c1, sp - 1, final_sp);
                    break;
                }
                final_sp = sp;
                if (cap || r.max_overflow <= 1) break;
            }
            fin = 1;
        }
    }
    ColdOut o;
    o.r = r;
    o.final_sp = final_sp;
    o.fin = fin;
    return o;
}

// Time segments (blockIdx.y; LABNOTES.md 4.3): with fewer channels than fill the chip, a channel's stream is cut into
// pieces of `seg_frames` frames encoded side by side.  Piece 0 starts from the caller's history, the others from a
// guess -- the two INPUT samples before the piece -- and gc_encode_seam_kernel closes the seams afterwards.
// seg_state[piece][channel] receives every piece's final history.  At BASELINE configs[1] there is one piece.
// REPAIR is a template parameter only so that the two launches carry different names in profiles (the repair launch
// normally returns at once and would halve the kernel's average duration).
// RAGGED (the `*_v` entry points): channel slots map to channels through rg.order (longest first: slot 0 of a workgroup
// holds its longest channel), every slot has its own length and offsets, a piece exists for a slot only as far as its
// channel reaches.  The workgroup runs as many tiles as its longest slot needs; a slot whose frames have run out keeps
// encoding (clamped loads, nothing flushed) with its history frozen.
// One (channel group, time piece) = what a workgroup of the plain launch does: gc_encode_kernel calls it once with its
// block indices, gc_encode_persistent_kernel once per item it takes from the queue.
template <bool REPAIR, int CPW, bool RAGGED>
__device__ __forceinline__ void gc_encode_piece(
    const int bx, const int by,
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg,
    const int16_t *__restrict__ coefs, const int16_t *__restrict__ hist1,
    const int16_t *__restrict__ hist2, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, int16_t *__restrict__ seg_state,
    const int *__restrict__ first_open, const Ragged &rg)
{
    // Repair launch (first_open != nullptr, one workgroup row): channels with a seam that would not close are encoded
    // again, serially, from the earliest such seam among the workgroup's four channels to the end of the stream --
    // starting from seg_state of the piece before, which is the real history for all four (every seam before it
    // closed).  Workgroups without such a channel leave at once.
    constexpr bool repair = REPAIR;
    constexpr int CS = Lay<CPW>::CS, TF = Lay<CPW>::TF;
    using GcTile = GcTileT<CS, TF>;
    int64_t first_frame = seg.first(by);
    int repair_piece = 0;
    if (repair) {
        int k = 0x7f000000;
        for (int g = 0; g < CS; g++) {
            const int slot = bx * CS + g;
            if (slot < nch) {
                const int c = RAGGED ? rg.order[slot] : slot;
                k = first_open[c] < k ? first_open[c] : k;
            }
        }
        if (k <= 0 || k >= 0x7f000000) return;
        repair_piece = k;
        first_frame = seg.first(k);
    }
    // the workgroup's longest channel (slot 0 when ragged) decides whether the piece exists and how many tiles it has
    const int total_wg = RAGGED ? rg.length[rg.order[bx * CS]] : total_samples;
    if (first_frame * 14 >= total_wg) return;
    const int64_t piece_samples = repair ? (int64_t)0x7fffffff : (int64_t)seg.frames(by) * 14;
    auto piece_samples_of = [&](int total) {
        const int64_t rem = (int64_t)total - first_frame * 14;
        return (int)(rem < 0 ? 0 : (rem < piece_samples ? rem : piece_samples));
    };
    pcm += first_frame * 14;
    adpcm += first_frame * 8;
    __shared__ GcTile s_tile[2];
    // the winner's frame, unpacked: q[0..13], predictor, scale; the helper packs it (pack_frame) when it flushes
    __shared__ int4 s_out[2][CS][TF][4];
    __shared__ int s_ncold[SW];
    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const bool helper = wave == SW;
    const int lane = tid & 63;
    // encoder wave: CPW channels x 8 predictors (x 2 candidates when CPW = 4); helper wave: CS channel slots x TF frames
    constexpr int LPC = 64 / CPW;                          // lanes per channel: 16 or 8
    int grp = helper ? lane / TF : wave * CPW + lane / LPC;        // channel slot of this lane
    const int l16 = lane & (LPC - 1);
    const int hfr = lane % TF;                         // helper lanes: the frame inside the tile
    const int slot_raw = bx * CS + grp;
    const bool live = slot_raw < nch;
    const int slot = live ? slot_raw : nch - 1;
    const int ch = RAGGED ? rg.order[slot] : slot;
    const int16_t *src = pcm + (RAGGED ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    uint8_t *dst = adpcm + (RAGGED ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch);

    const int sample_count = piece_samples_of(RAGGED ? rg.length[ch] : total_samples);   // this slot's share of the piece
    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    const int frames = full_frames + (tail ? 1 : 0);
    const int frames_wg = RAGGED ? (piece_samples_of(total_wg) + 13) / 14 : frames;
    const int tiles = (frames_wg + TF - 1) / TF;

    if (helper) {
        // ---------------------------------------------------------------- helper wave
        __builtin_amdgcn_s_setprio(VGA_GC_HELPER_PRIO);
        uint32_t cpk[8];                               // (c1, c0) of predictor p as a packed pair: low half c1
#pragma unroll
        for (int p = 0; p < 8; p++)
            cpk[p] = (uint32_t)(uint16_t)coefs[ch * 16 + 2 * p + 1] | ((uint32_t)(uint16_t)coefs[ch * 16 + 2 * p] << 16);
        auto prepare = [&](int tile) {
            const int fr = RAGGED ? imax(imin(tile * TF + hfr, frames - 1), 0) : imin(tile * TF + hfr, frames - 1);
            int in[14];
            uint32_t w[7];                             // the frame as packed pairs (in[2i], in[2i+1])
            if (fr < full_frames) {
                const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + (int64_t)fr * 14);
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
#pragma unroll
                for (int i = 0; i < 7; i++) w[i] = (uint32_t)(in[2 * i] & 0xFFFF) | ((uint32_t)in[2 * i + 1] << 16);
            }
            GcTile &T = s_tile[tile & 1];
            int4 *xr = reinterpret_cast<int4 *>(&T.x[grp][hfr][0]);
            int4 *mr = reinterpret_cast<int4 *>(&T.in2048[grp][hfr][0]);
            xr[0] = make_int4(in[0], in[1], in[2], in[3]);
            xr[1] = make_int4(in[4], in[5], in[6], in[7]);
            xr[2] = make_int4(in[8], in[9], in[10], in[11]);
            xr[3] = make_int4(in[12], in[13], 0, 0);
            mr[0] = make_int4(in[0] * 2048, in[1] * 2048, in[2] * 2048, in[3] * 2048);
            mr[1] = make_int4(in[4] * 2048, in[5] * 2048, in[6] * 2048, in[7] * 2048);
            mr[2] = make_int4(in[8] * 2048, in[9] * 2048, in[10] * 2048, in[11] * 2048);
            mr[3] = make_int4(in[12] * 2048, in[13] * 2048, 0, 0);
            int4 *qr = reinterpret_cast<int4 *>(&T.in2048p[grp][hfr][0]);
            qr[0] = make_int4(in[0] * 2048 + 1024, in[1] * 2048 + 1024, in[2] * 2048 + 1024, in[3] * 2048 + 1024);
            qr[1] = make_int4(in[4] * 2048 + 1024, in[5] * 2048 + 1024, in[6] * 2048 + 1024, in[7] * 2048 + 1024);
            qr[2] = make_int4(in[8] * 2048 + 1024, in[9] * 2048 + 1024, in[10] * 2048 + 1024, in[11] * 2048 + 1024);
            qr[3] = make_int4(in[12] * 2048 + 1024, in[13] * 2048 + 1024, 0, 0);
            // pre-scan distances of samples 2..13 (:107-115): predicted = (in[k]*c1 + in[k+1]*c0) / 2048 for the
            // pair starting at k = s - 2.  One v_dot2c_i32_i16 per (predictor, sample): the pairs at even k are
            // the loaded dwords, the pairs at odd k one v_alignbit each (shared by the 8 predictors); the
            // wrap of int32 is the reference's (unchecked arithmetic).
            uint32_t pair[12];
#pragma unroll
            for (int k = 0; k < 12; k++)
                pair[k] = (k & 1) ? __builtin_amdgcn_alignbit(w[(k + 1) / 2], w[(k - 1) / 2], 16) : w[k / 2];
            uint32_t pre[8];
#pragma unroll
            for (int p = 0; p < 8; p++) {
                int dmax = 0, dmin = 0;
#ifdef VGA_GC_ABLATE_HELPER_PRESCAN                                  // timing only: what the helper's 96 distances per frame cost the launch
                dmax = in[2 + p] & 1023;
                dmin = -(in[3 + p] & 1023);
#else
#pragma unroll
                for (int k = 0; k < 12; k++) {
                    const int predicted = div2048(__builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, pair[k]),
                                                                         __builtin_bit_cast(short2v, cpk[p]), 0, false));
                    const int d = in[k + 2] - predicted;
                    dmax = imax(dmax, d);
                    dmin = imin(dmin, d);
                }
#endif
                pre[p] = (uint32_t)(clamp16i(dmax) & 0xFFFF) | ((uint32_t)clamp16i(dmin) << 16);
            }
            uint4 *pr = reinterpret_cast<uint4 *>(&T.pre[grp][hfr][0]);
            pr[0] = make_uint4(pre[0], pre[1], pre[2], pre[3]);
            pr[1] = make_uint4(pre[4], pre[5], pre[6], pre[7]);
        };
        auto flush = [&](int tile) {
            const int fr = tile * TF + hfr;
            if (!live || fr >= frames) return;
            const int4 *rec = &s_out[tile & 1][grp][hfr][0];
            const int4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
            const int q[14] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y};
            uint2 v;
            pack_frame(q, r3.z, r3.w, v.x, v.y);
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
    __builtin_amdgcn_s_setprio(VGA_GC_ENCODER_PRIO);   // its latency is the kernel's run time: win every issue arbitration
    int p = CPW == 4 ? l16 >> 1 : l16;
    const bool cand_b = CPW == 4 && (l16 & 1) != 0;
    const int c0 = coefs[ch * 16 + 2 * p];
    const int c1 = coefs[ch * 16 + 2 * p + 1];
    const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;   // predictor cannot wrap int32
    int h0 = hist2 ? hist2[ch] : 0;   // pcmBuffer[0] = History2 (GcAdpcmEncoder.cs:24)
    int h1 = hist1 ? hist1[ch] : 0;   // pcmBuffer[1] = History1 (:25)
    if (repair) {
        h0 = seg_state[((int64_t)(repair_piece - 1) * nch + ch) * 2];
        h1 = seg_state[((int64_t)(repair_piece - 1) * nch + ch) * 2 + 1];
    } else if (by > 0 && (!RAGGED || frames > 0)) {   // a later piece: the guess
        h0 = src[-2];
        h1 = src[-1];
    }
    VGA_OPAQUE(h0);
    VGA_OPAQUE(h1);

    if (lane == 0) s_ncold[wave] = 0;                  // cold blocks this piece (diagnostics, g_vga_gc_stats)
    struct Row { int x[16]; int m[14]; int mp[14]; uint32_t pre; };
    auto read_row = [&](const GcTile &T, int j, Row &R) {
        const int4 *xr = reinterpret_cast<const int4 *>(&T.x[grp][j][0]);
        const int4 *mr = reinterpret_cast<const int4 *>(&T.in2048[grp][j][0]);
        const int4 a0 = xr[0], a1 = xr[1], a2 = xr[2], a3 = xr[3];
        const int4 b0 = mr[0], b1 = mr[1], b2 = mr[2], b3 = mr[3];
        // keep the padding lanes "used": hipcc otherwise splits the 16-byte row reads into 7 odd-sized ones
        asm volatile("" ::"v"(a3.z), "v"(a3.w), "v"(b3.z), "v"(b3.w));
        int *x = R.x, *m = R.m;
        x[2] = a0.x; x[3] = a0.y; x[4] = a0.z; x[5] = a0.w; x[6] = a1.x; x[7] = a1.y; x[8] = a1.z; x[9] = a1.w;
        x[10] = a2.x; x[11] = a2.y; x[12] = a2.z; x[13] = a2.w; x[14] = a3.x; x[15] = a3.y;
        m[0] = b0.x; m[1] = b0.y; m[2] = b0.z; m[3] = b0.w; m[4] = b1.x; m[5] = b1.y; m[6] = b1.z; m[7] = b1.w;
        m[8] = b2.x; m[9] = b2.y; m[10] = b2.z; m[11] = b2.w; m[12] = b3.x; m[13] = b3.y;
        const int4 *qr = reinterpret_cast<const int4 *>(&T.in2048p[grp][j][0]);
        const int4 e0 = qr[0], e1 = qr[1], e2 = qr[2], e3 = qr[3];
        asm volatile("" ::"v"(e3.z), "v"(e3.w));
        int *mp = R.mp;
        mp[0] = e0.x; mp[1] = e0.y; mp[2] = e0.z; mp[3] = e0.w; mp[4] = e1.x; mp[5] = e1.y; mp[6] = e1.z; mp[7] = e1.w;
        mp[8] = e2.x; mp[9] = e2.y; mp[10] = e2.z; mp[11] = e2.w; mp[12] = e3.x; mp[13] = e3.y;
        R.pre = T.pre[grp][j][p];
    };
    auto pack = [](const int (&x)[16]) {
        X16 xs;
#pragma unroll
        for (int i = 0; i < 16; i++) xs.v[i] = x[i];
        return xs;
    };

    auto encode_frame = [&](Row &R, int buf, int j, bool upd) {
        int (&x)[16] = R.x;
        x[0] = h0;
        x[1] = h1;
        // ---- pre-scan (:107-124): two history-dependent distances + the helper's range for s = 2..13
        int s1;
        {
            const int d0 = x[2] - div2048(VGA_MUL24(x[0], c1) + VGA_MUL24(x[1], c0));
            const int d1 = x[3] - div2048(VGA_MUL24(x[1], c1) + VGA_MUL24(x[2], c0));
            const int dmax = imax(imax((int)(int16_t)(R.pre & 0xFFFF), d0), d1);
            const int dmin = imin(imin((int)R.pre >> 16, d0), d1);
            s1 = first_scale_power_from_range(dmax, dmin);
            if (__any(s1 == -100)) {                   // +M and -M both present: first occurrence decides
                if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential_cold(pack(x), c0, c1));
            }
        }
        // ---- first trip: candidate A at s1, B at s1+1 (speculation on the loop of :127-170)
        int final_sp = imin(s1 + (cand_b ? 1 : 0), 12);
        const bool at_cap = final_sp >= 12;            // the loop never goes past 12: this pass ends it
        const unsigned ov_limit = at_cap ? 3u : 248u;  // see `rare` below
        PassOut r = pass_fast_core(x, R.m, R.mp, c0, c1, final_sp);
        // Straight-line resolution, valid when no lane is `rare`:
        //   * no overflow can start the bump loop (:166-168 needs max_overflow + 8 > 256),
        //   * the 32-bit error sum of every lane that can become final is exact (gc_encode_core.hpp S3:
        //     final lanes have overflow <= 1, or <= 3 at the cap, so (2 ov + 1) << (k - 11) <= 34996),
        //   * (total << 4) of the final lanes fits the 32-bit argmin key (`wide` otherwise; lanes that
        //     overflowed are not final and their error sums -- often huge -- are never looked at).
        // eff = overflow as the loop condition sees it (a pass at the cap ends the loop whatever it overflowed).
        // `rare`: the bump loop can start, or the coefficients can wrap -- the pair's A lane redoes the reference's loop
        // as written, for the whole wave (hostile input).  A pass at the cap that overflowed by more than 3 (loud noise,
        // clipped waves) is right except for its 32-bit error sum: `inexact`, the same pass again with a 64-bit sum.
        const bool rare = !coef_ok || (!at_cap && (unsigned)r.max_overflow > ov_limit);
        const int eff = at_cap ? 0 : r.max_overflow;
        const int eff_other = dpp<DPP_QUAD_XOR1>(eff);
        // A is final iff its pass did not overflow by more than 1; B is final iff A is not and B did not.
        bool fin = eff < 2 && (eff_other | (cand_b ? 0 : 2)) >= 2;
        const bool resume = !cand_b && imin(eff, eff_other) >= 2;     // both overflowed: A carries on at s1+2
        const bool inexact = fin && at_cap && (unsigned)r.max_overflow > ov_limit;
        // ---- the frame's tail: argmin over the 8 predictors (first index wins ties, :66-76), winner's history
        // broadcast, winner's record to LDS.  A lambda so that the hot and the cold branch each get their own
        // copy: merging the two branches' PassOut registers instead put the copies on the hot path.
        // 32-bit keys: error sums from 2^28 on share one key (such a predictor loses to any below -- as a rule it is one of
        // several wild ones); only when a channel's BEST is that large (`sat`, wave-uniform) do the 64-bit keys decide -- in
        // the cold block: the hot copy of the frame's tail holds no 64-bit code.
        constexpr unsigned SAT = (1u << 28) - 1;
        auto argmin32 = [&](const PassOut &r, bool fin, bool &sat) __attribute__((always_inline)) -> int {
            const unsigned tot = umin32((unsigned)(r.total >> 32) ? SAT : (unsigned)r.total, SAT);
            const unsigned key = fin ? ((tot << 4) | (unsigned)l16) : 0xFFFFFFFFu;
            const unsigned best = row16_reduce(key, [](unsigned a, unsigned b) { return a < b ? a : b; });
            sat = __any((best >> 4) >= SAT);
            return (int)(best & 15u);
        };
        auto argmin64 = [&](const PassOut &r, bool fin) __attribute__((always_inline)) -> int {
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
            return (int)(key & 15u);
        };
        // ---- the frame's tail: winner's history broadcast, winner's record to LDS.  A lambda so that the hot and the cold
        // branch each get their own copy: merging the two branches' PassOut registers instead put the copies on the hot path.
        auto commit = [&](const PassOut &r, int final_sp, int winner) __attribute__((always_inline)) {
            const bool won = l16 == winner;
            const unsigned pay = row16_reduce(won ? r.hist_pair : 0u,
                                              [](unsigned a, unsigned b) { return a | b; });
            if (won) {                                           // packed and flushed by the helper, a tile at a time
                int4 *rec = &s_out[buf][grp][j][0];
                rec[0] = make_int4(r.q[0], r.q[1], r.q[2], r.q[3]);
                rec[1] = make_int4(r.q[4], r.q[5], r.q[6], r.q[7]);
                rec[2] = make_int4(r.q[8], r.q[9], r.q[10], r.q[11]);
                rec[3] = make_int4(r.q[12], r.q[13], p, final_sp);
            }
            if (!RAGGED || upd) {
                h0 = (int)(int16_t)(pay & 0xFFFF);   // pcmBuffer[0] = pcmBuffer[14] (:40)
                h1 = (int)pay >> 16;                 // pcmBuffer[1] = pcmBuffer[15] (:41)
            }
            VGA_OPAQUE(h0);                      // hide the 16-bit range: keeps the 24-bit multiplies the next frame
            VGA_OPAQUE(h1);                      // asks for (the compiler otherwise widens them to 64-bit mads)
        };
        bool sat = false;
        const int winner32 = argmin32(r, fin && !inexact, sat);      // (a lane whose sum is not to be trusted yet is not in it)
        if (__builtin_expect(__any(rare || resume || inexact) || sat, 0)) {
            // ---- cold block (third trips: a third of the wave-frames on the synthetic set, LABNOTES 8.4)
#ifdef VGA_GC_STATS                                                  // (a -DVGA_GC_STATS build only: five instructions per cold block = 1 ms of the launch)
            if (lane == 0) atomicAdd(&s_ncold[wave], 1);
#endif
            ColdState st;
#pragma unroll
            for (int i = 0; i < 16; i++) st.x[i] = x[i];
#pragma unroll
            for (int i = 0; i < 14; i++) { st.m[i] = R.m[i]; st.mp[i] = R.mp[i]; }
            st.c0 = c0; st.c1 = c1; st.s1 = s1;
            const bool redo = __any(rare);             // (this layout: every pair's A lane walks the whole loop again, B lanes are out)
            st.generic = redo && !cand_b; st.start = s1 - 1; st.drop = redo && cand_b;
            st.wide = !redo && inexact; st.resume = !redo && resume;
            st.bump_a = 0; st.ov_a = 0; st.ov_b = 0;
            const ColdOut o = encode_frame_cold(st, r, final_sp, fin);
            bool sat2 = false;
            int winner = argmin32(o.r, o.fin != 0, sat2);
            if (sat2) winner = argmin64(o.r, o.fin != 0);
            commit(o.r, o.final_sp, winner);
        } else
            commit(r, final_sp, winner32);
    };

    // ---- CPW = 8: lane = (channel, predictor); candidate B (s1 + 1) and candidate A (s1) are two passes of the SAME lane,
    // inlined back to back (two independent dependent chains: the wave has instructions to issue while one waits).
    auto encode_frame8 = [&](Row &R, int buf, int j, bool upd) {
        int (&x)[16] = R.x;
        x[0] = h0;
        x[1] = h1;
        int s1;
        {
            const int d0 = x[2] - div2048(VGA_MUL24(x[0], c1) + VGA_MUL24(x[1], c0));
            const int d1 = x[3] - div2048(VGA_MUL24(x[1], c1) + VGA_MUL24(x[2], c0));
            const int dmax = imax(imax((int)(int16_t)(R.pre & 0xFFFF), d0), d1);
            const int dmin = imin(imin((int)R.pre >> 16, d0), d1);
            s1 = first_scale_power_from_range(dmax, dmin);
            if (__any(s1 == -100)) {                   // +M and -M both present: first occurrence decides
                if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential_cold(pack(x), c0, c1));
            }
        }
        const int sp_a = imin(s1, 12), sp_b = imin(s1 + 1, 12);
        // Round 5: when every lane of the wave quantises at scale 9 or below (70 % of the synthetic set's wave-frames) the two
        // passes run without the detour through f32 -- (int)(float)d is d below 2^24, two conversions a sample less -- and each
        // lane checks from its overflow that no distance reached 2^24 (gc_encode_core.hpp: NO_ROUND); a lane that cannot
        // tell sends the wave through the passes as they always were.
        PassOut ra, rb;
#ifdef VGA_GC_NO_FAST_PASSES                                        // (timing-only switch, tools/build_variants.sh)
        bool short_passes = false;
#else
        bool short_passes = !__any(sp_b > 9);
#endif
        if (short_passes) {
            rb = pass_fast_core_no_round(x, R.m, R.mp, c0, c1, sp_b);
            ra = pass_fast_core_no_round(x, R.m, R.mp, c0, c1, sp_a);
            // (hostile coefficients: the lane walks the reference's loop as written whatever these passes say)
            const bool trusted = !coef_ok || (pass_no_round_is_exact(sp_a, ra.max_overflow) && pass_no_round_is_exact(sp_b, rb.max_overflow));
            short_passes = !__any(!trusted);
        }
        if (!short_passes) {
            rb = pass_fast_core(x, R.m, R.mp, c0, c1, sp_b);
            ra = pass_fast_core(x, R.m, R.mp, c0, c1, sp_a);
        }
        const bool cap_a = sp_a >= 12, cap_b = sp_b >= 12;         // a pass at the cap ends the loop whatever it overflowed
        const int eff_a = cap_a ? 0 : ra.max_overflow, eff_b = cap_b ? 0 : rb.max_overflow;
        const bool fin_a = eff_a < 2;                              // the reference stops after the pass at s1
        // Which of the two passes the reference ends on, and what can stand in the way (each lane for itself; round 5 --
        // until then any of these in any lane redid the whole wave's frame from scratch):
        //   * bump_a / bump_b: the pass the loop has reached overflowed by more than 248, the bump loop (:166-168) moves the
        //     scale by more than one step -- the reference's loop as written, from the scale it moves to;
        //   * hostile coefficients (they can wrap int32): the whole loop as written;
        //   * inexact: the final pass ran at the cap and overflowed by more than 3 -- its 32-bit error sum may have wrapped
        //     (gc_encode_core.hpp S3); the pass again with a 64-bit sum.  (B's overflow is nobody's business when A is final.)
        const bool bump_a = !cap_a && (unsigned)ra.max_overflow > 248u;
        const bool bump_b = !fin_a && !cap_b && (unsigned)rb.max_overflow > 248u;
        const bool generic = !coef_ok || bump_a || bump_b;
        const bool inexact = !generic && (fin_a ? (cap_a && (unsigned)ra.max_overflow > 3u) : (cap_b && (unsigned)rb.max_overflow > 3u));
#ifdef VGA_GC_ABLATE_THIRD_TRIPS                                     // timing only (tools/build_variants.sh): what the cold block costs
        const bool resume = false;
#else
        const bool resume = !generic && !fin_a && eff_b >= 2;      // both overflowed: on to s1 + 2 in the cold block
#endif
        // (Round 6, measured and taken out again: the winning lanes storing the right pass's nibbles under their own masks
        // instead of fourteen selects in every lane -- 18 VALU instructions a frame less, two more exec-masked branches on
        // the wave's critical path: 148.5 ms against 146.0, profiles/r06_e_encode_variants.log.)
        PassOut r;
#pragma unroll
        for (int i = 0; i < 14; i++) r.q[i] = fin_a ? ra.q[i] : rb.q[i];
        r.total = fin_a ? ra.total : rb.total;
        r.hist_pair = fin_a ? ra.hist_pair : rb.hist_pair;
        r.max_overflow = fin_a ? ra.max_overflow : rb.max_overflow;
        r.o12 = r.o13 = 0;
        r.exact = true;
        const int final_sp = fin_a ? sp_a : sp_b;
        // 32-bit keys: error sums from 2^28 on share one key (such a predictor loses to any below); only when a channel's BEST
        // is that large (`sat`, wave-uniform) do the 64-bit keys decide -- in the cold block
        constexpr unsigned SAT = (1u << 28) - 1;
        auto argmin32 = [&](const PassOut &r, bool fin, bool &sat) __attribute__((always_inline)) -> int {
            const unsigned tot = umin32((unsigned)(r.total >> 32) ? SAT : (unsigned)r.total, SAT);
            unsigned key = fin ? ((tot << 3) | (unsigned)p) : 0xFFFFFFFFu;
            key = umin32(key, (unsigned)dpp<DPP_QUAD_XOR1>((int)key));
            key = umin32(key, (unsigned)dpp<DPP_QUAD_XOR2>((int)key));
            key = umin32(key, (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)key));
            sat = __any((key >> 3) >= SAT);
            return (int)(key & 7u);
        };
        auto argmin64 = [&](const PassOut &r, bool fin) __attribute__((always_inline)) -> int {
            uint64_t key = fin ? ((r.total << 3) | (uint64_t)p) : ~0ull;
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
#undef VGA_MIN64_STAGE
            return (int)(key & 7u);
        };
        auto commit = [&](const PassOut &r, int final_sp, int winner) __attribute__((always_inline)) {
            const bool won = p == winner;
            unsigned pay = won ? r.hist_pair : 0u;
            pay |= (unsigned)dpp<DPP_QUAD_XOR1>((int)pay);
            pay |= (unsigned)dpp<DPP_QUAD_XOR2>((int)pay);
            pay |= (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)pay);
            if (won) {
                int4 *rec = &s_out[buf][grp][j][0];
                rec[0] = make_int4(r.q[0], r.q[1], r.q[2], r.q[3]);
                rec[1] = make_int4(r.q[4], r.q[5], r.q[6], r.q[7]);
                rec[2] = make_int4(r.q[8], r.q[9], r.q[10], r.q[11]);
                rec[3] = make_int4(r.q[12], r.q[13], p, final_sp);
            }
            if (!RAGGED || upd) {                // (a slot past its last frame keeps the history it ended on)
                h0 = (int)(int16_t)(pay & 0xFFFF);
                h1 = (int)pay >> 16;
            }
            VGA_OPAQUE(h0);
            VGA_OPAQUE(h1);
        };
        bool sat = false;
        const int winner32 = argmin32(r, !(generic || resume || inexact), sat);
        if (__builtin_expect(__any(generic || resume || inexact) || sat, 0)) {
#ifdef VGA_GC_STATS                                                  // (a -DVGA_GC_STATS build only: five instructions per cold block = 1 ms of the launch)
            if (lane == 0) atomicAdd(&s_ncold[wave], 1);
#endif
            ColdState st;
#pragma unroll
            for (int i = 0; i < 16; i++) st.x[i] = x[i];
#pragma unroll
            for (int i = 0; i < 14; i++) { st.m[i] = R.m[i]; st.mp[i] = R.mp[i]; }
            st.c0 = c0; st.c1 = c1; st.s1 = s1;
            st.generic = generic; st.drop = 0; st.wide = inexact; st.resume = resume;
            // where the reference's loop stands (the value of scalePower before its next ++): at its start for hostile
            // coefficients; behind the bumps of the pass that started them otherwise (encode_frame_cold works that out)
#ifdef VGA_GC_R05_COLD                                                // (timing only: the cold block's entry as it was in round 5)
            st.start = !coef_ok ? s1 - 1 : (bump_a ? apply_bumps(s1, ra.max_overflow) : apply_bumps(s1 + 1, rb.max_overflow));
#else
            st.start = !coef_ok ? s1 - 1 : -100;
#endif
            st.bump_a = bump_a; st.ov_a = ra.max_overflow; st.ov_b = rb.max_overflow;
            const ColdOut o = encode_frame_cold(st, r, final_sp, (generic || resume) ? 0 : 1);
            bool sat2 = false;
            int winner = argmin32(o.r, o.fin != 0, sat2);
            if (sat2) winner = argmin64(o.r, o.fin != 0);
            commit(o.r, o.final_sp, winner);
        } else
            commit(r, final_sp, winner32);
    };
    auto encode_one = [&](Row &R, int buf, int j, bool upd) __attribute__((always_inline)) {
        if constexpr (CPW == 4) encode_frame(R, buf, j, upd);
        else encode_frame8(R, buf, j, upd);
    };

#ifdef VGA_DEBUG_TIMESTAMPS
    if (!repair && tid == 0 && gridDim.y != 1) g_vga_enc_ts[2 * ((blockIdx.y * gridDim.x + blockIdx.x) & 32767)] = wall_clock64();
#endif
    __syncthreads();                                   // tile 0 prepared
    for (int tile = 0; tile < tiles; tile++) {
        const int buf = tile & 1;
        const int nf = imin(TF, frames_wg - tile * TF);
        const int left = frames - tile * TF;           // RAGGED: this slot's own frames in the tile
#ifndef VGA_GC_NO_TILE_REDERIVE                        // (timing-only switch, tools/build_variants.sh)
        {
            // the lane's LDS addresses follow from its slot and predictor: derived again for every tile (three VALU ops) --
            // kept across the piece they were spilled and came back through scratch loads the tile had to wait for
            int l = lane;
            asm volatile("" : "+v"(l));
            grp = wave * CPW + l / LPC;
            p = CPW == 4 ? (l & (LPC - 1)) >> 1 : (l & (LPC - 1));
        }
#endif
        const GcTile &T = s_tile[buf];
        // two row register sets, ping-pong: the LDS reads of frame j+1 are in flight during frame j
        Row RA, RB;
        read_row(T, 0, RA);
#pragma unroll 1
        for (int j = 0; j < nf; j += 2) {
            read_row(T, imin(j + 1, TF - 1), RB);
            encode_one(RA, buf, j, j < left);
            if (j + 1 < nf) {
                read_row(T, imin(j + 2, TF - 1), RA);
                encode_one(RB, buf, j + 1, j + 1 < left);
            }
        }
        __syncthreads();                               // tile done: helper may flush it and refill this buffer later
    }
    if (seg_state && !repair && live && l16 == 0) {   // one lane per channel
        int16_t *st = seg_state + ((int64_t)by * nch + ch) * 2;
        st[0] = (int16_t)h0;
        st[1] = (int16_t)h1;
    }
    if (lane == 0) {
        atomicAdd(&g_vga_gc_stats[3], (unsigned long long)frames_wg);
        atomicAdd(&g_vga_gc_stats[4], (unsigned long long)s_ncold[wave]);
        if (wave == 0) atomicAdd(&g_vga_gc_stats[6], 1ull);
    }
#ifdef VGA_DEBUG_TIMESTAMPS
    if (!repair && tid == 0 && gridDim.y != 1) g_vga_enc_ts[2 * ((blockIdx.y * gridDim.x + blockIdx.x) & 32767) + 1] = wall_clock64();
#endif
}


template <bool REPAIR, int CPW, bool RAGGED>
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void gc_encode_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg,
    const int16_t *__restrict__ coefs, const int16_t *__restrict__ hist1,
    const int16_t *__restrict__ hist2, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, int16_t *__restrict__ seg_state,
    const int *__restrict__ first_open, const Ragged rg)
{
    gc_encode_piece<REPAIR, CPW, RAGGED>(blockIdx.x, blockIdx.y, pcm, pcm_pitch, nch, total_samples, seg, coefs, hist1, hist2, adpcm,
                                         adpcm_pitch, seg_state, first_open, rg);
}

// ---------------------------------------------------------------- seams of the time segments
__device__ __forceinline__ void store_frame_bytes(uint8_t *dst, uint32_t d0, uint32_t d1, int nbytes)
{
    const uint64_t bits = ((uint64_t)d1 << 32) | d0;
    for (int b = 0; b < nbytes; b++) dst[b] = (uint8_t)(bits >> (8 * b));
}

// Eight lanes per (channel, seam) -- one per predictor, as DspEncodeFrame's loop (:58-77).  seam_run: from the TRUE history
// (h0, h1) at the start of piece k encode again frame by frame, next to a replay of the reconstruction of the run whose
// bytes the piece holds (decoding its frames from ITS start history (g2, g1), before they are overwritten), until both
// histories coincide at a frame end: from there on the piece holds what the serial encoder writes.  Returns with
// open == true when the piece ended first; (h0, h1) is then the true history at the piece's end.
// LPC (round 6): lanes per channel.  8: a lane per predictor, the two speculative passes of a frame one after the other in the
// same lane (two chains for the instruction stream to interleave: the layout of the seams inside the persistent kernel,
// where every SIMD holds other waves).  16: a lane per (predictor, candidate) -- pass A (scale s1) in the even lane, pass B
// (s1 + 1) in the odd one, as the encoder's CPW = 4 layout -- half the instructions per frame on the frame's critical
// path: for the seam and chain launches of batches below 512 channels, whose run time IS their slowest seam's.
template <int LPC>
__device__ __forceinline__ void seam_run(const int16_t *__restrict__ src, uint8_t *__restrict__ dst, int c0, int c1, bool coef_ok,
                                         int pr, int ch, int k, int64_t f0, int total_samples, int piece_frames, int max_frames, int force_open,
                                         int &h0, int &h1, int g2, int g1, bool &open, int &frames_run)
{
    static_assert(LPC == 8 || LPC == 16, "a lane per predictor, or per (predictor, candidate)");
    const bool cand_b = LPC == 16 && (threadIdx.x & 1) != 0;
    const int full_frames = total_samples / 14;
    // The frame loop is one wave's dependent chain (nothing else runs on its SIMD for long: the slowest seam IS the
    // kernel's run time), so it is the encoder's fast path -- range pre-scan, the two speculative passes of the
    // (channel, predictor) layout, DPP argmin -- with the next frame's PCM and old bytes already in flight; the
    // reference's loop as written (resume_passes) only behind the same `rare` / `resume` conditions as there.
    const int64_t f_end = imin((int)imin((int)(f0 + piece_frames), (int)(f0 + max_frames)), full_frames);   // frames [f0, f_end)
    const int64_t f_last = full_frames > 0 ? full_frames - 1 : 0;
    // (Round 6, measured and taken out again: the frames a block of LPC at a time -- lane j of the group loading frame fb + j,
    // the next block in flight, a frame's nine dwords handed round through the crossbar.  Nothing gained, 5 % lost at 512
    // channels: the loop does not wait for its loads -- a frame of a seam run is ~1000 instructions of a lone wave at
    // 2.3 ns each (replay 200, pre-scan 160, passes 220-450, argmin / pack / store 120), profiles/r06_m_channel_scaling.log.)
    auto fetch = [&](int64_t f, uint32_t (&w)[7], uint2 &old) {
        const int64_t fc = f < f_last ? f : f_last;     // clamped: always a valid full frame (unconditional loads)
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + fc * 14);
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = p32[i];
        old = *reinterpret_cast<const uint2 *>(dst + fc * 8);
    };
    uint32_t w[7], wn[7];
    uint2 old, oldn;
    fetch(f0, w, old);
    for (int64_t f = f0; ; f++) {
        const bool in_range = f < f_end;
        if (!__any(open && in_range)) break;
        const bool act = open && in_range;
        frames_run += act ? 1 : 0;
        fetch(f + 1, wn, oldn);                         // in flight during this frame
        int x[16], m[14], mp[14];
        x[0] = h0;
        x[1] = h1;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            x[2 + 2 * i] = (int)(int16_t)(w[i] & 0xFFFF);
            x[3 + 2 * i] = (int)w[i] >> 16;
        }
#pragma unroll
        for (int i = 0; i < 14; i++) {
            m[i] = x[2 + i] * 2048;
            mp[i] = m[i] + 1024;
        }
        // the guessed run's reconstruction of this frame (GcAdpcmDecoder.cs:25-45), from its bytes before they change
        {
            const int ps = (int)(old.x & 0xFFu);
            const int scale = (1 << (ps & 0xF)) * 2048;
            const int pred = (ps >> 4) & 7;
            // that predictor's pair sits in lane `pred` (LPC = 16: lanes 2 pred, 2 pred + 1) of the group
            const int k1 = __shfl(c0, LPC == 16 ? 2 * pred : pred, LPC), k2 = __shfl(c1, LPC == 16 ? 2 * pred : pred, LPC);
            const uint64_t bits = ((uint64_t)old.y << 32) | old.x;            // byte b of the frame = bits >> 8b
#pragma unroll
            for (int s2 = 0; s2 < 14; s2++) {
                const int byte = (int)((bits >> (8 * (1 + (s2 >> 1)))) & 0xFFu);
                const int nib = (s2 & 1) ? (byte & 0xF) : (byte >> 4);
                const int v = clamp16i((k1 * g1 + k2 * g2 + scale * ((nib ^ 8) - 8) + 1024) >> 11);
                g2 = g1;
                g1 = v;
            }
        }
        // the true frame: pre-scan (:107-124), candidates A and B (:127-170), see encode_frame8
        int s1;
        {
            int dmax = 0, dmin = 0;
            prescan_range(x, c0, c1, 0, 14, dmax, dmin);
            s1 = first_scale_power_from_range(dmax, dmin);
            if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
        }
        const int sp_a = imin(s1, 12), sp_b = imin(s1 + 1, 12);
        PassOut r;
        int final_sp;
        bool fin = true;                               // LPC = 16: this lane's pass is the one the reference ends on
        if (LPC == 8) {
        // (the passes as they always were: without the f32 detour -- tried in round 5 -- a seam run is no faster, LABNOTES 9.7)
        const PassOut rb = pass_fast_core(x, m, mp, c0, c1, sp_b);
        const PassOut ra = pass_fast_core(x, m, mp, c0, c1, sp_a);
        const bool cap_a = sp_a >= 12, cap_b = sp_b >= 12;
        const bool rare = !coef_ok || (unsigned)ra.max_overflow > (cap_a ? 3u : 248u) || (unsigned)rb.max_overflow > (cap_b ? 3u : 248u);
        const int eff_a = cap_a ? 0 : ra.max_overflow, eff_b = cap_b ? 0 : rb.max_overflow;
        const bool fin_a = eff_a < 2;
        const bool resume = !fin_a && eff_b >= 2;
#pragma unroll
        for (int i = 0; i < 14; i++) r.q[i] = fin_a ? ra.q[i] : rb.q[i];
        r.total = fin_a ? ra.total : rb.total;
        r.hist_pair = fin_a ? ra.hist_pair : rb.hist_pair;
        final_sp = fin_a ? sp_a : sp_b;
        if (__any(rare || resume)) {
            if (rare || resume) {
                const PassOut rc = resume_passes(x, c0, c1, rare ? s1 - 1 : s1 + 1, final_sp);
#pragma unroll
                for (int i = 0; i < 14; i++) r.q[i] = rc.q[i];
                r.total = rc.total;
                r.hist_pair = rc.hist_pair;
            }
        }
        } else {
            // one pass per lane: A at s1 (even lane), B at s1 + 1 (odd lane); the pair decides from the two overflows which of
            // them the reference's loop ends on (encode_frame, CPW = 4); anything else -- both overflowed, a bump, an inexact
            // sum, hostile coefficients -- and the A lane walks the reference's loop as written, the B lane is out
            final_sp = cand_b ? sp_b : sp_a;
            r = pass_fast_core(x, m, mp, c0, c1, final_sp);
            const bool cap = final_sp >= 12;
            const bool rare_mine = !coef_ok || (unsigned)r.max_overflow > (cap ? 3u : 248u);
            const int eff = cap ? 0 : r.max_overflow;
            const int eff_other = dpp<DPP_QUAD_XOR1>(eff);
            // (the exchange on its own line: behind `rare_mine ||` it would run only in the lanes whose own flag is clear, and
            // read their partners -- the lanes it is there to hear from -- as inactive)
            const int rare_other = dpp<DPP_QUAD_XOR1>(rare_mine ? 1 : 0);
            const bool rare = rare_mine || rare_other != 0;
            const int eff_a = cand_b ? eff_other : eff, eff_b = cand_b ? eff : eff_other;
            const bool fin_a = eff_a < 2;
            const bool resume = !fin_a && eff_b >= 2;
            fin = cand_b ? !fin_a : fin_a;
            if (__any(rare || resume)) {
                if (rare || resume) {
                    fin = !cand_b;
                    if (!cand_b) r = resume_passes(x, c0, c1, rare ? s1 - 1 : s1 + 1, final_sp);
                }
            }
        }
        // totals are below 2^60 (14 squares of 17-bit errors): the predictor (LPC = 16: the lane of the group, predictor-major
        // -- one lane of a pair at most is in it) rides in the low bits
        const int lidx = LPC == 16 ? (int)(threadIdx.x & 15) : pr;
        uint64_t key = (act && fin) ? ((r.total << 4) | (uint64_t)lidx) : ~0ull;
#define VGA_MIN64_STAGE(CTRL)                                                          \
        {                                                                              \
            const unsigned olo = (unsigned)dpp<CTRL>((int)(uint32_t)key);              \
            const unsigned ohi = (unsigned)dpp<CTRL>((int)(uint32_t)(key >> 32));      \
            const uint64_t okey = ((uint64_t)ohi << 32) | olo;                         \
            key = okey < key ? okey : key;                                             \
        }
        VGA_MIN64_STAGE(DPP_QUAD_XOR1)
        VGA_MIN64_STAGE(DPP_QUAD_XOR2)
        VGA_MIN64_STAGE(DPP_ROW_HALF_MIRROR)
        if (LPC == 16) VGA_MIN64_STAGE(DPP_ROW_MIRROR)
#undef VGA_MIN64_STAGE
        const bool won = act && fin && lidx == (int)(key & 15u);
        unsigned pay = won ? r.hist_pair : 0u;
        pay |= (unsigned)dpp<DPP_QUAD_XOR1>((int)pay);
        pay |= (unsigned)dpp<DPP_QUAD_XOR2>((int)pay);
        pay |= (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)pay);
        if (LPC == 16) pay |= (unsigned)dpp<DPP_ROW_MIRROR>((int)pay);
        if (won) {
            uint32_t d0, d1;
            pack_frame(r.q, pr, final_sp, d0, d1);
            *reinterpret_cast<uint2 *>(dst + f * 8) = make_uint2(d0, d1);
        }
        if (act) {
            h0 = (int)(int16_t)(pay & 0xFFFF);          // :40-41
            h1 = (int)pay >> 16;
            if (h0 == g2 && h1 == g1 && !seam_forced_open(force_open, ch, k)) open = false;   // closed
        }
#pragma unroll
        for (int i = 0; i < 7; i++) w[i] = wn[i];
        old = oldn;
    }
}

// The zero-padded partial last frame of a stream (GcAdpcmEncoder.cs:32-38) from the true history (h0, h1), by the eight
// lanes of a channel exactly as seam_run encodes a frame; SampleCountToByteCount(tail) bytes are stored.  For the chain
// kernel: a run that is still apart at the very end of the stream has only this frame left to correct.
template <int LPC>
__device__ __forceinline__ void encode_tail_frame(const int16_t *__restrict__ src, uint8_t *__restrict__ dst, int c0, int c1,
                                                  bool coef_ok, int pr, int total_samples, int h0, int h1, bool act)
{
    const int full_frames = total_samples / 14, tail = total_samples - full_frames * 14;
    int x[16], m[14], mp[14];
    x[0] = h0;
    x[1] = h1;
#pragma unroll
    for (int i = 0; i < 14; i++) x[2 + i] = i < tail ? (int)src[(int64_t)full_frames * 14 + i] : 0;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        m[i] = x[2 + i] * 2048;
        mp[i] = m[i] + 1024;
    }
    int s1;
    {
        int dmax = 0, dmin = 0;
        prescan_range(x, c0, c1, 0, 14, dmax, dmin);
        s1 = first_scale_power_from_range(dmax, dmin);
        if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
    }
    const int sp_a = imin(s1, 12), sp_b = imin(s1 + 1, 12);
    const PassOut rb = pass_fast_core(x, m, mp, c0, c1, sp_b);
    const PassOut ra = pass_fast_core(x, m, mp, c0, c1, sp_a);
    const bool cap_a = sp_a >= 12, cap_b = sp_b >= 12;
    const bool rare = !coef_ok || (unsigned)ra.max_overflow > (cap_a ? 3u : 248u) || (unsigned)rb.max_overflow > (cap_b ? 3u : 248u);
    const int eff_a = cap_a ? 0 : ra.max_overflow, eff_b = cap_b ? 0 : rb.max_overflow;
    const bool fin_a = eff_a < 2;
    const bool resume = !fin_a && eff_b >= 2;
    PassOut r;
#pragma unroll
    for (int i = 0; i < 14; i++) r.q[i] = fin_a ? ra.q[i] : rb.q[i];
    r.total = fin_a ? ra.total : rb.total;
    int final_sp = fin_a ? sp_a : sp_b;
    if (rare || resume) {
        const PassOut rc = resume_passes(x, c0, c1, rare ? s1 - 1 : s1 + 1, final_sp);
#pragma unroll
        for (int i = 0; i < 14; i++) r.q[i] = rc.q[i];
        r.total = rc.total;
    }
    // (LPC = 16: both lanes of a predictor hold the same frame -- the even one is in the argmin and stores)
    const bool in_it = act && (LPC == 8 || (threadIdx.x & 1) == 0);
    uint64_t key = in_it ? ((r.total << 3) | (uint64_t)pr) : ~0ull;
#define VGA_MIN64_STAGE(CTRL)                                                          \
    {                                                                                  \
        const unsigned olo = (unsigned)dpp<CTRL>((int)(uint32_t)key);                  \
        const unsigned ohi = (unsigned)dpp<CTRL>((int)(uint32_t)(key >> 32));          \
        const uint64_t okey = ((uint64_t)ohi << 32) | olo;                             \
        key = okey < key ? okey : key;                                                 \
    }
    VGA_MIN64_STAGE(DPP_QUAD_XOR1)
    VGA_MIN64_STAGE(DPP_QUAD_XOR2)
    VGA_MIN64_STAGE(DPP_ROW_HALF_MIRROR)
    if (LPC == 16) VGA_MIN64_STAGE(DPP_ROW_MIRROR)
#undef VGA_MIN64_STAGE
    if (in_it && pr == (int)(key & 7u)) {
        uint32_t d0, d1;
        pack_frame(r.q, pr, final_sp, d0, d1);
        const int nibbles = tail + 2;                                   // SampleCountToNibbleCount of the tail (GcAdpcmMath.cs:24-30)
        store_frame_bytes(dst + (int64_t)full_frames * 8, d0, d1, nibbles / 2 + (nibbles & 1));
    }
}

// All seams at once, each from the history the piece before ended on (seg_state: the real one provided THAT piece's own
// seam closes).  A seam still open at the end of its piece leaves a flag and the true history it arrived at
// (seam_flag / seam_end) and its index in first_open[channel]: gc_encode_chain_kernel carries on from there.
template <int LPC>
__device__ __forceinline__ void seam_piece(
    const int slot_raw, const int k, const int lane,
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg,
    const int16_t *__restrict__ coefs, uint8_t *adpcm, int64_t adpcm_pitch,
    const int16_t *seg_state, int *first_open, int *seam_flag, int *seam_end, int force_open, const Ragged &rg)
{
    const int pr = LPC == 16 ? (lane >> 1) & 7 : lane & 7;        // this lane's predictor
    const bool lead = (lane & (LPC - 1)) == 0;          // one lane per channel: flags, statistics
    const int64_t f0 = seg.first(k);
    const int slot = slot_raw < nch ? slot_raw : nch - 1;
    const int ch = rg.order ? rg.order[slot] : slot;
    if (rg.order) total_samples = rg.length[ch];        // ragged: the seam exists only where the channel reaches piece k
    const bool valid = slot_raw < nch && f0 * 14 < total_samples;
    const int16_t *src = pcm + (rg.order ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    uint8_t *dst = adpcm + (rg.order ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch);
    const int16_t *cf = coefs + ch * 16;
    const int c0 = cf[2 * pr], c1 = cf[2 * pr + 1];
    const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;
    int h0 = valid ? seg_state[((int64_t)(k - 1) * nch + ch) * 2] : 0;            // true history (x[0], x[1])
    int h1 = valid ? seg_state[((int64_t)(k - 1) * nch + ch) * 2 + 1] : 0;
    const int g2 = valid ? src[f0 * 14 - 2] : 0, g1 = valid ? src[f0 * 14 - 1] : 0;   // the guessed run's history (g1 = newest)
    bool open = valid;                                  // uniform inside a group of eight
    int frames_run = 0;
    seam_run<LPC>(src, dst, c0, c1, coef_ok, pr, ch, k, f0, total_samples, seg.frames(k), seg.frames(k), force_open, h0, h1, g2, g1, open, frames_run);
    // still open at the piece's end (half the seams close within nine frames, one in a hundred needs more than 400, a
    // few channels never meet)
    if (valid && lead) {
        atomicAdd(&g_vga_gc_stats[open ? 1 : 0], 1ull);
        atomicAdd(&g_vga_gc_stats[2], (unsigned long long)frames_run);
        seam_flag[(int64_t)(k - 1) * nch + ch] = open ? 1 : 0;
        if (open) {
            seam_end[(int64_t)(k - 1) * nch + ch] = (int)(((unsigned)h1 << 16) | ((unsigned)h0 & 0xFFFFu));
            atomicMin(&first_open[ch], k);
        }
    }
}

template <int LPC>
__global__ __launch_bounds__(64) void gc_encode_seam_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg,
    const int16_t *__restrict__ coefs, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch,
    const int16_t *__restrict__ seg_state, int *__restrict__ first_open, int *__restrict__ seam_flag,
    int *__restrict__ seam_end, int force_open, const Ragged rg)
{
    seam_piece<LPC>(blockIdx.x * (64 / LPC) + (threadIdx.x / LPC), blockIdx.y + 1, threadIdx.x, pcm, pcm_pitch, nch, total_samples, seg, coefs, adpcm,
               adpcm_pitch, seg_state, first_open, seam_flag, seam_end, force_open, rg);
}

// The same work from a queue: as many workgroups as the chip holds at once (cus x 4: two encoder waves and a helper on
// every SIMD, placed once), each taking (channel group, piece) items -- piece-major, so the early items are the ones every
// channel has -- until none is left.  Measured at BASELINE configs[1] with one workgroup per item and four pieces per
// channel (profiles/r04_a_wave_ends.log): the four pieces of a channel group share a CU, the groups' frames differ in how
// often they take the third-trip block, and the CUs end between 136 and 162 ms (mean 147) -- the launch lasts as long as
// its slowest CU.  More, shorter pieces from a plain grid made it worse (212 ms with eight): workgroups of the second
// round land on whichever SIMD has a free slot and two encoder waves end up sharing one.  Persistent workgroups keep
// their SIMDs and balance at the granularity of an item.
//
// The seams travel with the items: a workgroup that has finished piece y of a group announces it on the counters of the
// seams at the piece's two ends (seam_count[seam - 1][group]); whoever finds the other side already there closes that
// seam at once -- its two encoder waves have the seam code's lane layout (8 channels x 8 predictors), the group's sixteen
// channels are theirs -- while the other workgroups carry on with their items.  What the separate seam launch cost at the
// end of the encode (8 ms at four pieces per channel, as long as its slowest seam; 36 ms at 64 pieces) runs underneath
// the other workgroups' items.  A seam reads what two other workgroups wrote (the piece state before it, the bytes
// after it): both sides put a device-scope fence between their stores / loads and the counter.
template <int CPW, bool RAGGED>
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void gc_encode_persistent_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg,
    const int16_t *__restrict__ coefs, const int16_t *__restrict__ hist1,
    const int16_t *__restrict__ hist2, uint8_t *adpcm, int64_t adpcm_pitch, int16_t *seg_state,
    const Ragged rg, int groups, int segments, int *__restrict__ queue, int *__restrict__ seam_count, int *first_open,
    int *seam_flag, int *seam_end, int force_open)
{
    static_assert(CPW == 8, "the seams inside the persistent kernel use the (channel, predictor) layout");
    constexpr int CS = Lay<CPW>::CS;
    __shared__ int s_item;
    __shared__ int s_seam[2];
    const int items = RAGGED && rg.items ? rg.n_items : groups * segments;
#ifdef VGA_DEBUG_TIMESTAMPS
    if (threadIdx.x == 0) g_vga_enc_ts[2 * (blockIdx.x & 32767)] = wall_clock64();
#endif
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(queue, 1);
        __syncthreads();
        const int item = s_item;
        __syncthreads();                               // everyone has read it before thread 0 takes the next one
        if (item >= items) {
#ifdef VGA_DEBUG_TIMESTAMPS
            if (threadIdx.x == 0) g_vga_enc_ts[2 * (blockIdx.x & 32767) + 1] = wall_clock64();
#endif
            return;
        }
        int g = item % groups, y = item / groups;      // piece-major ...
        if (RAGGED && rg.items) {                      // ... or the host's list: biggest items first
            const uint32_t v = rg.items[item];
            g = (int)(v & 0xFFFFFu);
            y = (int)(v >> 20);
        }
        gc_encode_piece<false, CPW, RAGGED>(g, y, pcm, pcm_pitch, nch, total_samples, seg, coefs, hist1, hist2, adpcm, adpcm_pitch,
                                            seg_state, (const int *)nullptr, rg);
        if (segments <= 1) continue;
        // ---- the seams at this piece's ends
        __threadfence();                               // this workgroup's bytes and piece state are out before it says so
        __syncthreads();
        if (threadIdx.x == 0) {
            // (pieces past the end of the group's longest channel do not exist: nobody waits for them)
            const int total_wg = RAGGED ? rg.length[rg.order[g * CS]] : total_samples;
            auto exists = [&](int piece) { return piece < segments && seg.first(piece) * 14 < total_wg; };
            int at_start = 0, at_end = 0;
            if (exists(y)) {
                if (y >= 1 && atomicAdd(&seam_count[(int64_t)(y - 1) * groups + g], 1) == 1) at_start = y;
                if (exists(y + 1) && atomicAdd(&seam_count[(int64_t)y * groups + g], 1) == 1) at_end = y + 1;
            }
            s_seam[0] = at_start;
            s_seam[1] = at_end;
        }
        __syncthreads();
        const int seam_a = s_seam[0], seam_b = s_seam[1];
        if ((seam_a | seam_b) != 0) {
            __threadfence();                           // the other workgroup's stores, before anything of them is read
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (wave < SW) {
                const int slot = g * CS + wave * CPW + (lane >> 3);
                if (seam_a)
                    seam_piece<8>(slot, seam_a, lane, pcm, pcm_pitch, nch, total_samples, seg, coefs, adpcm, adpcm_pitch, seg_state,
                               first_open, seam_flag, seam_end, force_open, rg);
                if (seam_b)
                    seam_piece<8>(slot, seam_b, lane, pcm, pcm_pitch, nch, total_samples, seg, coefs, adpcm, adpcm_pitch, seg_state,
                               first_open, seam_flag, seam_end, force_open, rg);
            }
        }
    }
}

// The channels with an open seam, piece after piece: where the true history V at the start of piece k is not the one the
// seam launch assumed there (T = seg_state[k - 1]), the piece's bytes are the run from T, so the same seam_run from V next
// to a replay from T finds where the two meet -- as a rule a few frames in, and the chain ends unless a later seam of the
// channel was left open too (then V is that seam's recorded end).  Only a run that is still apart at the very end of a
// stream with a partial last frame goes to the serial repair launch (first_open[channel] = last piece; seg_state gets
// that piece's true start), which also remains the fall-back when the scratch cannot hold the flags.
template <int LPC>
__global__ __launch_bounds__(64) void gc_encode_chain_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int total_samples, const Pieces seg, int segments,
    const int16_t *__restrict__ coefs, uint8_t *__restrict__ adpcm, int64_t adpcm_pitch,
    int16_t *__restrict__ seg_state, int *__restrict__ first_open, const int *__restrict__ seam_flag,
    const int *__restrict__ seam_end, int force_open, const Ragged rg)
{
    const int lane = threadIdx.x;
    const int pr = LPC == 16 ? (lane >> 1) & 7 : lane & 7;
    const bool lead = (lane & (LPC - 1)) == 0;          // one lane per channel
    const int slot_raw = blockIdx.x * (64 / LPC) + lane / LPC;
    const bool live = slot_raw < nch;
    const int slot = live ? slot_raw : nch - 1;
    const int ch = rg.order ? rg.order[slot] : slot;
    if (rg.order) total_samples = rg.length[ch];        // ragged: per lane from here on
    const int mine = live ? first_open[ch] : 0x7f7f7f7f;
    int kmin = mine;
#pragma unroll
    for (int o = LPC; o < 64; o <<= 1) kmin = imin(kmin, __shfl_xor(kmin, o));
    if (kmin >= segments) return;                       // no open seam among this wave's channels
    if (live && lead && mine < segments) atomicAdd(&g_vga_gc_stats[5], 1ull);
    int frames_run = 0;
    const int16_t *src = pcm + (rg.order ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    uint8_t *dst = adpcm + (rg.order ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch);
    const int16_t *cf = coefs + ch * 16;
    const int c0 = cf[2 * pr], c1 = cf[2 * pr + 1];
    const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;
    bool have = false;                                  // V differs from what the seam launch assumed at this piece
    int v0 = 0, v1 = 0;                                 // V (x[0], x[1])
    int last_k = 0;
    for (int k = kmin; k < segments; k++) {
        const int64_t f0 = seg.first(k);
        // pieces past a channel's end do not exist (ragged batches: per lane -- such a lane keeps what it had at its
        // own end, for the partial last frame below)
        const bool exists = f0 * 14 < total_samples;
        if (!__any(exists)) break;
        last_k = k;
        const bool is_last = seg.first(k + 1) * 14 >= total_samples || k == segments - 1;
        const int64_t idx = (int64_t)(k - 1) * nch + ch;
        const bool flagged = live && exists && seam_flag[idx] != 0;
        const int ended = seam_end[idx];
        const bool ran = live && have && exists;
        bool open = ran;
        int h0 = v0, h1 = v1;
        if (__any(ran)) {
            const int g2 = seg_state[idx * 2], g1 = seg_state[idx * 2 + 1];      // T: what the piece's bytes were encoded from
            if (ran && lead && is_last) {                                     // the repair launch would start here
                seg_state[idx * 2] = (int16_t)v0;
                seg_state[idx * 2 + 1] = (int16_t)v1;
            }
            seam_run<LPC>(src, dst, c0, c1, coef_ok, pr, ch, k, f0, total_samples, seg.frames(k), seg.frames(k), force_open, h0, h1, g2, g1, open, frames_run);
        }
        if (!exists) {
        } else if (ran && open) {                       // still apart at the piece's end: carry on into the next one
            v0 = h0;
            v1 = h1;
        } else if (flagged) {                           // met the run from T, whose own seam ran out of frames: its end is the truth
            have = true;
            v0 = (int)(int16_t)(ended & 0xFFFF);
            v1 = ended >> 16;
        } else
            have = false;
    }
    // apart to the very end: only the partial last frame is left to encode from V -- here, by the channel's eight lanes (the
    // serial repair launch that used to take over re-ran the whole last piece: 4.9 ms of a 256-channel encode's 31)
    if (__any(live && have && total_samples % 14 != 0))
        encode_tail_frame<LPC>(src, dst, c0, c1, coef_ok, pr, total_samples, v0, v1, live && have && total_samples % 14 != 0);
    (void)last_k;
    if (live && lead && frames_run) atomicAdd(&g_vga_gc_stats[2], (unsigned long long)frames_run);
    if (live && lead) first_open[ch] = 0x7f7f7f7f;
}

// A piece must be longer than the slowest seam of the batch takes to close, or that channel's seams all stay open and the
// chain kernel ends up walking the whole channel serially: channels 64..95 of the synthetic set hold one whose seams need
// ~3000 frames (profiles/r03_b_encode_pieces.log: 96 channels x 60 s in 20.9 ms with 64 pieces of 3214 frames, 60.8 ms
// with 128, 150 ms with 256; without such a channel more pieces only help: 64 channels 6.7 / 5.5 / 4.0 ms).  Round 2 cut
// pieces down to 512 frames.
// (3072 until round 4: at 128 channels that made 66 pieces of 3117 frames and the slow channel's seams stayed open -- 33.5 ms
// against 21.2 ms with 64 pieces of 3215)
constexpr int MIN_PIECE_FRAMES = 3584;

// Ragged batches cut their channels into pieces of ONE length (a channel has as many as it reaches into); with a few
// times more workgroups than the chip holds at once, the ones that end early (short channels, short last pieces) make
// room for the rest instead of leaving their SIMDs idle.
constexpr int RAGGED_OVERSUBSCRIPTION = 4;
// Persistent workgroups from one channel group per 32 workgroups on (512 channels on an MI355X): 512 channels 30.3 ms
// against 34.1 for the plain grid with the same 32 pieces, 1024 channels 42.9 against 47.1 (16 pieces) -- the seams run
// inside instead of in a launch of their own; 256 channels and fewer: no difference or slower (25.0 grid, 26.8 persistent;
// profiles/r04_a_encode_persistent.log).  The two-size schedule only from one group per 8 workgroups on (2048 channels):
// below, a channel has to be cut into so many pieces to fill the queue's rounds that its seams cost more than the balance
// gains (1024 channels: 46-62 ms over the schedules tried; profiles/r04_a_encode_schedules.log).
constexpr int PERSISTENT_MIN_GROUPS_FACTOR = 32;
constexpr int PERSISTENT_SCHEDULE_GROUPS_FACTOR = 8;
constexpr int PERSISTENT_ITEMS_PER_WORKGROUP = 4;      // uniform pieces (the test hook's mode 2 without a schedule): items per workgroup
constexpr int PERSISTENT_BIG_ROUNDS = 3;       // a workgroup's share of the frames in this many big items ...
constexpr int PERSISTENT_SMALL_ROUNDS = 2;     // ... followed by this many rounds of short items
constexpr int PERSISTENT_SMALL_FRAMES = 4096;  // ... of this many frames
constexpr int MAX_PIECES = 1024;               // what encode_scratch_bytes() and the ragged item list (capi_gcadpcm_v.hip) are sized for

// The piece schedule (see Pieces, gcadpcm_kernels.hpp).  Plain grid: as many equal pieces as put two encoder waves on
// every SIMD, each at least MIN_PIECE_FRAMES long.  Persistent workgroups: a workgroup's share of the frames in
// PERSISTENT_BIG_ROUNDS big items followed by PERSISTENT_SMALL_ROUNDS rounds of short ones -- with items of one size d
// the workgroups end spread over the last d of the launch (152 ms at configs[1] with 16 pieces of 35 ms: mean life
// 136 ms + d / 2).
// persistent workgroups per compute unit: four (two encoder waves and a helper on every SIMD)
static int persistent_wgs_per_cu()
{
#ifdef VGA_TUNING   // tools/time_corun.py: fewer workgroups leave registers and LDS for another kernel's waves
    if (const char *e = std::getenv("VGA_HIP_GC_WGS_PER_CU")) return imin(imax(std::atoi(e), 1), 4);
#endif
    return 4;
}

int plan_encode_pieces(int groups, int frames, int64_t group_frames, bool ragged, bool *persistent_out, Pieces *seg_out, int layout)
{
    return plan_encode_pieces_on(device_cu_count(), groups, frames, group_frames, ragged, persistent_out, seg_out, layout);
}

// (the schedule as a function of the compute-unit count: no device needed, the CPU suite walks it through
// vga_testing_gc_plan_pieces)
int plan_encode_pieces_on(int cus, int groups, int frames, int64_t group_frames, bool ragged, bool *persistent_out, Pieces *seg_out, int layout)
{
    // persistent workgroups taking (channel group, piece) items from a queue (gc_encode_persistent_kernel) once the batch
    // has enough channel groups; test hook: 1 = never, 2 = always
    const int pmode = encoder_persistent_mode();
    const bool persistent = layout != 4 && (pmode == 2 || (pmode == 0 && (ragged || groups * PERSISTENT_MIN_GROUPS_FACTOR >= cus * 4)));
    int segments = cus * 4 / (groups > 0 ? groups : 1);   // = SW encoder waves on every SIMD
    if (persistent && (pmode == 2 || ragged)) segments = (cus * 4 * PERSISTENT_ITEMS_PER_WORKGROUP + groups - 1) / groups;
    if (ragged) {
        // pieces of total / (items wanted) frames for every channel; `segments` = what the longest channel needs
        const int64_t want = (int64_t)cus * 4 * (persistent ? PERSISTENT_ITEMS_PER_WORKGROUP : RAGGED_OVERSUBSCRIPTION);
        int64_t piece = (group_frames + want - 1) / want;
        if (piece < MIN_PIECE_FRAMES) piece = MIN_PIECE_FRAMES;
        segments = (int)((frames + piece - 1) / piece);
    }
    if (segments > frames / MIN_PIECE_FRAMES) segments = frames / MIN_PIECE_FRAMES;
    if (segments < 1) segments = 1;
    if (segments > MAX_PIECES) segments = MAX_PIECES;
    if (encoder_segments_override() > 0) segments = imin(imin(imax(frames / 64, 1), encoder_segments_override()), MAX_PIECES);   // test hook
    Pieces seg;
    seg.big = (frames + segments - 1) / segments;
    seg.nb = segments;
    seg.small = seg.big;
    if (persistent && encoder_segments_override() <= 0 && frames >= 4 * MIN_PIECE_FRAMES &&
        (ragged || groups * PERSISTENT_SCHEDULE_GROUPS_FACTOR >= cus * 4)) {
        // (ragged batches, mixed-lengths set of bench.py: 4 rounds 151.7 ms, 3: 158.2, 6: 154.5, 8: 169.5; profiles/r04_a_ragged_schedules.log)
        int big_rounds = ragged ? PERSISTENT_BIG_ROUNDS + 1 : PERSISTENT_BIG_ROUNDS, small_rounds = PERSISTENT_SMALL_ROUNDS, small = PERSISTENT_SMALL_FRAMES;
#ifdef VGA_TUNING   // tools/build_variants.sh only: the product never reads its schedule from the environment
        if (const char *e = std::getenv("VGA_HIP_GC_SCHEDULE")) std::sscanf(e, "%d,%d,%d", &big_rounds, &small_rounds, &small);
#endif
        small = imax(small, MIN_PIECE_FRAMES);
        big_rounds = imax(big_rounds, 1);
        const int wgs = cus * persistent_wgs_per_cu();
        const int64_t per_wg = group_frames / wgs + 1;                                      // a workgroup's share of the frames
        int ns = (small_rounds * wgs + groups / 2) / groups;                                // piece indices that make small_rounds rounds of items
        if ((int64_t)ns * small > frames / 2) ns = frames / (2 * small);
        int big = (int)imax((int)((per_wg - (int64_t)small_rounds * small) / big_rounds), small);
        int nb = (int)((frames - (int64_t)ns * small + big - 1) / big);
        if (nb < 1) nb = 1;
        if (!ragged) big = (int)((frames - (int64_t)ns * small + nb - 1) / nb);             // equal big pieces
        if (big < small) big = small;
        seg.big = big;
        seg.nb = nb;
        seg.small = small;
        segments = nb + ns;
        while (segments > 1 && seg.first(segments - 1) >= frames) segments--;
        if (segments > MAX_PIECES) {
            // A long channel in a batch of few groups: the two sizes would need more pieces than the scratch arrays and the
            // host's item list hold.  Equal pieces instead -- the pieces must COVER the channel (nobody encodes what lies
            // past seg.first(segments)).
            segments = MAX_PIECES;
            seg.big = seg.small = (frames + segments - 1) / segments;
            seg.nb = segments;
            while (segments > 1 && seg.first(segments - 1) >= frames) segments--;
        }
    }
    if (seg.first(segments) < frames) {                // (every branch above covers; should one ever not: equal pieces do)
        seg.big = seg.small = (frames + segments - 1) / segments;
        seg.nb = segments;
    }
    *persistent_out = persistent;
    *seg_out = seg;
    return segments;
}

template <int CPW, bool RAGGED>
static int launch_encode_layout(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                                const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                                hipStream_t stream, void *d_scratch, size_t scratch_bytes, const Ragged &rg)
{
    constexpr int CS = Lay<CPW>::CS;
    if (RAGGED) sample_count = rg.max_length;
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    // two encoder waves per SIMD fill the chip: fewer channels than that are cut into time pieces (each at least
    // MIN_PIECE_FRAMES frames: a seam re-encodes a few dozen as a rule; the ones that take longer than their piece go to
    // the chain launch)
    const int groups = (nch + CS - 1) / CS;
    const int cus = device_cu_count();
    const int frames = (sample_count + 13) / 14;
    // the piece schedule: planned by the host for ragged batches (with the item list), here otherwise
    bool persistent = false;
    Pieces seg;
    int segments;
    if (RAGGED && rg.items) {
        persistent = rg.persistent != 0;
        seg = rg.seg;
        segments = rg.segments;
    } else {
        segments = plan_encode_pieces(groups, frames, RAGGED ? rg.total_frames / CS : (int64_t)groups * frames, RAGGED, &persistent, &seg, CPW);
    }
    AsyncBuf scratch;                                  // freed (stream-ordered) on every exit path
    int16_t *seg_state = nullptr;
    int *first_open = nullptr, *seam_flag = nullptr, *seam_end = nullptr, *queue = nullptr, *seam_count = nullptr;
    if (segments > 1 || persistent) {
        const size_t state_bytes = segments > 1 ? (size_t)round_up((int64_t)segments * nch * 2 * (int64_t)sizeof(int16_t), 16) : 0;
        const size_t open_bytes = (size_t)round_up((int64_t)nch * (int64_t)sizeof(int), 16);
        const size_t count_bytes = persistent && segments > 1 ? (size_t)round_up((int64_t)(segments - 1) * groups * (int64_t)sizeof(int), 16) : 0;
        const size_t need = 3 * state_bytes + open_bytes + 16 + count_bytes;
        // the caller's scratch when it brought one (the host pipeline: hipMallocAsync next to busy copy streams stalled
        // its launching thread for up to 300 ms per call), a stream-ordered allocation otherwise
        unsigned char *base = static_cast<unsigned char *>(d_scratch);
        if (!base || scratch_bytes < need) {
            VGA_HIP_TRY(scratch.alloc(need, stream));
            base = scratch.as<unsigned char>();
        }
        seam_flag = reinterpret_cast<int *>(base + state_bytes);
        seam_end = reinterpret_cast<int *>(base + 2 * state_bytes);
        first_open = reinterpret_cast<int *>(base + 3 * state_bytes);
        queue = reinterpret_cast<int *>(base + 3 * state_bytes + open_bytes);
        if (segments > 1) {
            seg_state = reinterpret_cast<int16_t *>(base);
            VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
        } else
            seg_state = nullptr;
        seam_count = queue + 4;
        if (persistent) VGA_HIP_TRY(hipMemsetAsync(queue, 0, 16 + count_bytes, stream));    // the queue's head and every seam's counter
    }
    if (persistent) {
        const int items = RAGGED && rg.items ? rg.n_items : groups * segments;
        const int wgs = imin(items, cus * persistent_wgs_per_cu());
        if constexpr (CPW == 8)
            hipLaunchKernelGGL((gc_encode_persistent_kernel<CPW, RAGGED>), dim3(wgs), dim3(ENC_THREADS), 0, stream, d_pcm, pcm_pitch, nch,
                               sample_count, seg, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch, seg_state, rg, groups, segments, queue,
                               seam_count, first_open, seam_flag, seam_end, force_open_seams());
    } else
        hipLaunchKernelGGL((gc_encode_kernel<false, CPW, RAGGED>), dim3(groups, segments), dim3(ENC_THREADS), 0, stream, d_pcm, pcm_pitch, nch,
                           sample_count, seg, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch, seg_state,
                           (const int *)nullptr, rg);
    VGA_HIP_TRY(hipGetLastError());
    if (segments > 1) {
        if (!persistent) {                             // (the persistent workgroups close the seams themselves)
            // The seam runs in the lane-per-candidate layout (seam_run<16>) where the launch lasts as long as its slowest seam
            // and a batch is likely to hold one: more than 64 channels of the encoder's CPW = 4 layout.  Measured at 60 s per
            // channel (profiles/r06_m_channel_scaling.log): 96 channels 18.3 -> 16.4 ms (seam launch 4.8 + chain 4.6 ms, the
            // synthetic set's slow-closing tone among them), 256: 23.2 -> 23.4, 384: 25.5 -> 25.1; 1 / 8 / 64 channels lose
            // 0.2-0.4 ms (3.06 -> 3.47, 3.48 -> 3.84, 6.79 -> 7.00) and keep the lane-per-predictor runs.
#ifdef VGA_GC_SEAM16_ALWAYS                                          // (debug builds)
            const bool wide = CPW == 4;
#else
            const bool wide = CPW == 4 && nch > 64;
#endif
            if (wide)
                hipLaunchKernelGGL(gc_encode_seam_kernel<16>, dim3((nch + 3) / 4, segments - 1), dim3(64), 0, stream, d_pcm, pcm_pitch, nch,
                                   sample_count, seg, d_coefs, d_adpcm, adpcm_pitch, seg_state, first_open, seam_flag, seam_end,
                                   force_open_seams(), rg);
            else
                hipLaunchKernelGGL(gc_encode_seam_kernel<8>, dim3((nch + 7) / 8, segments - 1), dim3(64), 0, stream, d_pcm, pcm_pitch, nch,
                                   sample_count, seg, d_coefs, d_adpcm, adpcm_pitch, seg_state, first_open, seam_flag, seam_end,
                                   force_open_seams(), rg);
            VGA_HIP_TRY(hipGetLastError());
        }
        // the seams that were still open at the end of their piece, chained piece after piece (none: every wave returns)
#ifdef VGA_GC_SEAM16_ALWAYS
        if (CPW == 4)
#else
        if (CPW == 4 && nch > 64)
#endif
            hipLaunchKernelGGL(gc_encode_chain_kernel<16>, dim3((nch + 3) / 4), dim3(64), 0, stream, d_pcm, pcm_pitch, nch, sample_count,
                               seg, segments, d_coefs, d_adpcm, adpcm_pitch, seg_state, first_open, seam_flag, seam_end,
                               force_open_seams(), rg);
        else
            hipLaunchKernelGGL(gc_encode_chain_kernel<8>, dim3((nch + 7) / 8), dim3(64), 0, stream, d_pcm, pcm_pitch, nch, sample_count,
                               seg, segments, d_coefs, d_adpcm, adpcm_pitch, seg_state, first_open, seam_flag, seam_end,
                               force_open_seams(), rg);
        VGA_HIP_TRY(hipGetLastError());
        // repair: the same encoder, serially over the last piece, for a channel the chain could not finish (a partial
        // last frame after a run that never met; none: every workgroup returns)
        hipLaunchKernelGGL((gc_encode_kernel<true, CPW, RAGGED>), dim3(groups, 1), dim3(ENC_THREADS), 0, stream, d_pcm, pcm_pitch, nch, sample_count,
                           seg, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch, seg_state, (const int *)first_open, rg);
        VGA_HIP_TRY(hipGetLastError());
    }
    return VGA_OK;
}

// upper bound of what launch_encode wants as scratch for nch channels (1024 pieces x (state, flag, end) + 4 B per channel, + alignment)
// (+ the persistent kernel's queue and seam counters: 4 B per channel group and seam)
size_t encode_scratch_bytes(int nch) { return (size_t)(nch > 0 ? nch : 0) * (1024 * 12 + 4 + 256) + 8192; }

int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                  const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                  hipStream_t stream, void *d_scratch, size_t scratch_bytes, const Ragged *rg)
{
    if (rg)                                            // ragged batches: the (channel, predictor) layout only
        return launch_encode_layout<8, true>(d_pcm, 0, nch, 0, d_coefs, d_hist1, d_hist2, d_adpcm, 0, stream, d_scratch,
                                             scratch_bytes, *rg);
    // The wave layout: (channel, predictor) lanes run the two scale candidates of a pair back to back -- fewer instructions
    // per channel, the layout of every batch that fills the chip (and of the persistent workgroups).  A batch too small for
    // those is a few hundred workgroups on 1024 SIMDs, each alone on its SIMD and as fast as its own instruction stream: with a
    // lane per (channel, predictor, candidate) that stream is half as long per frame and there are twice the workgroups
    // (60 s channels: 1 channel 3.0 ms against 4.8, 96: 18.5 / 21.8, 256: 23.3 / 25.7, 384: 25.5 / 31.5, 512: 28.9 / 28.2 with
    // persistent workgroups; tools/time_layouts_small.py).
    int layout = encoder_layout();
    if (layout == 0)                                   // (the test hook that forces persistent workgroups needs their layout)
        layout = encoder_persistent_mode() != 2 && ((nch + 15) / 16) * PERSISTENT_MIN_GROUPS_FACTOR < device_cu_count() * 4 ? 4 : 8;
    if (layout == 4)
        return launch_encode_layout<4, false>(d_pcm, pcm_pitch, nch, sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch, stream,
                                              d_scratch, scratch_bytes, Ragged{});
    return launch_encode_layout<8, false>(d_pcm, pcm_pitch, nch, sample_count, d_coefs, d_hist1, d_hist2, d_adpcm, adpcm_pitch, stream,
                                          d_scratch, scratch_bytes, Ragged{});
}

}  // namespace gc
}  // namespace vga
