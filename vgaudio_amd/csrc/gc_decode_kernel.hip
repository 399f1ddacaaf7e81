// gc_decode_kernel.hip -- GC-ADPCM decoder for gfx950.
//
// Replaces VGAudio/Codecs/GcAdpcm/GcAdpcmDecoder.cs:10-54, bit-exact.
//
// The decoder is a second-order IIR with saturation and a truncating shift per sample (:38-45): serial inside a
// channel.  Lane = channel; a channel's stream is cut into time pieces decoded side by side (every piece but the first
// from a guessed history), gc_decode_fixup_kernel closes the seams, gc_decode_tail_kernel chains the ones that stay
// open.  Rounds 1-2 split the work between a serial wave and three helper waves through LDS tiles (the helpers unpacked
// scale * nibble + 1024 and wrote the samples out); the tiles' 108 KB per 128 channels limited a CU to one workgroup and
// 4096 channels to eight pieces.  The kernel below needs no helpers: 9.9 -> 8.1 ms at configs[1] -- bound by its stores,
// 3 TB/s (LABNOTES.md 4.3) -- and 4.4 -> 1.2 ms for 256 channels.
#include "common.hpp"

#include <algorithm>
#include "gcadpcm_kernels.hpp"

#include <cstdlib>

namespace vga {
namespace gc {

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------------------------
// Lane = channel, one wave per 64 channels and time piece, as many pieces as put one wave on every SIMD (the ADX decoder
// in adx_kernels.hip has the measurements behind this shape).  A lane reads its own
// frames eight at a time (64 contiguous bytes, the next block in flight during this one), takes the nibbles out of the
// loaded dwords with one v_bfe_i32 each, looks its coefficient pair up in the wave's LDS table and runs the recurrence
// (:38-45).  The 8 x 28 bytes of samples a lane produces per block are contiguous in its row; they leave through the
// wave's LDS block turned, fourteen lanes per row, as 16-byte stores of whole 224-byte runs (TURNED; rows that are not
// 16-byte aligned are stored by their own lane, a dword at a time).  Later pieces start from the guess (0, 0) GC_DECODE_WARM
// frames early without storing, so that they have, as a rule, fallen into step with the true run where they begin and
// their seam closes on the first frame gc_decode_fixup_kernel checks.
constexpr int GC_DECODE_WARM = 512;                  // frames; a multiple of 8
constexpr int GC_DECODE_SLOW_SEAM = 2048;            // frames a seam may stay open before it counts as slow (a multiple of 256)
constexpr int GC_DECODE_TAIL_BUDGET = 4096;          // frames one lane of the tail kernel decodes again before it hands over
// RAGGED (the `*_v` entry points): lane i of workgroup x decodes channel order[64 x + i] (longest first), with its own
// length and offsets; a piece exists for a lane only as far as its channel reaches, the wave runs as many blocks as its
// longest lane has (lane 0) and every row of the turned store is guarded by its own block count.
// (REPAIR is a template parameter so that the launch carries a name of its own in profiles: it returns at once as a rule and
// would halve the kernel's average duration)
template <bool TURNED, bool RAGGED, bool REPAIR>
__global__ __launch_bounds__(64) void gc_decode_direct_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ coefs, int nch,
    int total_samples, int seg_frames, const int16_t *__restrict__ hist1, const int16_t *__restrict__ hist2,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ status, const Ragged rg,
    const int *__restrict__ first_open, const int *__restrict__ slow_seams)
{
    const int lane = threadIdx.x;
    const int slot_raw = blockIdx.x * 64 + lane;
    const bool live = slot_raw < nch;
    const int slot = live ? slot_raw : nch - 1;
    const int ch = RAGGED ? rg.order[slot] : slot;
    // REPAIR launch (round 5): the batch holds many seams that would not close -- pure tones, clipped
    // waves: a decoder run from a wrong history never falls into step when the predictor's poles sit on the unit circle.  The
    // wave decodes its 64 channels again as ONE piece, from the first piece any of them left open to the end of the stream,
    // from the samples before it (final: every earlier seam of every lane closed).  This is the serial floor -- a lone wave
    // needs ~57 ms for 60 s -- against 1.36 s for the chained tail kernel on a batch of 440 Hz sines (bench.py signal_sensitivity).
    constexpr bool repair = REPAIR;
    int repair_piece = 0;
    if (repair) {
        if (slow_seams[0] < slow_seams[1]) return;                        // few open seams: gc_decode_tail_kernel has them
        int k = live ? first_open[ch] : 0x7f000000;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) k = imin(k, __shfl_xor(k, o));
        if (k <= 0 || k >= 0x7f000000) return;                            // none of these channels has an open seam
        repair_piece = k;
    }
    const int64_t first_frame = (int64_t)(repair ? repair_piece : (int)blockIdx.y) * seg_frames;   // a multiple of 8 (seg_frames is)
    if (repair) seg_frames = 0x7fffff00 / 14;                              // ... to the end of the stream
    const int64_t first_sample = first_frame * 14;
    if (RAGGED) {
        if (first_sample >= rg.length[rg.order[blockIdx.x * 64]]) return;  // the wave's longest channel ends before this piece
        total_samples = rg.length[ch];
    } else if (first_sample >= total_samples)
        return;
    const bool exists = !RAGGED || first_sample < total_samples;           // this lane's channel reaches the piece
    const int sample_count = !exists ? 0 : (int)((int64_t)total_samples - first_sample < (int64_t)seg_frames * 14
                                                     ? (int64_t)total_samples - first_sample : (int64_t)seg_frames * 14);
    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    // a lane without the piece points at its channel's first bytes (valid memory for its clamped loads) and stores nothing
    const uint8_t *src = adpcm + (RAGGED ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch) + (exists ? first_frame * 8 : 0);
    int16_t *dst = pcm + (RAGGED ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch) + (exists ? first_sample : 0);
    __shared__ uint32_t s_cf[8 * 64];                                      // [predictor][lane]: (coef1 & 0xFFFF) | coef2 << 16
    __shared__ int4 s_turn[64 * 15];                                       // 64 rows of 14 int4 (8 frames), one int4 apart from a multiple of 8
    __shared__ int64_t s_row[RAGGED ? 64 : 1];                             // RAGGED: a row's first sample of the piece, counted from pcm
    __shared__ int s_rowblocks[RAGGED ? 64 : 1];                           // ... and its whole blocks of eight frames
    if (RAGGED) {
        s_row[lane] = (int64_t)(dst - pcm);
        s_rowblocks[lane] = live ? full_frames / 8 : 0;
    }
#pragma unroll
    for (int q = 0; q < 8; q++)
        s_cf[q * 64 + lane] = (uint32_t)(uint16_t)coefs[ch * 16 + 2 * q] | ((uint32_t)(uint16_t)coefs[ch * 16 + 2 * q + 1] << 16);
    int h1 = 0, h2 = 0;
    if (repair) {
        if (exists) { h1 = dst[-1]; h2 = dst[-2]; }
    } else if (blockIdx.y == 0) {
        h1 = hist1 ? hist1[ch] : 0;
        h2 = hist2 ? hist2[ch] : 0;
    }
    bool bad_seen = false;
    bool bad = false;
    // one frame (bits = its 8 bytes, little-endian) -> 14 samples as 7 packed pairs
    auto decode_frame = [&](uint32_t lo, uint32_t hi, uint32_t (&o7)[7]) {
        const int ps = (int)(lo & 0xFF);
        const int sh = (ps & 0xF) + 11;                                    // scale = (1 << (ps & 0xF)) * 2048 (:26)
        int predictor = (ps >> 4) & 0xF;                                   // :27
        if (predictor > 7) { bad = true; predictor &= 7; }
        const uint32_t cf = s_cf[predictor * 64 + lane];
        const int c1 = (int)(int16_t)(cf & 0xFFFF), c2 = (int)cf >> 16;
        int o[14];
#pragma unroll
        for (int s = 0; s < 14; s++) {
            const int b = 1 + (s >> 1);                                    // the byte that holds sample s: high nibble first
            const uint32_t w = b < 4 ? lo : hi;
            const int nib = __builtin_amdgcn_sbfe((int)w, 8 * (b & 3) + ((s & 1) ? 0 : 4), 4);   // SignedNibbles (Helpers.cs:50)
            // :38-45: (coef1*hist1 + coef2*hist2 + scale*nibble + 1024) >> 11, clamped; int32 wrap like the reference
            int rest = __mul24(c2, h2) + (int)(((uint32_t)nib << sh) + 1024u);
            asm("" : "+v"(rest));
            const int t = __mul24(c1, h1) + rest;
            const int v = imin(imax(t >> 11, -32768), 32767);
            h2 = h1;
            h1 = v;
            o[s] = v;
        }
#pragma unroll
        for (int q = 0; q < 7; q++) o7[q] = (uint32_t)(o[2 * q] & 0xFFFF) | ((uint32_t)o[2 * q + 1] << 16);
    };
    auto turned_row = [&](int i, int l) {                                  // rows past the last channel: channel nch - 1 again,
        const int c = blockIdx.x * 64 + l / 14 + 4 * i;                    // whose samples those lanes hold -- unconditional stores
        return pcm + (int64_t)(c < nch ? c : nch - 1) * pcm_pitch + first_sample + (l % 14) * 8;
    };
    // ---- warm-up of a later piece (not stored)
    if (!repair && blockIdx.y > 0 && exists) {
        const int warm = (int)(first_frame < GC_DECODE_WARM ? first_frame : GC_DECODE_WARM);
        const uint2 *wsrc = reinterpret_cast<const uint2 *>(src) - warm;
#pragma unroll 1
        for (int k = 0; k < warm; k += 4) {                                // warm is a multiple of 8
            uint2 f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = wsrc[k + j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t o7[7];
                decode_frame(f[j].x, f[j].y, o7);
            }
        }
    }
    bad_seen = bad;                                                        // (the warm-up reads real frames: a bad header there is a bad header)
    // ---- whole blocks of eight full frames
    const int my_blocks = full_frames / 8;
    const int blocks = RAGGED ? __shfl(my_blocks, 0) : my_blocks;          // lane 0 holds the wave's longest channel
    const uint4 *bsrc = reinterpret_cast<const uint4 *>(src);
    uint4 cur[4], nxt[4];
#pragma unroll
    for (int q = 0; q < 4; q++) cur[q] = blocks > 0 ? bsrc[q] : make_uint4(0, 0, 0, 0);
    // the first block is waited for HERE, not at the loop header (where the wait would also mean "every store of the
    // block before has completed")
#pragma unroll
    for (int q = 0; q < 4; q++) asm volatile("" : "+v"(cur[q].x), "+v"(cur[q].y), "+v"(cur[q].z), "+v"(cur[q].w));
#pragma unroll 1
    for (int k = 0; k < blocks; k++) {
        const uint4 *f = bsrc + (int64_t)(RAGGED ? imax(imin(k + 1, my_blocks - 1), 0) : imin(k + 1, blocks - 1)) * 4;
        const int keep1 = h1, keep2 = h2;                                  // RAGGED: a lane past its last block keeps its history
#pragma unroll
        for (int q = 0; q < 4; q++) nxt[q] = f[q];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint32_t d28[28];                                              // four frames = 112 bytes = seven int4
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint4 v = cur[2 * half + (j >> 1)];
                uint32_t o7[7];
                decode_frame((j & 1) ? v.z : v.x, (j & 1) ? v.w : v.y, o7);
#pragma unroll
                for (int q = 0; q < 7; q++) d28[7 * j + q] = o7[q];
            }
            if (TURNED) {
#pragma unroll
                for (int q = 0; q < 7; q++)
                    s_turn[lane * 15 + half * 7 + q] = make_int4((int)d28[4 * q], (int)d28[4 * q + 1], (int)d28[4 * q + 2], (int)d28[4 * q + 3]);
            } else if (live && (!RAGGED || k < my_blocks)) {
                uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + ((int64_t)k * 8 + half * 4) * 14);
#pragma unroll
                for (int q = 0; q < 28; q++) d32[q] = d28[q];
            }
        }
        if (TURNED) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int l = imin(lane, 55);                                  // 4 rows x 14 lanes per store; the last eight lanes repeat lane 55's
#pragma unroll
            for (int i = 0; i < 16; i++) {                                 // read, store -- one at a time (see adx_kernels.hip)
                const int row = l / 14 + 4 * i;
                if (RAGGED) {
                    if (k < s_rowblocks[row])
                        *reinterpret_cast<int4 *>(pcm + s_row[row] + (l % 14) * 8 + (int64_t)k * 8 * 14) = s_turn[row * 15 + l % 14];
                } else
                    *reinterpret_cast<int4 *>(turned_row(i, l) + (int64_t)k * 8 * 14) = s_turn[row * 15 + l % 14];
            }
            asm volatile("" ::: "memory");
        }
        if (RAGGED && k >= my_blocks) {                                    // blocks past this lane's end decoded whatever was there
            bad = bad_seen;
            h1 = keep1;
            h2 = keep2;
        }
        bad_seen = bad;
#pragma unroll
        for (int q = 0; q < 4; q++) cur[q] = nxt[q];
    }
    // ---- what is left of the piece: fewer than eight full frames and a partial one
#pragma unroll 1
    for (int fr = my_blocks * 8; fr < full_frames + (tail ? 1 : 0); fr++) {
        uint32_t lo = 0, hi = 0;
        if (fr < full_frames) {
            const uint2 v = reinterpret_cast<const uint2 *>(src)[fr];
            lo = v.x;
            hi = v.y;
        } else {                                                           // partial last frame: only its bytes exist
            const int nbytes = (tail + 2 + 1) / 2;
            uint64_t bits = 0;
            for (int b = 0; b < nbytes; b++) bits |= (uint64_t)src[(int64_t)fr * 8 + b] << (8 * b);
            lo = (uint32_t)bits;
            hi = (uint32_t)(bits >> 32);
        }
        uint32_t o7[7];
        decode_frame(lo, hi, o7);
        if (!live) continue;
        int16_t *d = dst + (int64_t)fr * 14;
        if (fr < full_frames) {
#pragma unroll
            for (int q = 0; q < 7; q++) reinterpret_cast<uint32_t *>(d)[q] = o7[q];
        } else {
            for (int s2 = 0; s2 < tail; s2++) d[s2] = (int16_t)(o7[s2 >> 1] >> (16 * (s2 & 1)));
        }
    }
    if (bad && live && status) atomicOr(status, 1);
}

// One frame of GcAdpcmDecoder.Decode (:25-45) from the history (h1, h2) into o[0 .. valid).  A full frame is one 8-byte load
// (round 5: a byte at a time until then -- nine dependent loads a frame on a path that can walk a whole channel).
__device__ __forceinline__ void gc_decode_frame_serial(const uint8_t *fr, const int (&cf)[16], int valid, int &h1, int &h2,
                                                       int16_t *o)
{
    uint64_t bits = 0;
    if (valid == 14) {
        const uint2 v = *reinterpret_cast<const uint2 *>(fr);
        bits = ((uint64_t)v.y << 32) | v.x;
    } else {
        const int nbytes = (valid + 2 + 1) / 2;                             // the stream's last, partial frame: only its bytes exist
        for (int b = 0; b < nbytes; b++) bits |= (uint64_t)fr[b] << (8 * b);
    }
    const int ps = (int)(bits & 0xFF);
    const int scale = (1 << (ps & 0xF)) * 2048;
    const int predictor = (ps >> 4) & 7;
    int c1 = 0, c2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (predictor == i) { c1 = cf[2 * i]; c2 = cf[2 * i + 1]; }
    int16_t out14[14];
#pragma unroll
    for (int s = 0; s < 14; s++) {
        const int byte = (int)((bits >> (8 * (1 + (s >> 1)))) & 0xFF);
        const int nib = (s & 1) ? (byte & 0xF) : (byte >> 4);
        const int v = imin(imax((c1 * h1 + c2 * h2 + scale * ((nib ^ 8) - 8) + 1024) >> 11, -32768), 32767);
        if (s < valid) {
            h2 = h1;
            h1 = v;
        }
        out14[s] = (int16_t)v;
    }
    if (valid == 14) {
        uint32_t *o32 = reinterpret_cast<uint32_t *>(o);                    // rows and frames are 4-byte aligned (28-byte frames)
#pragma unroll
        for (int q = 0; q < 7; q++) o32[q] = (uint32_t)(uint16_t)out14[2 * q] | ((uint32_t)(uint16_t)out14[2 * q + 1] << 16);
    } else {
        for (int s = 0; s < valid; s++) o[s] = out14[s];
    }
}

// Closes the seams between time segments: one lane per (channel, seam), all seams at once.  From the history the piece
// before ended on (its last two samples: final provided THAT piece's own seam closes) decode again frame by frame over
// the guessed run's samples until both histories coincide at a frame end -- from there on the guessed run decoded
// exactly what the serial decoder would have.  A seam that does not close inside its piece records its index in
// first_open[channel]; gc_decode_tail_kernel then decodes that channel serially from the next piece on.  Always exact.
__global__ __launch_bounds__(64) void gc_decode_fixup_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ coefs, int nch,
    int total_samples, int seg_frames, int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ first_open,
    int *__restrict__ seam_open, int force_open, const Ragged rg, int *__restrict__ slow_seams)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int k = blockIdx.y + 1;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (slot >= nch) return;
    const int ch = rg.order ? rg.order[slot] : slot;
    if (rg.order) total_samples = rg.length[ch];
    if (f0 * 14 >= total_samples) return;
    const uint8_t *src = adpcm + (rg.order ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch);
    int16_t *dst = pcm + (rg.order ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    int cf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cf[i] = coefs[ch * 16 + i];
    const int full_frames = total_samples / 14;
    // Seed = the last two samples of piece k-1, read while seam k-1's lane (another thread, all seams run at once) may
    // still be rewriting that piece.  Invariant that makes this safe: if seam k-1 CLOSES, the samples it rewrites past
    // its closing frame are untouched and the ones before it get the values of the serial run -- the tail of piece
    // k-1, which is what is read here, is identical before and after (a closing seam never reaches the last frame
    // without having matched the guessed run there, i.e. it rewrites those two samples with the values they hold);
    // if seam k-1 stays OPEN, first_open[ch] <= k-1 and gc_decode_tail_kernel decodes pieces k.. again from the final
    // samples, overwriting whatever this lane produced from a possibly stale seed.  Either way the output is exact.
    int h1 = dst[f0 * 14 - 1], h2 = dst[f0 * 14 - 2];
    // Round 5: a seam still open after GC_DECODE_SLOW_SEAM frames counts as slow (slow_seams[0]); once the batch holds
    // slow_seams[1] of them -- a batch of tones: their seams never close -- the lanes stop walking their pieces (12 800 frames
    // each at configs[1]) and leave everything from their piece on to the REPAIR launch of the direct kernel, which decodes
    // the affected waves as one piece.  Below that count nothing changes: a seam runs to its piece's end and
    // gc_decode_tail_kernel chains the few that stay open.  (A lane only gives up when the count has been reached, so
    // "somebody gave up" implies the REPAIR launch runs.)
    int walked = 0;
    bool counted = false, gave_up = false;
    // (seams the test hook holds open are not counted -- modes 1 and 2 exercise the tail kernel as before -- unless it asks for it: 3)
    const bool countable = !seam_forced_open(force_open, ch, k) || force_open == 3;
    for (int64_t f = f0; f < f0 + seg_frames && f * 14 < total_samples; f++) {
        const int valid = f < full_frames ? 14 : total_samples - (int)(f * 14);
        int16_t *o = dst + f * 14;
        int g1 = 0, g2 = 0;                            // the guessed run's history at this frame's end
        if (valid == 14) { g1 = o[13]; g2 = o[12]; }
        gc_decode_frame_serial(src + f * 8, cf, valid, h1, h2, o);
        if (valid == 14 && h1 == g1 && h2 == g2 && !seam_forced_open(force_open, ch, k)) return;
        if (valid < 14) return;                        // the stream's last, partial frame: nothing follows
        if (++walked == GC_DECODE_SLOW_SEAM && countable) {
            atomicAdd(&slow_seams[0], 1);
            counted = true;
        }
        if (walked >= GC_DECODE_SLOW_SEAM && (walked & 255) == 0 &&
            __hip_atomic_load(&slow_seams[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slow_seams[1]) {
            gave_up = true;
            break;
        }
    }
    const bool piece_follows = (f0 + seg_frames) * 14 < total_samples;
    if (!counted && piece_follows && countable) atomicAdd(&slow_seams[0], 1);   // (a piece shorter than the limit that never closed)
    if (piece_follows || gave_up) {                    // open (and a piece follows), or this piece itself is left unfinished
        if (piece_follows) seam_open[(int64_t)(k - 1) * nch + ch] = 1;
        atomicMin(&first_open[ch], k);
    }
}

// Channels with an open seam (practically none): seam k re-decoded all of piece k, so that piece is final; everything
// after it was seeded from samples that have changed since and is decoded again here, serially.
// The channels with an open seam, piece after piece (one lane per channel; lanes without one leave at once).  A seam
// that ran out of frames has made ITS piece final, but the piece after it was seeded from samples that have changed
// since: that piece is decoded again from the final samples -- next to what it holds, which is a run of the same
// recurrence from some other history -- until both agree at a frame end; from there on the stored samples are the
// serial decoder's.  (Round 1 decoded the whole rest of the channel again: 220 ms for a 60 s channel.)  A run that
// does not meet by the end of a piece carries on into the next one; a later open seam of the channel starts the same
// again.  seam_open[(k - 1) * nch + ch] != 0: seam k (the start of piece k) stayed open.
__global__ __launch_bounds__(64) void gc_decode_tail_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ coefs, int nch,
    int total_samples, int seg_frames, int segments, int16_t *__restrict__ pcm, int64_t pcm_pitch,
    int *__restrict__ first_open, const int *__restrict__ seam_open, int force_open, const Ragged rg,
    int *__restrict__ slow_seams)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    if (slot >= nch) return;
    // many seams that would not close -- or a lane of this launch has handed a channel over (below): the REPAIR launch of the
    // direct kernel runs, and it takes every channel whose first_open is still set, this one included
    if (__hip_atomic_load(&slow_seams[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= slow_seams[1]) return;
    const int ch = rg.order ? rg.order[slot] : slot;
    if (rg.order) total_samples = rg.length[ch];
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    int walked_total = 0;                              // frames this lane has decoded again (see GC_DECODE_TAIL_BUDGET)
    const uint8_t *src = adpcm + (rg.order ? rg.adpcm_off[ch] : (int64_t)ch * adpcm_pitch);
    int16_t *dst = pcm + (rg.order ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    int cf[16];
#pragma unroll
    for (int i = 0; i < 16; i++) cf[i] = coefs[ch * 16 + i];
    const int full_frames = total_samples / 14;
    bool carry = false;                                // the piece before ended on samples other than the ones this piece was seeded from
    int h1 = 0, h2 = 0;
    for (int k = k0; k < segments; k++) {
        const int64_t f0 = (int64_t)k * seg_frames;
        if (f0 * 14 >= total_samples) break;
        const bool flagged = seam_open[(int64_t)(k - 1) * nch + ch] != 0;
        bool apart = false;
        if (carry) {
            apart = true;
            for (int64_t f = f0; f < f0 + seg_frames && f * 14 < total_samples; f++) {
                const int valid = f < full_frames ? 14 : total_samples - (int)(f * 14);
                int16_t *o = dst + f * 14;
                int g1 = 0, g2 = 0;
                if (valid == 14) { g1 = o[13]; g2 = o[12]; }
                gc_decode_frame_serial(src + f * 8, cf, valid, h1, h2, o);
                walked_total++;
                if (valid == 14 && h1 == g1 && h2 == g2 && !seam_forced_open(force_open, ch, k)) { apart = false; break; }
            }
        }
        const int64_t f1 = f0 + seg_frames;            // the next piece's first frame
        if (apart) {
            carry = true;                              // (h1, h2): the true samples at the end of this piece
            // A run that has not met after GC_DECODE_TAIL_BUDGET frames (a pure tone: it never will) is not walked to the end
            // of the stream by ONE lane: the pieces up to this one are final now, the REPAIR launch decodes the channel's wave
            // from the next piece on at the direct kernel's speed.  Seams the test hook holds open do not count.
            if (walked_total >= GC_DECODE_TAIL_BUDGET && f1 * 14 < total_samples && (force_open == 0 || force_open == 3)) {
                first_open[ch] = k + 1;
                atomicMax(&slow_seams[0], slow_seams[1]);
                return;
            }
        } else if (flagged && f1 * 14 < total_samples) {
            carry = true;                              // this piece's own seam ran out of frames: the piece is final, its end the truth
            h1 = dst[f1 * 14 - 1];
            h2 = dst[f1 * 14 - 2];
        } else
            carry = false;
    }
    first_open[ch] = 0x7f7f7f7f;                       // done: nothing of this channel is left for the REPAIR launch
}

int launch_decode(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch, int sample_count,
                  const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int64_t pcm_pitch, int *d_status,
                  hipStream_t stream, const Ragged *rgp)
{
    const Ragged rg = rgp ? *rgp : Ragged{};
    if (rgp) sample_count = rg.max_length;
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    // as many time pieces as put one wave on every SIMD (a wave = 64 channels of one piece), each at least 1024 frames long
    // and a multiple of eight frames; at most 64 (every seam is a chance of a run that never meets, see adx_kernels.hip)
    const int frames = (sample_count + 13) / 14;
    const int groups = (nch + 63) / 64;
    const int cus = device_cu_count();
    int segments = cus * 4 / groups;
    if (rgp) {
        // ragged: pieces of one length for every channel, twice as many waves as the chip holds at once (the short ones
        // make room), `segments` = what the longest channel needs
        const int64_t want = (int64_t)cus * 4 * 2;
        const int64_t piece = std::max<int64_t>((rg.total_frames + 64 * want - 1) / (64 * want), 1024);
        segments = (int)((frames + piece - 1) / piece);
    }
    if (segments > frames / 1024) segments = frames / 1024;
    if (segments < 1) segments = 1;
    if (segments > 64) segments = 64;
    if (encoder_segments_override() > 0) segments = std::min(std::max(frames / 8, 1), encoder_segments_override());   // test hook
    const int seg_frames = ((frames + segments - 1) / segments + 7) / 8 * 8;
    // rows of samples on 16-byte boundaries: whole 224-byte runs leave as 16-byte stores
    const bool turned = (pcm_pitch % 8) == 0 && ((uintptr_t)d_pcm % 16) == 0;
#define VGA_GC_DIRECT(TURNED_, RAGGED_, REPAIR_, PIECES_, OPEN_, SLOW_)                                                          \
    hipLaunchKernelGGL((gc_decode_direct_kernel<TURNED_, RAGGED_, REPAIR_>), dim3(groups, PIECES_), dim3(64), 0, stream, d_adpcm, \
                       adpcm_pitch, d_coefs, nch, sample_count, seg_frames, d_hist1, d_hist2, d_pcm, pcm_pitch, d_status, rg,     \
                       (const int *)(OPEN_), (const int *)(SLOW_))
    auto direct = [&](int pieces, const int *first_open, const int *slow_seams) {
        const bool repair = first_open != nullptr;
        if (rgp) {                                     // (the ragged layout keeps every row on a 16-byte boundary)
            if (repair) VGA_GC_DIRECT(true, true, true, pieces, first_open, slow_seams);
            else VGA_GC_DIRECT(true, true, false, pieces, nullptr, nullptr);
        } else if (turned) {
            if (repair) VGA_GC_DIRECT(true, false, true, pieces, first_open, slow_seams);
            else VGA_GC_DIRECT(true, false, false, pieces, nullptr, nullptr);
        } else {
            if (repair) VGA_GC_DIRECT(false, false, true, pieces, first_open, slow_seams);
            else VGA_GC_DIRECT(false, false, false, pieces, nullptr, nullptr);
        }
        return hipGetLastError();
    };
#undef VGA_GC_DIRECT
    VGA_HIP_TRY(direct(segments, nullptr, nullptr));
    if (segments > 1) {
        AsyncBuf scratch;                              // freed (stream-ordered) on every exit path
        const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
        VGA_HIP_TRY(scratch.alloc((size_t)nch * sizeof(int) + flag_bytes + 16, stream));
        int *first_open = scratch.as<int>();
        int *seam_open = first_open + nch;
        int *slow_seams = seam_open + (size_t)(segments - 1) * nch;        // [0] seams that stayed open, [1] how many make "many"
        VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
        VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes + 16, stream));
        // "many": one seam in 64, and at least 8 (the synthetic set's handful of slow channels stays with the tail kernel, whose
        // runs meet again after a while; a batch of tones has every seam open)
        const int many = std::max<int64_t>(8, (int64_t)nch * (segments - 1) / 64) > 0x7fffffff ? 0x7fffffff
                       : (int)std::max<int64_t>(8, (int64_t)nch * (segments - 1) / 64);
        // (a fill, not a copy from this stack frame: a pageable host-to-device copy makes the call wait for the stream)
        VGA_HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(slow_seams + 1), many, 1, stream));
        hipLaunchKernelGGL(gc_decode_fixup_kernel, dim3((nch + 63) / 64, segments - 1), dim3(64), 0, stream, d_adpcm, adpcm_pitch,
                           d_coefs, nch, sample_count, seg_frames, d_pcm, pcm_pitch, first_open, seam_open, force_open_seams(), rg, slow_seams);
        VGA_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gc_decode_tail_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_adpcm, adpcm_pitch, d_coefs, nch,
                           sample_count, seg_frames, segments, d_pcm, pcm_pitch, first_open, seam_open, force_open_seams(), rg, slow_seams);
        VGA_HIP_TRY(hipGetLastError());
        VGA_HIP_TRY(direct(1, first_open, slow_seams));
    }
    return VGA_OK;
}

}  // namespace gc
}  // namespace vga
