// gc_decode_kernel.hip -- GC-ADPCM decoder for gfx950, serial wave + helper waves.
//
// Replaces VGAudio/Codecs/GcAdpcm/GcAdpcmDecoder.cs:10-54, bit-exact.
//
// The decoder is a second-order IIR with saturation and a truncating shift per sample (:38-45): serial
// inside a channel, and 4096 channels are only 64 waves -- so time = frames x (instructions per frame on
// the wave that carries the recurrence).  As in gc_encode_kernel.hip everything that does not depend on the
// history leaves that wave:
//   helper waves (3 per workgroup), one tile of frames AHEAD: load the 8-byte frames, split the header
//     (:25-29), look the coefficient pair up, turn every nibble into scale * nibble + 1024 (:36-37, :41 with
//     the rounding constant folded in) and lay all of it out in LDS so that the decoder's reads are one
//     conflict-free b128 per lane; they also write the previous tile's samples to global memory;
//   decoder wave (lane = channel, 64 channels per workgroup): per sample mad, mad, shift, clamp.
#include "common.hpp"

#include <algorithm>
#include "gcadpcm_kernels.hpp"

#include <cstdlib>

namespace vga {
namespace gc {

constexpr int DTF = 4;                    // frames per tile (2 would fit two workgroups per CU, i.e. twice the pieces: 15.4
                                          // instead of 10.0 ms at configs[1] -- the barrier every other frame costs more)
constexpr int DCW = 128;                  // channels per workgroup: TWO decoder waves (they land on different SIMDs of the
                                          // CU; two 64-channel workgroups would put both of theirs on SIMD 0) + 6 helpers
constexpr int DTHREADS = DCW * 4;
constexpr int DHELPERS = DTHREADS - DCW;
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

struct GcDecodeTile {
    int4 dist[DTF][4][DCW];                // [frame][quarter][channel]: 14 x (scale*nibble + 1024), 2 padding
    int2 coef[DTF][DCW];                   // [frame][channel]: (coef1, coef2) of the frame's predictor
    int4 out[DTF][2][DCW];                 // [frame][half][channel]: 14 samples as 7 packed pairs, 1 padding
};

// Time segments (blockIdx.y): a channel's stream is cut into pieces of `seg_frames` frames that are decoded side by
// side.  Segment 0 starts from the caller's history; the others start from (0, 0) -- a guess -- and
// gc_decode_fixup_kernel afterwards re-decodes the head of each until its history meets the guessed run's.
__global__ __launch_bounds__(DTHREADS) void gc_decode_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ coefs, int nch,
    int total_samples, int seg_frames, const int16_t *__restrict__ hist1, const int16_t *__restrict__ hist2,
    int16_t *__restrict__ pcm, int64_t pcm_pitch, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    GcDecodeTile *s_tile = reinterpret_cast<GcDecodeTile *>(s_raw);            // [2]
    int16_t *s_coefs = reinterpret_cast<int16_t *>(s_raw + 2 * sizeof(GcDecodeTile));   // [DCW][16]: the coefficient sets of the workgroup's DCW (= 128) channels

    const int tid = threadIdx.x;
    const int ch0 = blockIdx.x * DCW;
    // this workgroup's segment, seen as a stream of its own: frames are 8 bytes / 14 samples, so both rows stay
    // aligned (28-byte sample offsets for the dword stores, 8-byte frame offsets for the loads)
    const int64_t first_frame = (int64_t)blockIdx.y * seg_frames;
    const int64_t first_sample = first_frame * 14;
    if (first_sample >= total_samples) return;
    const int sample_count = (int)((int64_t)total_samples - first_sample < (int64_t)seg_frames * 14
                                       ? (int64_t)total_samples - first_sample : (int64_t)seg_frames * 14);
    adpcm += first_frame * 8;
    pcm += first_sample;
    if (blockIdx.y > 0) hist1 = hist2 = nullptr;
    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    const int frames = full_frames + (tail ? 1 : 0);
    const int tiles = (frames + DTF - 1) / DTF;

    for (int i = tid; i < DCW * 16; i += DTHREADS) {
        const int c = imin(ch0 + (i >> 4), nch - 1);
        s_coefs[i] = coefs[c * 16 + (i & 15)];
    }
    __syncthreads();

    if (tid >= DCW) {
        // ------------------------------------------------------------ helper waves (192 lanes)
        const int hl = tid - DCW;
        bool bad = false;
        // the lane's (up to 3) frames of a tile, loaded a whole tile period before they are unpacked:
        // unconditional loads with a clamped frame index (a load under a divergent condition is waited for
        // at once, and the helpers would then pay three HBM round trips per tile)
        constexpr int ITEMS = (DCW * DTF + DHELPERS - 1) / DHELPERS;
        auto load_tile = [&](int tile, uint2 (&raw)[ITEMS]) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const int item = imin(hl + DHELPERS * k, DCW * DTF - 1);
                const int c = item / DTF, j = item - c * DTF;
                const int fr = imin(tile * DTF + j, imax(full_frames - 1, 0));
                const int ch = imin(ch0 + c, nch - 1);
                // (no full frame at all: frame 0's 8 bytes still lie inside the row -- pitch is a multiple of 8)
                raw[k] = *reinterpret_cast<const uint2 *>(adpcm + (int64_t)ch * adpcm_pitch + (int64_t)fr * 8);
            }
        };
        auto prepare = [&](int tile, const uint2 (&raw)[ITEMS]) {
            GcDecodeTile &T = s_tile[tile & 1];
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const int item = hl + DHELPERS * k;
                if (item >= DCW * DTF) continue;
                const int c = item / DTF, j = item - c * DTF;         // consecutive lanes: consecutive frames
                const int fr = tile * DTF + j;
                if (fr >= frames) continue;
                uint64_t bits = ((uint64_t)raw[k].y << 32) | raw[k].x;
                if (fr >= full_frames) {                               // partial last frame: only its bytes exist
                    const int ch = imin(ch0 + c, nch - 1);
                    const uint8_t *src = adpcm + (int64_t)ch * adpcm_pitch + (int64_t)fr * 8;
                    const int nbytes = (tail + 2 + 1) / 2;
                    bits = 0;
                    for (int b = 0; b < nbytes; b++) bits |= (uint64_t)src[b] << (8 * b);
                }
                const int ps = (int)(bits & 0xFF);
                const int scale = (1 << (ps & 0xF)) * 2048;            // :26
                int predictor = (ps >> 4) & 0xF;                       // :27
                if (predictor > 7) { bad = true; predictor &= 7; }
                T.coef[j][c] = make_int2(s_coefs[c * 16 + predictor * 2], s_coefs[c * 16 + predictor * 2 + 1]);
                int d[16];
#pragma unroll
                for (int s = 0; s < 14; s++) {
                    const int byte = (int)((bits >> (8 * (1 + s / 2))) & 0xFF);
                    const int nib = (s & 1) ? (byte & 0xF) : (byte >> 4);
                    d[s] = scale * ((nib ^ 8) - 8) + 1024;             // SignedNibbles (Helpers.cs:50), :36, + the 1024 of :41
                }
                d[14] = d[15] = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) T.dist[j][q][c] = make_int4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
            }
        };
        auto flush = [&](int tile) {
            const GcDecodeTile &T = s_tile[tile & 1];
            for (int item = hl; item < DCW * DTF; item += DHELPERS) {
                const int c = item / DTF, j = item - c * DTF;
                const int fr = tile * DTF + j;
                if (fr >= frames || ch0 + c >= nch) continue;
                const int4 a = T.out[j][0][c], b = T.out[j][1][c];
                const uint32_t w[7] = {(uint32_t)a.x, (uint32_t)a.y, (uint32_t)a.z, (uint32_t)a.w,
                                       (uint32_t)b.x, (uint32_t)b.y, (uint32_t)b.z};
                int16_t *dst = pcm + (int64_t)(ch0 + c) * pcm_pitch + (int64_t)fr * 14;
                if (fr < full_frames) {
                    uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
                    for (int i = 0; i < 7; i++) d32[i] = w[i];
                } else {
                    for (int s = 0; s < tail; s++) dst[s] = (int16_t)(w[s >> 1] >> (16 * (s & 1)));
                }
            }
        };
        uint2 ra[ITEMS], rb[ITEMS];                    // ping-pong: tile t+1 being unpacked, tile t+2 in flight
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
        if (bad && status) atomicOr(status, 1);
        return;
    }

    // ---------------------------------------------------------------- decoder wave: lane = channel
    __builtin_amdgcn_s_setprio(3);
    const int ch = imin(ch0 + tid, nch - 1);
    int h1 = hist1 ? hist1[ch] : 0;
    int h2 = hist2 ? hist2[ch] : 0;
    lds_barrier();                                   // tile 0 prepared
    struct Row { int2 cf; int4 q0, q1, q2, q3; };
    auto read_row = [&](const GcDecodeTile &T, int j, Row &R) {
        R.cf = T.coef[j][tid];
        R.q0 = T.dist[j][0][tid]; R.q1 = T.dist[j][1][tid]; R.q2 = T.dist[j][2][tid]; R.q3 = T.dist[j][3][tid];
    };
    auto decode_frame = [&](GcDecodeTile &T, int j, const Row &R) {
        const int d[14] = {R.q0.x, R.q0.y, R.q0.z, R.q0.w, R.q1.x, R.q1.y, R.q1.z, R.q1.w,
                           R.q2.x, R.q2.y, R.q2.z, R.q2.w, R.q3.x, R.q3.y};
        int o[14];
#pragma unroll
        for (int s = 0; s < 14; s++) {
            // :38-45: (coef1*hist1 + coef2*hist2 + distance + 1024) >> 11, clamped; int32 wrap like the reference.
            // Two mads: the hist2 term is ready one sample early, so the dependent chain is mad, shift, clamp.
            int rest = __mul24(R.cf.y, h2) + d[s];
            asm("" : "+v"(rest));
            const int t = __mul24(R.cf.x, h1) + rest;
            const int v = imin(imax(t >> 11, -32768), 32767);
            h2 = h1;
            h1 = v;
            o[s] = v;
        }
        // a partial last frame decodes all 14 positions here; the flush writes only the valid ones and the
        // history is not used afterwards
        T.out[j][0][tid] = make_int4((o[0] & 0xFFFF) | (o[1] << 16), (o[2] & 0xFFFF) | (o[3] << 16),
                                     (o[4] & 0xFFFF) | (o[5] << 16), (o[6] & 0xFFFF) | (o[7] << 16));
        T.out[j][1][tid] = make_int4((o[8] & 0xFFFF) | (o[9] << 16), (o[10] & 0xFFFF) | (o[11] << 16),
                                     (o[12] & 0xFFFF) | (o[13] << 16), 0);
    };
    for (int tile = 0; tile < tiles; tile++) {
        GcDecodeTile &T = s_tile[tile & 1];
        const int nf = imin(DTF, frames - tile * DTF);
        // two row register sets, ping-pong: the LDS reads of frame j+1 are in flight during frame j
        Row RA, RB;
        read_row(T, 0, RA);
#pragma unroll 1
        for (int j = 0; j < nf; j += 2) {
            read_row(T, imin(j + 1, DTF - 1), RB);
            decode_frame(T, j, RA);
            if (j + 1 < nf) {
                read_row(T, imin(j + 2, DTF - 1), RA);
                decode_frame(T, j + 1, RB);
            }
        }
        lds_barrier();
    }
}

// One frame of GcAdpcmDecoder.Decode (:25-45) from the history (h1, h2) into o[0 .. valid).
__device__ __forceinline__ void gc_decode_frame_serial(const uint8_t *fr, const int (&cf)[16], int valid, int &h1, int &h2,
                                                       int16_t *o)
{
    const int ps = fr[0];
    const int scale = (1 << (ps & 0xF)) * 2048;
    const int predictor = (ps >> 4) & 7;
    int c1 = 0, c2 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (predictor == i) { c1 = cf[2 * i]; c2 = cf[2 * i + 1]; }
    for (int s = 0; s < valid; s++) {
        const int byte = fr[1 + (s >> 1)];
        const int nib = (s & 1) ? (byte & 0xF) : (byte >> 4);
        const int v = imin(imax((c1 * h1 + c2 * h2 + scale * ((nib ^ 8) - 8) + 1024) >> 11, -32768), 32767);
        h2 = h1;
        h1 = v;
        o[s] = (int16_t)v;
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
    int *__restrict__ seam_open, int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    const int k = blockIdx.y + 1;
    const int64_t f0 = (int64_t)k * seg_frames;
    if (ch >= nch || f0 * 14 >= total_samples) return;
    const uint8_t *src = adpcm + (int64_t)ch * adpcm_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
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
    for (int64_t f = f0; f < f0 + seg_frames && f * 14 < total_samples; f++) {
        const int valid = f < full_frames ? 14 : total_samples - (int)(f * 14);
        int16_t *o = dst + f * 14;
        int g1 = 0, g2 = 0;                            // the guessed run's history at this frame's end
        if (valid == 14) { g1 = o[13]; g2 = o[12]; }
        gc_decode_frame_serial(src + f * 8, cf, valid, h1, h2, o);
        if (valid == 14 && h1 == g1 && h2 == g2 && !seam_forced_open(force_open, ch, k)) return;
        if (valid < 14) return;                        // the stream's last, partial frame: nothing follows
    }
    if ((f0 + seg_frames) * 14 < total_samples) {      // open, and a piece follows
        seam_open[(int64_t)(k - 1) * nch + ch] = 1;
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
    const int *__restrict__ first_open, const int *__restrict__ seam_open, int force_open)
{
    const int ch = blockIdx.x * 64 + threadIdx.x;
    if (ch >= nch) return;
    const int k0 = first_open[ch];
    if (k0 <= 0 || k0 >= 0x7f000000) return;
    const uint8_t *src = adpcm + (int64_t)ch * adpcm_pitch;
    int16_t *dst = pcm + (int64_t)ch * pcm_pitch;
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
                if (valid == 14 && h1 == g1 && h2 == g2 && !seam_forced_open(force_open, ch, k)) { apart = false; break; }
            }
        }
        const int64_t f1 = f0 + seg_frames;            // the next piece's first frame
        if (apart) {
            carry = true;                              // (h1, h2): the true samples at the end of this piece
        } else if (flagged && f1 * 14 < total_samples) {
            carry = true;                              // this piece's own seam ran out of frames: the piece is final, its end the truth
            h1 = dst[f1 * 14 - 1];
            h2 = dst[f1 * 14 - 2];
        } else
            carry = false;
    }
}

int launch_decode(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch, int sample_count,
                  const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int64_t pcm_pitch, int *d_status,
                  hipStream_t stream)
{
    if (nch <= 0 || sample_count <= 0) return VGA_OK;
    const size_t lds = 2 * sizeof(GcDecodeTile) + DCW * 16 * sizeof(int16_t);
    VGA_HIP_TRY(allow_dynamic_lds(gc_decode_kernel, lds));
    // As many time segments as fill the device once (one workgroup of this LDS size per CU, 64 channels each), each
    // at least 1024 frames long so that the seams stay a small part of the work
    const int frames = (sample_count + 13) / 14;
    const int groups = (nch + DCW - 1) / DCW;
    const int cus = device_cu_count();
    const int per_cu = std::max(1, (int)((160 * 1024) / lds));
    int segments = cus * per_cu / groups;
    if (segments > frames / 1024) segments = frames / 1024;
    if (segments < 1) segments = 1;
    // every seam is a chance of a run that never meets (see adx_kernels.hip): at most 16 pieces, or ~4000 seams per launch
    const int piece_cap = 4096 / nch > 16 ? (4096 / nch > 64 ? 64 : 4096 / nch) : 16;
    if (segments > piece_cap) segments = piece_cap;
    if (encoder_segments_override() > 0) segments = std::min(std::max(frames / 8, 1), encoder_segments_override());   // test hook
    const int seg_frames = (frames + segments - 1) / segments;
    hipLaunchKernelGGL(gc_decode_kernel, dim3(groups, segments), dim3(DTHREADS), lds, stream, d_adpcm, adpcm_pitch, d_coefs, nch,
                       sample_count, seg_frames, d_hist1, d_hist2, d_pcm, pcm_pitch, d_status);
    VGA_HIP_TRY(hipGetLastError());
    if (segments > 1) {
        AsyncBuf scratch;                              // freed (stream-ordered) on every exit path
        const size_t flag_bytes = (size_t)(segments - 1) * nch * sizeof(int);
        VGA_HIP_TRY(scratch.alloc((size_t)nch * sizeof(int) + flag_bytes, stream));
        int *first_open = scratch.as<int>();
        int *seam_open = first_open + nch;
        VGA_HIP_TRY(hipMemsetAsync(first_open, 0x7f, (size_t)nch * sizeof(int), stream));
        VGA_HIP_TRY(hipMemsetAsync(seam_open, 0, flag_bytes, stream));
        hipLaunchKernelGGL(gc_decode_fixup_kernel, dim3((nch + 63) / 64, segments - 1), dim3(64), 0, stream, d_adpcm, adpcm_pitch,
                           d_coefs, nch, sample_count, seg_frames, d_pcm, pcm_pitch, first_open, seam_open, force_open_seams());
        VGA_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gc_decode_tail_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_adpcm, adpcm_pitch, d_coefs, nch,
                           sample_count, seg_frames, segments, d_pcm, pcm_pitch, first_open, seam_open, force_open_seams());
        VGA_HIP_TRY(hipGetLastError());
    }
    return VGA_OK;
}

}  // namespace gc
}  // namespace vga
