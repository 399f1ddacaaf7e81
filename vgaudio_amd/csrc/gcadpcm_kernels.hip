// gcadpcm_kernels.hip -- GC-ADPCM coefficient search for gfx950 (+ the synthetic PCM generator of the benchmark).
//
// Replaces VGAudio/Codecs/GcAdpcm/GcAdpcmCoefficients.cs:9-110 (+ helpers :112-396), bit-exact under RyuJIT x64
// semantics: built with -ffp-contract=off -fwrapv, f64 division/rint are IEEE correctly rounded on gfx950.
// The encoder lives in gc_encode_kernel.hip / gc_encode_core.hpp, the decoder in gc_decode_kernel.hip.
//
// gc_coefs_kernel: one wave per channel.  Frames are lane-parallel for the per-frame LPC records and the
// per-record cluster terms; the f64 bucket sums are accumulated IN RECORD ORDER (ordered_sum) so that every
// rounding matches the reference's loop.
#include "common.hpp"
#include "gcadpcm_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace vga {
namespace gc {

// ---------------------------------------------------------------- coefficients
struct Record {
    double r1, r2;
    bool valid;
};

__device__ __forceinline__ double get3(const double (&v)[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
__device__ __forceinline__ void set3(double (&v)[3], int i, double val)
{
    if (i == 0) v[0] = val; else if (i == 1) v[1] = val; else v[2] = val;
}

// One frame of the analysis loop, GcAdpcmCoefficients.cs:40-61.  x[0..1] are the
// two samples before the frame (the only part of the 14-sample history the
// reference reads), x[2..15] the frame (zero padded).
// the six sums of products a frame needs, as f64 (InnerProductMerge :112-120, OuterProductMerge :122-131)
struct FrameSums {
    double vec[3];
    double m11, m12, m22;
};

__device__ __forceinline__ FrameSums frame_sums(const int (&x)[16])
{
    FrameSums f;
    // 0.0 - p - p ...: every partial is an exact integer < 2^53
#pragma unroll
    for (int i = 0; i <= 2; i++) {
        long long s = 0;
#pragma unroll
        for (int t = 0; t < 14; t++) s -= (long long)(x[2 + t - i] * x[2 + t]);
        f.vec[i] = (double)s;
    }
    long long s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
    for (int z = 0; z < 14; z++) {
        s11 += (long long)(x[2 + z - 1] * x[2 + z - 1]);
        s12 += (long long)(x[2 + z - 1] * x[2 + z - 2]);
        s22 += (long long)(x[2 + z - 2] * x[2 + z - 2]);
    }
    f.m11 = (double)s11; f.m12 = (double)s12; f.m22 = (double)s22;
    return f;
}

// The same six sums from the window as 8 packed pairs w[i] = (x[2i], x[2i+1]): three 7-term dot-product sums
// with v_dot2_i32_i16 and six edge products instead of 70 multiplies with 64-bit adds.
//   S0 = sum_{j=2..15} x[j]^2         = -vec[0]      (pairs 1..7 with themselves)
//   S1 = sum_{j=0..13} x[j] x[j+1]    = m12          (pair i with the pair shifted by one sample)
//   S2 = sum_{j=0..13} x[j] x[j+2]    = -vec[2]      (pair i with pair i+1)
//   -vec[1] = S1 - x0 x1 + x14 x15;  m11 = S0 + x1^2 - x15^2;  m22 = m11 + x0^2 - x14^2
// Every value is an exact integer below 2^53, so the f64 additions are exact in any order.  A pair sum can
// be +2^31 (two products of -32768 * -32768): the accumulator input -1 keeps it inside int32.
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2_less1(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), -1, false);
}
__device__ __forceinline__ FrameSums frame_sums_packed(const uint32_t (&w)[8])
{
    // every term is (pair sum - 1): the seven missing ones are added back at the end (exact integers throughout)
    double s0 = 7.0, s1 = 7.0, s2 = 7.0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        s0 += (double)dot2_less1(w[i + 1], w[i + 1]);
        s1 += (double)dot2_less1(w[i], __builtin_amdgcn_alignbit(w[i + 1], w[i], 16));
        s2 += (double)dot2_less1(w[i], w[i + 1]);
    }
    const int x0 = (int)(int16_t)(w[0] & 0xFFFF), x1 = (int)w[0] >> 16;
    const int x14 = (int)(int16_t)(w[7] & 0xFFFF), x15 = (int)w[7] >> 16;
    FrameSums f;
    f.vec[0] = 0.0 - s0;                               // 0.0 - p - p ... (:115-117): +0.0 for an all-zero frame
    f.vec[1] = 0.0 - ((s1 - (double)(x0 * x1)) + (double)(x14 * x15));
    f.vec[2] = 0.0 - s2;
    f.m12 = s1;
    f.m11 = (s0 + (double)(x1 * x1)) - (double)(x15 * x15);
    f.m22 = (f.m11 + (double)(x0 * x0)) - (double)(x14 * x14);
    return f;
}

__device__ __forceinline__ Record frame_record(const FrameSums &fs)
{
    Record rec;
    rec.valid = false;
    rec.r1 = rec.r2 = 0.0;

    double vec[3] = {fs.vec[0], fs.vec[1], fs.vec[2]};
    if (!(fabs(vec[0]) > 10.0)) return rec;

    double m11 = fs.m11, m12 = fs.m12, m21 = fs.m12, m22 = fs.m22;

    // AnalyzeRanges :133-208 specialised to the 2x2 block (rows/cols 1..2), same
    // operation order, same comparisons (NaN-false semantics preserved).
    int idx1, idx2;
    {
        const double eps = 4.9406564584124654e-324;  // double.Epsilon
        double val = fmax(fabs(m11), fabs(m12));
        if (val < eps) return rec;
        double rc1 = 1.0 / val;
        val = fmax(fabs(m21), fabs(m22));
        if (val < eps) return rec;
        double rc2 = 1.0 / val;

        int max_index = 0;
        double tmp;
        // i = 1
        val = 0.0;
        tmp = fabs(m11) * rc1;
        if (tmp >= val) { val = tmp; max_index = 1; }
        tmp = fabs(m21) * rc2;
        if (tmp >= val) { val = tmp; max_index = 2; }
        if (max_index == 2) {
            double t;
            t = m21; m21 = m11; m11 = t;
            t = m22; m22 = m12; m12 = t;
            rc2 = rc1;
        }
        idx1 = max_index;
        tmp = 1.0 / m11;
        m21 *= tmp;
        // i = 2
        val = 0.0;
        tmp = m22;
        tmp -= m21 * m12;
        m22 = tmp;
        tmp = fabs(tmp) * rc2;
        if (tmp >= val) { val = tmp; max_index = 2; }
        if (max_index == 1) {
            double t;
            t = m11; m11 = m21; m21 = t;
            t = m12; m12 = m22; m22 = t;
            rc1 = rc2;
        }
        idx2 = max_index;

        double mn = 1.0e10, mx = 0.0;
        tmp = fabs(m11);
        if (tmp < mn) mn = tmp;
        if (tmp > mx) mx = tmp;
        tmp = fabs(m22);
        if (tmp < mn) mn = tmp;
        if (tmp > mx) mx = tmp;
        if (mn / mx < 1.0e-10) return rec;
    }

    // BidirectionalFilter :210-237
    {
        double tmp;
        int xx = 0;
        // i = 1
        tmp = get3(vec, idx1);
        set3(vec, idx1, vec[1]);
        if (xx != 0) { /* unreachable at i = 1 */ }
        else if (tmp != 0.0) xx = 1;
        vec[1] = tmp;
        // i = 2
        tmp = get3(vec, idx2);
        set3(vec, idx2, vec[2]);
        if (xx != 0) {
            if (xx == 1) tmp -= vec[1] * m21;   // y = x .. i-1
        } else if (tmp != 0.0) xx = 2;
        vec[2] = tmp;

        // back substitution
        tmp = vec[2];
        vec[2] = tmp / m22;
        tmp = vec[1];
        tmp -= vec[2] * m12;
        vec[1] = tmp / m11;
        vec[0] = 1.0;
    }

    // QuadraticMerge :239-255
    {
        const double v2 = vec[2];
        const double tmp = 1.0 - (v2 * v2);
        if (tmp == 0.0) return rec;
        const double v0 = (vec[0] - (v2 * v2)) / tmp;
        const double v1 = (vec[1] - (vec[1] * v2)) / tmp;
        vec[0] = v0;
        vec[1] = v1;
        if (fabs(v1) > 1.0) return rec;
    }

    // FinishRecord :257-269
#pragma unroll
    for (int z = 1; z <= 2; z++) {
        if (vec[z] >= 1.0) vec[z] = 0.9999999999;
        else if (vec[z] <= -1.0) vec[z] = -0.9999999999;
    }
    rec.r1 = (vec[2] * vec[1]) + vec[1];
    rec.r2 = vec[2];
    rec.valid = true;
    return rec;
}

// MatrixFilter :285-305 -> dst[1], dst[2] (dst[0] == 1.0).  mtx[0][1] and
// mtx[1][2] are written by the reference but never read afterwards.
__device__ __forceinline__ void matrix_filter(double r1, double r2, double &d1, double &d2)
{
    const double m21 = -r1;
    const double m22 = -r2;
    const double val = 1.0 - (m22 * m22);
    const double m11 = ((m22 * m21) + m21) / val;
    d1 = 0.0 + m11 * 1.0;
    d2 = (0.0 + m21 * d1) + m22 * 1.0;
}

// FinishRecord :271-283 (array overload)
__device__ __forceinline__ void finish_record(double (&in_r)[3], double (&out_r)[3])
{
#pragma unroll
    for (int z = 1; z <= 2; z++) {
        if (in_r[z] >= 1.0) in_r[z] = 0.9999999999;
        else if (in_r[z] <= -1.0) in_r[z] = -0.9999999999;
    }
    out_r[0] = 1.0;
    out_r[1] = (in_r[2] * in_r[1]) + in_r[1];
    out_r[2] = in_r[2];
}

// MergeFinishRecord :307-333
__device__ void merge_finish_record(const double (&src)[3], double (&dst)[3])
{
    double tmp[3] = {0.0, 0.0, 0.0};
    double val = src[0];
    dst[0] = 1.0;
    // i = 1
    {
        double v2 = 0.0;
        if (val > 0.0) dst[1] = -(v2 + src[1]) / val;
        else dst[1] = 0.0;
        tmp[1] = dst[1];
        val *= 1.0 - (dst[1] * dst[1]);
    }
    // i = 2
    {
        double v2 = 0.0;
        v2 += dst[1] * src[1];
        if (val > 0.0) dst[2] = -(v2 + src[2]) / val;
        else dst[2] = 0.0;
        tmp[2] = dst[2];
        dst[1] += dst[2] * dst[1];
        val *= 1.0 - (dst[2] * dst[2]);
    }
    finish_record(tmp, dst);
}


__device__ __forceinline__ void load_frame16(const int16_t *src, int f, int length, int (&x)[16])
{
    const int base = f * 14 - 2;
    if (f > 0 && base + 16 <= length) {
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + base);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t w = p32[i];
            x[2 * i] = (int)(int16_t)(w & 0xFFFF);
            x[2 * i + 1] = (int)w >> 16;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int idx = base + i;
            x[i] = (idx >= 0 && idx < length) ? (int)src[idx] : 0;
        }
    }
}

// ---------------------------------------------------------------- coefficients, v2
// One WAVE per channel, no cross-wave synchronisation, high occupancy (the chip's VALU
// throughput needs >= 4 waves per SIMD; tools/ubench_valu.hip).  Per 64-record chunk:
//   1. every lane: record -> (bucket index, d1, d2)            [parallel]
//   2. stable partition of the chunk by bucket (ballot + mbcnt) into one compacted LDS array:
//      bucket b occupies [start_b, start_b + n_b), in record order
//   3. lane (bucket, component) adds ITS bucket's entries in order -> the f64 roundings are
//      the reference's (GcAdpcmCoefficients.cs:67-72, :383-385); the loop runs max_b n_b
//      iterations instead of 64 compare-and-add steps per accumulator.
// MatrixFilter's mtx[1][1] and ContrastVectors' `val` are the same expression
// ((r2*r1 + -r1) / (1 - r2*r2), sign flips are exact), so one f64 divide serves both.
__device__ __forceinline__ int lane_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Exclusive prefix sum over the wave's lanes of a word of 8-bit fields (no field of the result may pass 255): the lane
// totals move up one lane (wave_shr:1), then Hillis-Steele inside each 16-lane row and the row totals handed on.
__device__ __forceinline__ uint32_t scan_fields(uint32_t v)
{
    uint32_t x = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0: + 0)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);            // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);            // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);            // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);            // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);            // row_bcast:15 into rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);            // row_bcast:31 into rows 2, 3
    return x;
}
__device__ __forceinline__ uint64_t scan_fields(uint64_t v)
{
    return (uint64_t)scan_fields((uint32_t)v) | ((uint64_t)scan_fields((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t last_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint64_t last_lane(uint64_t v)
{
    return (uint64_t)last_lane((uint32_t)v) | ((uint64_t)last_lane((uint32_t)(v >> 32)) << 32);
}

// acc += sd[0] + sd[1] + ... + sd[n-1], strictly in that order (the reference's running sums are
// order-dependent f64 adds, GcAdpcmCoefficients.cs:68-71 / :374-380).  sd is the 64-byte aligned start of a
// bucket whose slots n .. ceil8(n)-1 hold +0.0 (adding +0.0 leaves a sum that started at +0.0 unchanged bit
// for bit), so whole batches of 8 are read with b128 loads and added unconditionally.
// trip: wave-uniform upper bound of n.
__device__ __forceinline__ double ordered_sum(double acc, const double *sd, int n, int trip)
{
    for (int i = 0; i < trip; i += 8) {
        if (i < n) {
            const double2 *p = reinterpret_cast<const double2 *>(sd + i);
            const double2 v0 = p[0], v1 = p[1], v2 = p[2], v3 = p[3];
            acc += v0.x; acc += v0.y; acc += v1.x; acc += v1.y;
            acc += v2.x; acc += v2.y; acc += v3.x; acc += v3.y;
        }
    }
    return acc;
}

// The workgroup is ONE wave: its LDS operations execute in program order, so making another lane's
// LDS write visible needs no s_barrier -- only the LDS counter and a compiler ordering point.
// __syncthreads() would also drain vmcnt, i.e. wait for the record prefetch of the NEXT chunk
// (a full HBM round trip per chunk: 48 % of the wave's cycles were SQ_WAIT_ANY, profiles/r01_d_sq_counters.md).
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

#ifdef VGA_DEBUG_TIMESTAMPS
// tools/time_wave_ends.py: start, end of pass 0, end of every gc_coefs_kernel wave (100 MHz wall clock)
__device__ unsigned long long g_vga_coef_ts[3 * 16384];
extern "C" int vga_debug_coefs_timestamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vga_coef_ts), (size_t)n * sizeof(unsigned long long));
}
#endif
// timing-only ablations for tools/build_variants.sh (wrong results): 1 = no ordered sums in the Lloyd passes, 2 = no
// nearest-codeword evaluation, 4 = no partition (every record to bucket 0's slots), 8 = pass 0 without the record
// arithmetic, 16 = pass 0 without its ordered sum, 32 = no Lloyd passes at all
#ifndef VGA_COEFS_ABLATE
#define VGA_COEFS_ABLATE 0
#endif
#ifndef VGA_COEFS_PF_PCM
#define VGA_COEFS_PF_PCM 1
#endif
constexpr int COEF_PF_PCM = VGA_COEFS_PF_PCM;      // chunks of PCM in flight per wave in pass 0
// Lloyd passes: records per chunk of the one-wave-per-channel kernel, in units of 64.  K x 64: every lane classifies K
// records, the stable partition takes the K groups in record order, the ordered sums run over up to 64 K records per
// bucket -- the per-chunk costs (loop, LDS fill, hand-over) are paid once per 64 K records and the sums' batches of eight
// are rounded up once per 64 K records instead of once per 64 (measured at configs[1]: K = 1: 42.2 ms, K = 2: 39.0 ms).
#ifndef VGA_COEFS_CHUNKS
#define VGA_COEFS_CHUNKS 4
#endif
constexpr int COEF_K = VGA_COEFS_CHUNKS;
static_assert(COEF_K >= 1 && COEF_K <= 4 && COEF_RECORD_BLOCK % (64 * COEF_K) == 0, "the partition's 8-bit fields hold 63 K + K - 1");
// Where pass 0 stores record f: inside every block of 64 K records the Lloyd passes' lane l owns records K l .. K l + K - 1
// (lane-major order: one scan ranks the block), and their k-th load reads positions 64 k + l -- coalesced.
__device__ __forceinline__ int record_position(int f)
{
    const int j = f % (64 * COEF_K);
    return f - j + 64 * (j % COEF_K) + j / COEF_K;
}

// Issue priority of the four waves a SIMD holds (round 6).  The arbiter serves the oldest wave first: at equal priorities
// the waves of one SIMD left pass 0 between 6 and 22 ms and ended between 15 and 35 ms, the last one running alone at two
// thirds of the four-wave rate (LABNOTES 8.2 d, profiles/r06_s_coefs_wave_ends.log).  VGA_COEFS_PRIO:
//   0: no priorities                                                              35.1 ms at configs[1]
//   1: priorities rotate with the wall clock (wave slot + epoch of 2^SHIFT x 10 ns)  31.4-31.8 ms
//   3: priority by pass (pass 0: 3, two codewords: 2, four: 1, eight: 0) -- a wave that is ahead yields  30.7-31.0 ms
// Also measured and removed: the ordered sums at priority 3 and everything else at 0 (35.0 against 37.2 before the scan
// partition, does not add to the others); the steps one pass later (3 3 2 2 1 1 0): 31.0; priority from the progress
// through the whole job with steps that shrink towards the end (1/1024 units from the ISA's instruction counts,
// thresholds 440/711/911, 532/798/960, 600/850/975, 300/624/900): 30.5-31.0 -- the waves still end between 22.3 and
// 30.7 ms (p95 28.5), what is left is not the order inside a SIMD (profiles/r06_s_coefs_priority_maps*.log).
#ifndef VGA_COEFS_PRIO
#define VGA_COEFS_PRIO 3
#endif
#ifndef VGA_COEFS_PRIO_SHIFT
#define VGA_COEFS_PRIO_SHIFT 13
#endif
__device__ __forceinline__ void set_priority(uint32_t p)
{
    switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
__device__ __forceinline__ void rotate_priority(uint32_t slot)
{
    if (VGA_COEFS_PRIO == 1) set_priority(slot + (uint32_t)(wall_clock64() >> VGA_COEFS_PRIO_SHIFT));
}
constexpr int COEF_SLOTS = (64 * COEF_K + 56 + 63) / 64 * 64;   // 64 K + 8 x 7 slots of padding, a multiple of 64      // chunks of records in flight per wave in the Lloyd passes

__global__ __launch_bounds__(64) void gc_coefs_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int length,
    double2 *__restrict__ records, int16_t *__restrict__ coefs_out, const Ragged rg)
{
    // compacted (d1, d2) of the current chunk, bucket-major, every bucket starting on a multiple of 8 slots and
    // zero-padded to the next one (<= 64 + 8 * 7 slots); two buffers (chunk parity): one barrier per chunk
    // pass 0: [chunk parity][component][128 slots]; Lloyd passes: [component][COEF_SLOTS slots] (one buffer: a wave's LDS
    // operations execute in program order, the next chunk's fill cannot overtake this chunk's reads)
    constexpr int COEF_BUF = 2 * COEF_SLOTS > 512 ? 2 * COEF_SLOTS : 512;
    __shared__ __align__(64) double s_buf[COEF_BUF];
    auto p0 = [&](int par, int comp) { return s_buf + (2 * par + comp) * 128; };
    auto ll = [&](int comp) { return s_buf + comp * COEF_SLOTS; };
    __shared__ double s_vb[8][3];      // vecBest
    __shared__ double s_cw[8][3];      // val1, val2, val3 of ContrastVectors per codeword
    __shared__ double s_sum[8][3];     // bufferList
    __shared__ int s_cnt[8];           // buffer1

    // ragged batch: workgroup i takes channel order[i] (longest first), with its own length and offsets
    const int ch = rg.order ? rg.order[blockIdx.x] : (int)blockIdx.x;
    if (rg.order) length = rg.length[ch];
    const int lane = threadIdx.x;
    const int16_t *src = pcm + (rg.order ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    const int frames = (length + 13) / 14;
    const int rec_pitch = (int)coef_record_pitch(frames);
    double2 *rec = records + (rg.order ? rg.rec_off[ch] : (int64_t)ch * rec_pitch);
    const int my_bucket = lane >> 1;   // accumulator lanes: lane < 16
    const int my_comp = lane & 1;
    uint32_t wave_slot = 0;
    if (VGA_COEFS_PRIO == 1) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(wave_slot));
    if (VGA_COEFS_PRIO == 3) __builtin_amdgcn_s_setprio(3);
    // Ragged batches: the longest channel is the launch's critical path (120 s among files of 1-120 s: twice the mean wave's
    // work), and by-pass priorities alone would make it yield to every younger wave.  Priority = what the channel still has
    // to do, in quarters of what the longest channel starts with (frames x passes left): 3 3 2 2 1 1 0 for a longest channel,
    // a short file never above 0 or 1.  The mixed-lengths set: 38.6-39.1 ms against 39.7 with the priorities by pass; the
    // channel's length alone, or what is left on a geometric scale (1/2, 1/4, 1/8 of the longest job): 39.0 / 39.4 -- the
    // launch ends when its 120 s files do, ~32 ms even on their own (profiles/r06_z_coefs_wave_ends_ragged*.log).
    const int64_t ragged_job = rg.order ? (int64_t)((rg.max_length + 13) / 14) * 7 : 0;
    auto ragged_priority = [&](int passes_left) __attribute__((always_inline)) {
        const int64_t q = 4 * (int64_t)frames * passes_left / (ragged_job > 0 ? ragged_job : 1);
        set_priority((uint32_t)(q > 3 ? 3 : q));
    };
    if (VGA_COEFS_PRIO == 3 && rg.order) ragged_priority(7);

    // one wave: its LDS operations execute in program order, so the fill needs no barrier before the writes
    auto zero_fill = [&](int par) {
        reinterpret_cast<double2 *>(p0(par, 0))[lane] = make_double2(0.0, 0.0);
        reinterpret_cast<double2 *>(p0(par, 1))[lane] = make_double2(0.0, 0.0);
    };
    auto zero_fill2 = [&](int) {                       // both components' COEF_SLOTS slots: COEF_SLOTS / 64 x 16 bytes per lane
        double2 *z = reinterpret_cast<double2 *>(ll(0));
#pragma unroll
        for (int q = 0; q < COEF_SLOTS / 64; q++) z[q * 64 + lane] = make_double2(0.0, 0.0);
    };

#ifdef VGA_DEBUG_TIMESTAMPS
    if (lane == 0) g_vga_coef_ts[3 * (blockIdx.x & 16383)] = wall_clock64();
#endif
    // ---- pass 0: per-frame records (:40-61) + ordered mean of MatrixFilter outputs (:63-74)
    double acc = 0.0;
    int cnt = 0;
    int par = 0;
    // PCM prefetch, one chunk ahead: frames whose 16-sample window [14 f - 2, 14 f + 14) lies inside the
    // channel are read as 8 dwords with a clamped (always valid) frame index -- unconditional loads, see
    // the record prefetch below; frame 0 and the zero-padded tail take load_frame16()'s slow path.
    const int f_hi = (length - 14) / 14;               // last frame with a full window
    const bool have_interior = f_hi >= 1;
    uint32_t wr[COEF_PF_PCM][8];                       // ring: chunks of PCM in flight
    auto prefetch = [&](int f, uint32_t (&w)[8]) {
        const int fp = min(max(f, 1), max(f_hi, 1));
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + (int64_t)fp * 14 - 2);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = p32[i];
    };
    if (have_interior) {
#pragma unroll
        for (int q = 0; q < COEF_PF_PCM; q++) prefetch(lane + 64 * q, wr[q]);
    }
    auto record_chunk = [&](int base, const uint32_t (&wc)[8]) __attribute__((always_inline)) {
        const int f = base + lane;
        bool valid = false;
        double d1 = 0.0, d2 = 0.0;
        if (f < frames) {
            FrameSums fs;
            if (have_interior && f >= 1 && f <= f_hi) {
                fs = frame_sums_packed(wc);
            } else {                                   // frame 0 / the zero-padded tail
                int x[16];
                load_frame16(src, f, length, x);
                fs = frame_sums(x);
            }
            Record r;
            if (VGA_COEFS_ABLATE & 8) { r.valid = fs.vec[0] != 0.0; r.r1 = fs.m11 * 1e-12; r.r2 = fs.m12 * 1e-12; }
            else r = frame_record(fs);
            valid = r.valid;
            if (valid) { if (VGA_COEFS_ABLATE & 8) { d1 = r.r1; d2 = r.r2; } else matrix_filter(r.r1, r.r2, d1, d2); }
            // The scratch keeps MatrixFilter's output (dst[1], dst[2]), not the record: it is all the later
            // passes need -- ContrastVectors' `val` (:335-342) is the same expression as mtx[1][1] = dst[1]
            // (:295-296) and its second term, -r1*val + -r2, the same as dst[2] (sign flips are exact; a
            // -0.0 / +0.0 difference cannot change a comparison or a sum that started at +0.0).
            rec[record_position(f)] = valid ? make_double2(d1, d2) : make_double2(__builtin_nan(""), 0.0);
        }
        const uint64_t mask = __ballot(valid);
        const int n = __popcll(mask);
        zero_fill(par);
        if (valid) {
            const int slot = lane_rank(mask);
            p0(par, 0)[slot] = d1;
            p0(par, 1)[slot] = d2;
        }
        wave_lds_sync();
        if (lane < 2 && !(VGA_COEFS_ABLATE & 16)) acc = ordered_sum(acc, p0(par, lane), n, n);
        rotate_priority(wave_slot);
        cnt += n;
        par ^= 1;
    };
    for (int base = 0; base < frames; base += 64 * COEF_PF_PCM) {
#pragma unroll
        for (int q = 0; q < COEF_PF_PCM; q++) {
            if (base + 64 * q < frames) {              // wave-uniform
                uint32_t wc[8];
#pragma unroll
                for (int i = 0; i < 8; i++) wc[i] = wr[q][i];
                if (have_interior) prefetch(base + 64 * (q + COEF_PF_PCM) + lane, wr[q]);   // in flight during the next chunks
                record_chunk(base + 64 * q, wc);
            }
        }
    }
#ifdef VGA_DEBUG_TIMESTAMPS
    if (lane == 0) g_vga_coef_ts[3 * (blockIdx.x & 16383) + 1] = wall_clock64();
#endif
    __syncthreads();
    if (lane < 2) s_sum[0][1 + lane] = acc;
    __syncthreads();
    if (lane == 0) {
        double vec1[3];
        vec1[0] = 1.0;
        vec1[1] = s_sum[0][1];
        vec1[2] = s_sum[0][2];
        vec1[1] /= cnt;
        vec1[2] /= cnt;
        double vb[3];
        merge_finish_record(vec1, vb);
        s_vb[0][0] = vb[0]; s_vb[0][1] = vb[1]; s_vb[0][2] = vb[2];
    }
    __syncthreads();

    auto lloyd_iterations = [&](auto exp_c) {
        constexpr int EXP = decltype(exp_c)::value;
        for (int iter = 0; iter < 2; iter++) {
            if (VGA_COEFS_PRIO == 3) {
                if (rg.order) ragged_priority((EXP == 2 ? 6 : (EXP == 4 ? 4 : 2)) - iter);
                else set_priority(EXP == 2 ? 2 : (EXP == 4 ? 1 : 0));
            }
            if (lane < EXP) {
                const double a = s_vb[lane][0], b = s_vb[lane][1], c = s_vb[lane][2];
                s_cw[lane][0] = (a * a) + (b * b) + (c * c);
                s_cw[lane][1] = (a * b) + (b * c);
                s_cw[lane][2] = a * c;
            }
            __syncthreads();
            double cw1[EXP], cw2[EXP], cw3[EXP];
#pragma unroll
            for (int i = 0; i < EXP; i++) {
                cw1[i] = s_cw[i][0];
                cw2[i] = s_cw[i][1];
                cw3[i] = s_cw[i][2];
            }

            acc = 0.0;
            cnt = 0;
            auto classify = [&](int f, const double2 r, bool &valid, int &idx) __attribute__((always_inline)) {
                valid = false;
                idx = 0;
                if (f < frames && r.x == r.x) {
                    valid = true;
                    // r = (dst[1], dst[2]) of MatrixFilter, stored by pass 0 (one f64 divide per frame there, none here)
                    const double val_x2 = 2.0 * r.x, bterm_x2 = 2.0 * r.y;
                    double value = 1.0e30;
                    if (VGA_COEFS_ABLATE & 2) idx = (int)(__double_as_longlong(val_x2) >> 40) & (EXP - 1);
                    else {
#pragma unroll
                    for (int i = 0; i < EXP; i++) {
                        const double t = cw1[i] + (val_x2 * cw2[i]) + (bterm_x2 * cw3[i]);
                        // `if (t < value) { value = t; idx = i; }` (:355-360) with the running minimum as ONE v_min_f64
                        // instead of two selects: value is only ever compared (a -0.0 for a +0.0 changes no decision),
                        // and a NaN t leaves both alone either way
                        idx = t < value ? i : idx;
                        asm("v_min_f64 %0, %1, %2" : "=v"(value) : "v"(value), "v"(t));
                    }
                    }
                }
            };
            // One chunk = 64 K records, record base + K lane + k in r[k] (pass 0 stored them so that this is a coalesced
            // load, see record_position).  Stable partition by bucket: every lane adds a one-hot word of 8-bit fields per
            // record (its own K records in order), ONE exclusive scan over the lanes ranks all 64 K records in all
            // buckets at once (fields <= 63 K + K - 1 <= 255), the last lane's words give the bucket sizes.
            // (round 6; before: a ballot + mbcnt per bucket and group of 64, 6 VALU + 4 SALU x EXP x K per chunk)
            using Fields = std::conditional_t<(EXP > 4), uint64_t, uint32_t>;
            auto field = [](Fields w, int i) __attribute__((always_inline)) { return (int)((w >> (8 * i)) & 0xFF); };
            auto lloyd_chunks = [&](int base, const double2 (&r)[COEF_K]) __attribute__((always_inline)) {
                bool v[COEF_K];
                int ix[COEF_K];
                Fields before[COEF_K], mine = 0;
#pragma unroll
                for (int k = 0; k < COEF_K; k++) {
                    classify(base + COEF_K * lane + k, r[k], v[k], ix[k]);
                    if (VGA_COEFS_ABLATE & 4) ix[k] = 0;
                    before[k] = mine;
                    mine += v[k] ? (Fields)1 << (8 * ix[k]) : (Fields)0;
                }
                const Fields lower = scan_fields(mine);                    // the lanes below this one
                const Fields lower63 = last_lane(lower), mine63 = last_lane(mine);
                int start = 0, max_n = 0;
                Fields starts = 0;                                         // bucket starts / 8
#pragma unroll
                for (int b = 0; b < EXP; b++) {
                    const int n = field(lower63, b) + field(mine63, b);    // (<= 256: two terms, a field holds 255)
                    starts |= (Fields)(start >> 3) << (8 * b);
                    start += (n + 7) & ~7;
                    max_n = max(max_n, n);
                }
                zero_fill2(0);
#pragma unroll
                for (int k = 0; k < COEF_K; k++)
                    if (v[k]) {
                        const int slot = 8 * field(starts, ix[k]) + field(lower + before[k], ix[k]);
                        ll(0)[slot] = r[k].x;
                        ll(1)[slot] = r[k].y;
                    }
                wave_lds_sync();
                        if (lane < 2 * EXP && !(VGA_COEFS_ABLATE & 1)) {
                    const int my_n = field(lower63, my_bucket) + field(mine63, my_bucket);
                    acc = ordered_sum(acc, ll(my_comp) + 8 * field(starts, my_bucket), my_n, max_n);
                    cnt += my_n;
                }
                        rotate_priority(wave_slot);
            };
            // record prefetch, one chunk (K loads) ahead.  The loads are unconditional (every position below rec_pitch
            // exists): a load under a lane-divergent condition makes the compiler wait for it right at the join.
            double2 nxt[COEF_K];
#pragma unroll
            for (int k = 0; k < COEF_K; k++) nxt[k] = rec[lane + 64 * k];
            for (int base = 0; base < frames; base += 64 * COEF_K) {
                double2 cur[COEF_K];
#pragma unroll
                for (int k = 0; k < COEF_K; k++) {
                    cur[k] = nxt[k];
                    nxt[k] = rec[min(base + 64 * (COEF_K + k) + lane, rec_pitch - 1)];    // in flight during this chunk
                }
                lloyd_chunks(base, cur);
            }
            __syncthreads();
            if (lane < 2 * EXP) {
                s_sum[my_bucket][1 + my_comp] = acc;
                if (my_comp == 0) s_cnt[my_bucket] = cnt;
            }
            __syncthreads();
            if (lane < EXP) {
                double bl[3];
                const int n = s_cnt[lane];
                bl[0] = (double)n;                  // bufferList[i][0] sums 1.0 per record
                bl[1] = s_sum[lane][1];
                bl[2] = s_sum[lane][2];
                if (n > 0) { bl[0] /= n; bl[1] /= n; bl[2] /= n; }
                double vb[3] = {s_vb[lane][0], s_vb[lane][1], s_vb[lane][2]};
                merge_finish_record(bl, vb);
                s_vb[lane][0] = vb[0]; s_vb[lane][1] = vb[1]; s_vb[lane][2] = vb[2];
            }
            __syncthreads();
        }
    };

    // ---- 3 splits x 2 Lloyd iterations (:77-91, FilterRecords :344-396)
    for (int w = 0; w < ((VGA_COEFS_ABLATE & 32) ? 0 : 3); w++) {
        const int half = 1 << w;
        if (lane < half) {
            s_vb[half + lane][0] = (0.01 * 0.0) + s_vb[lane][0];
            s_vb[half + lane][1] = (0.01 * -1.0) + s_vb[lane][1];
            s_vb[half + lane][2] = (0.01 * 0.0) + s_vb[lane][2];
        }
        __syncthreads();

        // the two Lloyd iterations of this split, specialised on the codebook size (no per-codeword branches)
        if (w == 0) lloyd_iterations(std::integral_constant<int, 2>{});
        else if (w == 1) lloyd_iterations(std::integral_constant<int, 4>{});
        else lloyd_iterations(std::integral_constant<int, 8>{});
    }

    // ---- output :94-108
    if (lane < 16) {
        const int z = lane >> 1;
        const double d = -s_vb[z][1 + (lane & 1)] * 2048.0;
        int out;
        if (d > 0.0) out = (d > 32767.0) ? 32767 : (int)__builtin_rint(d);
        else out = (d < -32768.0) ? -32768 : ((d != d) ? 0 : (int)__builtin_rint(d));
        coefs_out[ch * 16 + lane] = (int16_t)out;
    }
#ifdef VGA_DEBUG_TIMESTAMPS
    if (lane == 0) g_vga_coef_ts[3 * (blockIdx.x & 16383) + 2] = wall_clock64();
#endif
}

// ---------------------------------------------------------------- coefficients, v3: four channels + a summing wave
// (used below ~3600 channels, see launch_coefs)
// The ordered bucket sums are a chain of dependent f64 adds per (bucket, component): in the one-wave-per-channel kernel
// above they occupy a whole wave-instruction for 2..16 active lanes, a quarter of that kernel's instructions.  Here a
// workgroup is four RECORD waves (one channel each: records, nearest codeword, stable partition into LDS, as above) and
// one SUMMING wave whose lanes are (channel, bucket, component) -- 4 x 8 x 2 = 64 -- so that one add instruction carries
// the chains of all four channels.  The summing wave works on chunk c - 1 while the record waves prepare chunk c
// (double-buffered LDS, one LDS-only barrier per chunk); it also owns the per-pass codebook updates.  Measured at
// 4096 x 60 s: 14.0 G VALU wave-instructions instead of 16.3 G, but the five waves of a workgroup move in lockstep and
// wait more (SQ_WAIT_ANY 40 % of the wave-cycles instead of 31 %): 43.0 ms against 41.6.
//
// SOLO (round 3): the same five waves on ONE channel.  One wave per channel makes a lone 60 s channel wait 15.6 ms for its
// coefficients (205 714 frames x seven passes down one wave's dependent chain) -- the latency floor of every small batch
// and of the host pipeline's last chunk.  In this mode record wave w takes chunks 4 r + w of round r, and the summing
// wave adds the four chunks of the previous round IN CHUNK ORDER on its first sixteen lanes (bucket, component): the
// order of every f64 addition is the reference's, only the records / nearest-codeword work runs four chunks wide.
constexpr int COEF_CW = 4;                                  // record waves per workgroup (= channels unless SOLO)
constexpr int COEF_THREADS = (COEF_CW + 1) * 64;

template <bool SOLO>
__global__ __launch_bounds__(COEF_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) void gc_coefs_kernel4(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int nch, int length,
    double2 *__restrict__ records, int16_t *__restrict__ coefs_out, const Ragged rg)
{
    __shared__ __align__(64) double s_d[COEF_CW][2][2][128];       // [channel][chunk parity][component][compacted slot]
    __shared__ int s_meta[2][COEF_CW][8];                          // (start << 8) | n of every bucket
    __shared__ int s_maxn[2][COEF_CW];                             // max_b n_b: the summing loop's trip bound
    __shared__ double s_vb[COEF_CW][8][3];                         // vecBest
    __shared__ double s_cw[COEF_CW][8][3];                         // ContrastVectors terms per codeword

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool summer = wave == COEF_CW;
    const int ws = summer ? 0 : wave;                              // this record wave's LDS slot (chunk hand-over)
    const int cs = SOLO ? 0 : ws;                                  // ... and its channel slot (codebooks)
    const int ch_raw = SOLO ? (int)blockIdx.x : (int)blockIdx.x * COEF_CW + cs;
    const bool live = !summer && ch_raw < nch;
    // ragged batch (SOLO only): workgroup i takes channel order[i] with its own length and offsets
    const bool ragged = SOLO && rg.order != nullptr;
    const int ch = ragged ? rg.order[blockIdx.x] : (ch_raw < nch ? ch_raw : nch - 1);
    if (ragged) length = rg.length[ch];
    const int16_t *src = pcm + (ragged ? rg.pcm_off[ch] : (int64_t)ch * pcm_pitch);
    const int frames = (length + 13) / 14;
    const int chunks = (frames + 63) / 64;
    // a record wave's chunks: c = round (four channels), c = 4 round + wave (SOLO)
    const int rounds = SOLO ? (chunks + COEF_CW - 1) / COEF_CW : chunks;
    const int cstep = SOLO ? COEF_CW : 1, cfirst = SOLO ? ws : 0;
    double2 *rec = records + (ragged ? rg.rec_off[ch] : (int64_t)ch * frames);
    // summing-wave lane = (channel slot, bucket, component); SOLO: the sixteen lanes of slot 0 carry the one channel
    const int sc = lane >> 4, sb = (lane >> 1) & 7, sk = lane & 1;

    auto zero_fill = [&](int par) {
        reinterpret_cast<double2 *>(&s_d[ws][par][0][0])[lane] = make_double2(0.0, 0.0);
        reinterpret_cast<double2 *>(&s_d[ws][par][1][0])[lane] = make_double2(0.0, 0.0);
    };
    double acc = 0.0;                                              // summing wave: this lane's chain
    int cnt = 0;
    // the summing wave's share of one round: buckets < nb take part
    auto sum_chunk = [&](int par, int nb) {
        if (SOLO) {
#pragma unroll
            for (int w4 = 0; w4 < COEF_CW; w4++) {                 // the round's four chunks, in chunk order
                const int meta = s_meta[par][w4][sb];
                const int n = (sc == 0 && sb < nb) ? (meta & 0xFF) : 0;
                // (round 6: the next batch's LDS reads issued before this batch's eight adds -- 9.2 against 8.75 ms for one
                // channel, 17.4 / 14.9 for 768; four waves per SIMD instead of five: 768 channels 23 ms.
                // profiles/r06_z_coefs_solo_variants.log)
                acc = ordered_sum(acc, &s_d[w4][par][sk][meta >> 8], n, s_maxn[par][w4]);
                cnt += n;
            }
        } else {
            const int meta = s_meta[par][sc][sb];
            const int trip = max(max(s_maxn[par][0], s_maxn[par][1]), max(s_maxn[par][2], s_maxn[par][3]));
            const int n = sb < nb ? (meta & 0xFF) : 0;
            acc = ordered_sum(acc, &s_d[sc][par][sk][meta >> 8], n, trip);
            cnt += n;
        }
    };

    // ---- pass 0: per-frame records (:40-61) + ordered mean of MatrixFilter outputs (:63-74)
    const int f_hi = (length - 14) / 14;
    const bool have_interior = f_hi >= 1;
    uint32_t w[8];
    auto prefetch = [&](int f) {
        const int fp = min(max(f, 1), max(f_hi, 1));
        const uint32_t *p32 = reinterpret_cast<const uint32_t *>(src + (int64_t)fp * 14 - 2);
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = p32[i];
    };
    if (!summer && have_interior) prefetch(cfirst * 64 + lane);
    for (int rd = 0; rd < rounds; rd++) {
        const int par = rd & 1;
        const int c = rd * cstep + cfirst;
        if (!summer) {
            const int f = c * 64 + lane;
            bool valid = false;
            double d1 = 0.0, d2 = 0.0;
            uint32_t wc[8];
#pragma unroll
            for (int i = 0; i < 8; i++) wc[i] = w[i];
            if (have_interior) prefetch(f + 64 * cstep);
            if (f < frames) {
                FrameSums fs;
                if (have_interior && f >= 1 && f <= f_hi) {
                    fs = frame_sums_packed(wc);
                } else {
                    int x[16];
                    load_frame16(src, f, length, x);
                    fs = frame_sums(x);
                }
                const Record r = frame_record(fs);
                valid = r.valid && live;
                if (valid) matrix_filter(r.r1, r.r2, d1, d2);
                if (live) rec[f] = valid ? make_double2(d1, d2) : make_double2(__builtin_nan(""), 0.0);
            }
            const uint64_t mask = __ballot(valid);
            const int n = __popcll(mask);
            zero_fill(par);
            if (valid) {
                const int slot = lane_rank(mask);
                s_d[ws][par][0][slot] = d1;
                s_d[ws][par][1][slot] = d2;
            }
            if (lane < 8) s_meta[par][ws][lane] = lane == 0 ? n : 0;
            if (lane == 8) s_maxn[par][ws] = n;
        } else if (rd > 0) {
            sum_chunk(par ^ 1, 1);
        }
        lds_barrier();
    }
    if (summer) {
        if (rounds > 0) sum_chunk((rounds - 1) & 1, 1);
        const double other = __shfl_xor(acc, 1);
        if (sb == 0 && sk == 0) {
            double vec1[3];
            vec1[0] = 1.0;
            vec1[1] = acc;
            vec1[2] = other;
            vec1[1] /= cnt;
            vec1[2] /= cnt;
            double vb[3];
            merge_finish_record(vec1, vb);
            s_vb[sc][0][0] = vb[0]; s_vb[sc][0][1] = vb[1]; s_vb[sc][0][2] = vb[2];
        }
    }
    lds_barrier();

    auto lloyd_iterations = [&](auto exp_c) {
        constexpr int EXP = decltype(exp_c)::value;
        for (int iter = 0; iter < 2; iter++) {
            if (!summer && lane < EXP && (!SOLO || wave == 0)) {
                const double a = s_vb[cs][lane][0], b = s_vb[cs][lane][1], c3 = s_vb[cs][lane][2];
                s_cw[cs][lane][0] = (a * a) + (b * b) + (c3 * c3);
                s_cw[cs][lane][1] = (a * b) + (b * c3);
                s_cw[cs][lane][2] = a * c3;
            }
            lds_barrier();
            double cw1[EXP], cw2[EXP], cw3[EXP];
#pragma unroll
            for (int i = 0; i < EXP; i++) {
                cw1[i] = s_cw[cs][i][0];
                cw2[i] = s_cw[cs][i][1];
                cw3[i] = s_cw[cs][i][2];
            }
            acc = 0.0;
            cnt = 0;
            double2 r_next = make_double2(0.0, 0.0);
            if (!summer && frames > 0) r_next = rec[min(cfirst * 64 + lane, frames - 1)];
            for (int rd = 0; rd < rounds; rd++) {
                const int par = rd & 1;
                const int c = rd * cstep + cfirst;
                if (!summer) {
                    const int f = c * 64 + lane;
                    bool valid = false;
                    int idx = 0;
                    double d1 = 0.0, d2 = 0.0;
                    const double2 r = r_next;
                    r_next = rec[min(f + 64 * cstep, frames - 1)];   // in flight during this chunk
                    if (live && f < frames && r.x == r.x) {
                        valid = true;
                        const double val_x2 = 2.0 * r.x, bterm_x2 = 2.0 * r.y;
                        double value = 1.0e30;
#pragma unroll
                        for (int i = 0; i < EXP; i++) {
                            const double t = cw1[i] + (val_x2 * cw2[i]) + (bterm_x2 * cw3[i]);
                            idx = t < value ? i : idx;                       // (one v_min_f64: see gc_coefs_kernel's classify)
                            asm("v_min_f64 %0, %1, %2" : "=v"(value) : "v"(value), "v"(t));
                        }
                        d1 = r.x;
                        d2 = r.y;
                    }
                    // stable partition by bucket; lane b keeps bucket b's (start, n) for the summing wave
                    int slot = 0, my_meta = 0, start = 0, max_n = 0;
#pragma unroll
                    for (int b = 0; b < EXP; b++) {
                        const bool mine = valid && idx == b;
                        const uint64_t m = __ballot(mine);
                        const int n_b = __popcll(m);
                        if (mine) slot = start + lane_rank(m);
                        if (lane == b) my_meta = (start << 8) | n_b;
                        start += (n_b + 7) & ~7;
                        max_n = max(max_n, n_b);
                    }
                    zero_fill(par);
                    if (valid) {
                        s_d[ws][par][0][slot] = d1;
                        s_d[ws][par][1][slot] = d2;
                    }
                    if (lane < 8) s_meta[par][ws][lane] = my_meta;
                    if (lane == 8) s_maxn[par][ws] = max_n;
                } else if (rd > 0) {
                    sum_chunk(par ^ 1, EXP);
                }
                lds_barrier();
            }
            if (summer) {
                if (rounds > 0) sum_chunk((rounds - 1) & 1, EXP);
                const double other = __shfl_xor(acc, 1);
                if (sb < EXP && sk == 0) {
                    double bl[3];
                    const int n = cnt;
                    bl[0] = (double)n;                  // bufferList[i][0] sums 1.0 per record
                    bl[1] = acc;
                    bl[2] = other;
                    if (n > 0) { bl[0] /= n; bl[1] /= n; bl[2] /= n; }
                    double vb[3] = {s_vb[sc][sb][0], s_vb[sc][sb][1], s_vb[sc][sb][2]};
                    merge_finish_record(bl, vb);
                    s_vb[sc][sb][0] = vb[0]; s_vb[sc][sb][1] = vb[1]; s_vb[sc][sb][2] = vb[2];
                }
            }
            lds_barrier();
        }
    };

    // ---- 3 splits x 2 Lloyd iterations (:77-91, FilterRecords :344-396)
    for (int wsplit = 0; wsplit < 3; wsplit++) {
        const int half = 1 << wsplit;
        if (!summer && lane < half && (!SOLO || wave == 0)) {
            s_vb[cs][half + lane][0] = (0.01 * 0.0) + s_vb[cs][lane][0];
            s_vb[cs][half + lane][1] = (0.01 * -1.0) + s_vb[cs][lane][1];
            s_vb[cs][half + lane][2] = (0.01 * 0.0) + s_vb[cs][lane][2];
        }
        lds_barrier();
        if (wsplit == 0) lloyd_iterations(std::integral_constant<int, 2>{});
        else if (wsplit == 1) lloyd_iterations(std::integral_constant<int, 4>{});
        else lloyd_iterations(std::integral_constant<int, 8>{});
    }

    // ---- output :94-108
    if (live && lane < 16 && (!SOLO || wave == 0)) {
        const int z = lane >> 1;
        const double d = -s_vb[cs][z][1 + (lane & 1)] * 2048.0;
        int out;
        if (d > 0.0) out = (d > 32767.0) ? 32767 : (int)__builtin_rint(d);
        else out = (d < -32768.0) ? -32768 : ((d != d) ? 0 : (int)__builtin_rint(d));
        coefs_out[ch * 16 + lane] = (int16_t)out;
    }
}

// ---------------------------------------------------------------- synthetic PCM (synth.py)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ long long tri(uint32_t phase)
{
    const long long q = (long long)(phase >> 15);
    return q < 65536 ? q - 32768 : 98303 - q;
}

__global__ __launch_bounds__(256) void synth_kernel(int16_t *__restrict__ pcm, int64_t pitch, int nch, int length,
                                                    int first_channel, const uint32_t *__restrict__ params)
{
    const int k = blockIdx.y;
    const int c = first_channel + k;
    const uint32_t f_inc = params[4 * k + 0], phi = params[4 * k + 1], amp = params[4 * k + 2], lfo = params[4 * k + 3];
    const uint64_t base = ((((uint64_t)0x5EED) << 32) ^ (uint64_t)c) * 0x100000001B3ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < length; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t i32 = (uint32_t)i;
        long long s = ((tri(i32 * f_inc) * (long long)amp) >> 15) + ((tri(i32 * (3u * f_inc) + phi) * (long long)(amp / 3)) >> 15);
        long long nz = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint64_t hh = splitmix64(base ^ ((uint64_t)i - (uint64_t)d));
            nz += (long long)(hh & 4095) - 2048;
        }
        s += nz >> 2;
        const long long env = 20480 + ((tri(i32 * lfo) * 12287) >> 15);
        s = (s * env) >> 15;
        s = s < -32768 ? -32768 : (s > 32767 ? 32767 : s);
        pcm[(int64_t)k * pitch + i] = (int16_t)s;
    }
}

// ---------------------------------------------------------------- launchers
// five waves on ONE channel (gc_coefs_kernel4<true>) up to three workgroups per CU: 768 channels 15.0 against 18.5 ms for
// one wave per channel, 896: 22.7 against 18.3 (profiles/r06_u_coefs_variants_small.log)
constexpr int SOLO_WORKGROUPS_PER_CU = 3;

// How many of a ragged batch's channels (in its longest-first order) get five waves each (gc_coefs_kernel4<true>): all that hold
// at least one chunk of records, or none.  A ragged launch of one-wave workgroups lasts as long as its longest channel --
// 29-33 ms for a 120 s file from 16 to 4096 files of 1-120 s -- until the chip's throughput takes over (10 008 files: 39 ms);
// five waves per channel cut that chain to 20 ms and cost 2.1x the chip time per frame: 16 files 19.9 against 29.3 ms, 768:
// 22.2 / 30.8, 2304: 23.9 / 32.7, 4096: 27.7 / 32.1 (profiles/r06_z_coefs_ragged_small.log).  The choice compares the two
// estimates max(longest channel's chain, frames / throughput) with the rates of those runs.  Host arithmetic only
// (capi_gcadpcm_v.hip calls it when it builds the batch's tables).
// Round 6 also tried the five-wave form for a big batch's LONGEST channels only, on a side stream next to the one-wave launch
// of the rest: a five-wave workgroup next to 4096 one-wave workgroups takes 48-50 ms for a 120 s channel at any count from 64
// to 512, stream and wave priorities at their highest (17 ms alone) -- its five waves meet at a barrier per chunk and each
// shares its SIMD (profiles/r06_z_ragged_coefs_kernel_timeline.log).  Not kept.
int ragged_solo_count(const int *lengths_longest_first, int nch, int64_t total_frames, int cus, int *usable_out)
{
    const int min_length = 14 * 64;                                 // (at least one chunk of records)
    int usable = 0;
    while (usable < nch && lengths_longest_first[usable] >= min_length) usable++;
    if (usable_out) *usable_out = usable;
    if (nch <= 0 || usable == 0) return 0;
    const double scale = 256.0 / (double)(cus > 0 ? cus : 256);     // (the rates below are an MI355X's: 256 CUs)
    const double longest = (double)((lengths_longest_first[0] + 13) / 14), total = (double)total_frames;
    const double five_waves = std::max(longest * 4.8e-5, total / 1.27e7 * scale);      // ms
    const double one_wave = std::max(longest * 7.1e-5, total / 2.16e7 * scale);
    return five_waves < one_wave ? usable : 0;
}

int launch_coefs(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int16_t *d_coefs,
                 void *d_workspace, hipStream_t stream, const Ragged *rg)
{
    if (nch <= 0) return VGA_OK;
    if (rg) {
        // ragged: one wave per channel, longest channel first (the workgroup forms of four channels share one `length`); a
        // batch of at most three workgroups per CU: five waves on every channel that holds a chunk of records, as for a
        // uniform batch (rg->solo_channels leading slots of the order, ragged_solo_count)
        const int solo = coefs_kernel_variant() == 1 ? 0 : (coefs_kernel_variant() == 3 ? rg->solo_usable : rg->solo_channels);
        Ragged rest = *rg;
        rest.order += solo;
        if (solo > 0)
            hipLaunchKernelGGL(gc_coefs_kernel4<true>, dim3(solo), dim3(COEF_THREADS), 0, stream, d_pcm, (int64_t)0, solo, 0,
                               reinterpret_cast<double2 *>(d_workspace), d_coefs, *rg);
        if (solo < nch)
            hipLaunchKernelGGL(gc_coefs_kernel, dim3(nch - solo), dim3(64), 0, stream, d_pcm, (int64_t)0, nch, 0,
                               reinterpret_cast<double2 *>(d_workspace), d_coefs, rest);
        VGA_HIP_TRY(hipGetLastError());
        return VGA_OK;
    }
    // Rounds 2-5: workgroups of four channels + a summing wave (gc_coefs_kernel4<false>) between the two, because a chunk
    // costs them max(records, sums) instead of the sum.  Since round 6 (scan partition, priorities by pass) one wave per
    // channel is faster at every size: 4096 channels 31.6 against 43.0 ms, 3072: 26.3 / 30.5, 1024: 18.6 / 18.9, 128:
    // 16.6 / 18.4 (profiles/r06_u_coefs_variants.log); the four-channel form stays behind the test hook (variant 2).
    const int variant = coefs_kernel_variant();
    const bool solo = variant == 3 || (variant == 0 && nch <= device_cu_count() * SOLO_WORKGROUPS_PER_CU);
    const bool per_channel = variant == 1 || (variant == 0 && !solo);
    if (per_channel)
        hipLaunchKernelGGL(gc_coefs_kernel, dim3(nch), dim3(64), 0, stream, d_pcm, pcm_pitch, nch, length,
                           reinterpret_cast<double2 *>(d_workspace), d_coefs, Ragged{});
    else if (solo)
        hipLaunchKernelGGL(gc_coefs_kernel4<true>, dim3(nch), dim3(COEF_THREADS), 0, stream, d_pcm, pcm_pitch, nch, length,
                           reinterpret_cast<double2 *>(d_workspace), d_coefs, Ragged{});
    else
        hipLaunchKernelGGL(gc_coefs_kernel4<false>, dim3((nch + COEF_CW - 1) / COEF_CW), dim3(COEF_THREADS), 0, stream, d_pcm, pcm_pitch,
                           nch, length, reinterpret_cast<double2 *>(d_workspace), d_coefs, Ragged{});
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_synth(int16_t *d_pcm, int64_t pitch, int nch, int length, int first_channel, const uint32_t *d_params,
                 hipStream_t stream)
{
    if (nch <= 0 || length <= 0) return VGA_OK;
    int gx = (int)((length + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(synth_kernel, dim3(gx, nch), dim3(256), 0, stream, d_pcm, pitch, nch, length, first_channel,
                       d_params);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace gc
}  // namespace vga
