// hca_encode_core.hpp -- what the two CRI HCA encoder kernels share (hca_encode_kernel.hip: a workgroup per run of frames,
// any channel count; hca_encode_wave_kernel.hip: a wave per run of frames, one or two channels): the encoder's LDS tables,
// the rank look-up behind the bit costs of a band (CalculateUsedBits, CriHcaEncoder.cs:554-597), FindScaleFactor
// (:691-709), CalculateResolution (CriHcaPacking.cs:60-69), the DPP reductions and scans of a wave, CRC-16 arithmetic.
#pragma once
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_decode_core.hpp"

namespace vga {
namespace hca {
namespace enc {

constexpr int RS = ROW_BYTES / 8;      // doubles between the rows of the spectra: [channel][sub-frame] rows of 128 + padding

struct EncTab {
    double dequant_scale[64];          // DequantizerScalingTable (FindScaleFactor)
    double quant_scale[64];            // QuantizerScalingTable
    double inv_step[16];               // QuantizerInverseStepSize
    uint8_t enc_pair[8][16];                     // QuantizeSpectrumValue << 4 | QuantizeSpectrumBits (index q + 8)
    uint8_t max_bits[16];
    uint8_t res_curve[64];
};

// CalculateUsedBits (:554-597) needs, per band and for each of the sixteen resolutions, the bits its eight scaled coefficients
// cost.  Resolutions 1..7: the code length of a coefficient depends on its quantised magnitude only and steps up ONCE
// (QuantizeSpectrumBits: 1: |q| >= 1, 2: >= 2, 3: >= 1, 4: >= 4, 5: >= 3, 6: >= 2, 7: >= 1).  The quantiser
// q = (int)(x * inv + up) - down (CriHcaEncoder.cs:589-591) is non-decreasing in x -- a product with a positive constant, a
// sum and a truncation of a positive value are, rounding included -- so "|q| >= k" is exactly "x >= thr_pos or x <= thr_neg"
// for two doubles found by bisection with the quantiser's own arithmetic (threshold_of).  Resolutions 8..15 cost one bit
// more outside the dead zone (|x| >= QuantizerDeadZone, CriHcaTables.cs:68-78).  A band's cost at resolution r is therefore
// a constant plus the NUMBER of its coefficients beyond r's threshold: fifteen thresholds for positive x, fifteen for
// negative x.
//
// Round 5: one coefficient used to be compared with all 22 thresholds (352 f64 compare / carry-add pairs per band).  Sorted,
// the fifteen thresholds of a sign cut the magnitudes into sixteen ranks, and a coefficient's contribution to all sixteen
// costs is a function of (sign, rank): one 128-bit pattern of 0/1 bytes.  The rank comes from the magnitude's own bits:
// exponent + top four mantissa bits name one of ~200 buckets, a byte per bucket says how many thresholds lie at or below
// the bucket's lower edge, and -- at most ONE threshold lies inside a bucket (the closest pair of thresholds is 29 %
// apart, a bucket is at most 6.25 % wide; checked when the table is built) -- one exact f64 compare against that threshold
// decides the rest.  Per coefficient: a byte read, a threshold read, one f64 compare, a pattern read, four adds.
constexpr int COST_BUCKETS = 208;      // 13 octaves below 1.0 x 16
struct CostLut {
    uint4 pat[2][16];                  // [sign][rank]: byte r = 1 if the threshold of resolution r is among the `rank` smallest
    double thr[2][16];                 // [sign][k]: the (k + 1)-th smallest threshold (magnitude); [15] = +inf
    uint8_t rank_base[2][COST_BUCKETS];// thresholds at or below the bucket's lower edge
    uint4 base;                        // byte r: 8 x the code length inside the threshold (QuantizeSpectrumBits / max bits - 1)
    int key_base;                      // bucket = (high dword of |x| >> 16) - key_base, clamped
};

// the magnitude from which resolution r (1..15) costs a coefficient of sign `neg` one bit more
__device__ __forceinline__ double threshold_of(int r, int neg)
{
    if (r >= 8) {                                              // QuantizerDeadZone (CriHcaTables.cs:68-78)
        const double st = f64_bits(HCA_QuantizerStepSizeBits[r]);
        return __longlong_as_double(__double_as_longlong(st / 2) - (long long)(HCA_ResolutionMaxValue[r] + 1));
    }
    const double inv = f64_bits(HCA_QuantizerInverseStepSizeBits[r]);
    const double up = inv + 1;
    const int down = (int)(inv + 0.5 - 8);
    const uint8_t *bits = HCA_QuantizeSpectrumBits[r];
    const int b0 = bits[8];
    int k = 1;
    while (k < 8 && bits[8 + k] == b0) k++;                  // the first magnitude that costs a bit more
    auto index_of = [&](double x) { return (int)(x * inv + up) - down; };
    // ScaleSpectra clamps to +-0.999999999999 (:668): the largest magnitude a coefficient can have
    const long long top = __double_as_longlong(0.999999999999);
    long long lo = 0, hi = top;                               // doubles >= 0 order like their bits
    while (hi - lo > 1) {
        const long long mid = (lo + hi) / 2;
        const double m = __longlong_as_double(mid);
        const bool beyond = neg ? index_of(-m) <= 8 - k : index_of(m) >= 8 + k;
        if (beyond) hi = mid;
        else lo = mid;
    }
    return __longlong_as_double(hi);
}

// Threads 0..127 of the workgroup, once: `tmp` = 64 doubles of scratch LDS.  Returns false (to every thread) when two
// thresholds share a bucket -- the tables would be wrong; never with the reference's constants.
template <int NT>
__device__ __forceinline__ bool cost_lut_build(CostLut &Q, double *tmp, int tid)
{
    double *val = tmp;                                         // [2][16] unsorted: index r - 1
    int *order = reinterpret_cast<int *>(tmp + 32);            // [2][16]: resolution of the k-th smallest threshold
    __shared__ int s_bad;
    if (tid < 30) val[(tid / 15) * 16 + tid % 15] = threshold_of(tid % 15 + 1, tid / 15);
    if (tid == 30) s_bad = 0;
    __syncthreads();
    if (tid < 30) {
        const int sg = tid / 15, i = tid % 15;
        const double v = val[sg * 16 + i];
        int rank = 0;
        for (int j = 0; j < 15; j++) {
            const double w = val[sg * 16 + j];
            rank += (w < v || (w == v && j < i)) ? 1 : 0;
        }
        Q.thr[sg][rank] = v;
        order[sg * 16 + rank] = i + 1;
    } else if (tid < 32) {
        Q.thr[tid - 30][15] = __longlong_as_double(0x7FF0000000000000ll);      // +inf: rank 15 is the last
    }
    __syncthreads();
    if (tid < 32) {                                            // pat[sign][rank]
        const int sg = tid >> 4, k = tid & 15;
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < k; j++) {
            const int r = order[sg * 16 + j];
            w[r >> 2] |= 1u << (8 * (r & 3));
        }
        Q.pat[sg][k] = make_uint4(w[0], w[1], w[2], w[3]);
    } else if (tid == 32) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int r = 1; r < 16; r++) {
            const int len = r < 8 ? HCA_QuantizeSpectrumBits[r][8] : HCA_QuantizedSpectrumMaxBits[r] - 1;
            w[r >> 2] |= (uint32_t)(8 * len) << (8 * (r & 3));
        }
        Q.base = make_uint4(w[0], w[1], w[2], w[3]);
        const double smallest = Q.thr[0][0] < Q.thr[1][0] ? Q.thr[0][0] : Q.thr[1][0];
        Q.key_base = (int)((uint32_t)__double2hiint(smallest) >> 16) - 1;       // bucket 0: everything below every threshold
    }
    __syncthreads();
    const int kb = Q.key_base;
    for (int i = tid; i < 2 * COST_BUCKETS; i += NT) {
        const int sg = i / COST_BUCKETS, b = i % COST_BUCKETS;
        // bucket b holds the magnitudes whose high dword >> 16 is kb + b (b = 0: that and everything below)
        const double lower = b == 0 ? 0.0 : __hiloint2double((kb + b) << 16, 0);
        const double upper = __hiloint2double((kb + b + 1) << 16, 0);
        int at_or_below = 0, inside = 0;
        for (int j = 0; j < 15; j++) {
            const double t = Q.thr[sg][j];
            at_or_below += t <= lower ? 1 : 0;
            inside += (t > lower && t < upper) ? 1 : 0;
        }
        if (inside > 1 || (b == 0 && (at_or_below | inside) != 0)) s_bad = 1;
        Q.rank_base[sg][b] = (uint8_t)at_or_below;
    }
    if (tid == 33) {                                           // the largest magnitude (:668) must have a bucket
        const int top = (int)((uint32_t)__double2hiint(0.999999999999) >> 16) - kb;
        if (top >= COST_BUCKETS) s_bad = 1;
    }
    __syncthreads();
    return s_bad == 0;
}

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// CriHcaEncoder.cs:691-709
__device__ __forceinline__ int find_scale_factor(const EncTab &T, double value)
{
    unsigned low = 0, high = 63;
#pragma unroll
    for (int step = 0; step < 6; step++) {               // 64 entries: six halvings, then low == high
        const unsigned mid = (low + high) / 2;
        const bool up = T.dequant_scale[mid] <= value;
        low = up ? mid + 1 : low;
        high = up ? high : mid;
    }
    return (int)low;
}

// (int)double for values known to be small
__device__ __forceinline__ int trunc_i(double d) { return (int)d; }

// multiply in GF(2)[x] / (x^16 + x^15 + x^2 + 1)
__device__ __forceinline__ unsigned gf_mul(unsigned a, unsigned b)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = ((r << 1) ^ ((r & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        if ((a >> i) & 1u) r ^= b;
    }
    return r;
}

// All sixteen costs of one band (each <= 8 * 12 bits: a byte; no byte can carry) from the look-up described at CostLut.
// (GROUP: look-ups in flight at once -- a kernel short of registers takes them four at a time)
template <int GROUP = 8>
__device__ __forceinline__ uint4 band_cost_table(const CostLut &Q, const double (&x)[8])
{
    uint32_t a0 = Q.base.x, a1 = Q.base.y, a2 = Q.base.z, a3 = Q.base.w;
    const int kb = Q.key_base;
#pragma unroll
    for (int sf = 0; sf < 8; sf++) {
        const uint32_t hi = (uint32_t)__double2hiint(x[sf]);
        const int sg = (int)(hi >> 31);
        const int b = min(max((int)((hi & 0x7FFFFFFFu) >> 16) - kb, 0), COST_BUCKETS - 1);
        const int rb = Q.rank_base[sg][b];
        const int rank = rb + (fabs(x[sf]) >= Q.thr[sg][rb] ? 1 : 0);
        const uint4 p = Q.pat[sg][rank];
        a0 += p.x;
        a1 += p.y;
        a2 += p.z;
        a3 += p.w;
        if (GROUP < 8 && (sf + 1) % GROUP == 0) stage_fence();
    }
    return make_uint4(a0, a1, a2, a3);
}

__device__ __forceinline__ int cost_at(const uint4 &t, int res)
{
    const uint64_t lo = ((uint64_t)t.y << 32) | t.x, hi = ((uint64_t)t.w << 32) | t.z;
    const uint64_t half = res >= 8 ? hi : lo;
    return (int)((half >> (8 * (res & 7))) & 0xFFu);
}

template <class Tab>
__device__ __forceinline__ int resolution_of(const Tab &T, int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int p = noise_level - 5 * scale_factor / 2 + 2;
    p = min(max(p, 0), 58);
    return T.res_curve[p];
}

__device__ __forceinline__ int wave_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);     // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_xor(int v)
{
    v ^= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    v ^= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    return __builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 16) ^ __builtin_amdgcn_readlane(v, 32) ^
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    // Hillis-Steele inside each 16-lane row with row_shr, then the row totals are handed on with row_bcast
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1 (no source lane: + 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// One probe of CalculateUsedBits for the four bands a lane of the searching wave owns: resolution from the noise level
// (CriHcaPacking.CalculateResolution), cost from the band's table (two 64-bit halves: resolutions 0-7, 8-15).
// (a free function, not a lambda: hipcc keeps by-reference captures of register arrays in scratch)
__device__ __forceinline__ int probe_partial(const EncTab &T, const uint64_t (&clo)[4], const uint64_t (&chi)[4],
                                             const int (&off)[4], const int (&bnd)[4], const bool (&on)[4], int noise_level,
                                             int eval_boundary)
{
    int partial = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int noise = bnd[k] < eval_boundary ? noise_level - 1 : noise_level;
        const int res = T.res_curve[min(max(noise + off[k], 0), 58)];
        const uint64_t half = res >= 8 ? chi[k] : clo[k];
        partial += on[k] ? (int)((half >> (8 * (res & 7))) & 0xFFu) : 0;
    }
    return partial;
}

// a lane's codes are consecutive in the stream: they are gathered in a 64-bit window and leave as whole dwords (one
// LDS atomic per dword; the first and last dword of a lane are shared with its neighbours)
struct Emitter {
    unsigned *buf;
    uint64_t acc;
    int word, p;
    __device__ __forceinline__ void put(unsigned value, int nbits)     // p < 32, nbits <= 15
    {
        acc |= (uint64_t)value << (64 - p - nbits);
        p += nbits;
        if (p >= 32) {
            const unsigned hi = (unsigned)(acc >> 32);
            if (hi) atomicOr(&buf[word], hi);
            acc <<= 32;
            word++;
            p -= 32;
        }
    }
    __device__ __forceinline__ void finish()
    {
        const unsigned hi = (unsigned)(acc >> 32);
        if (hi) atomicOr(&buf[word], hi);
    }
};
}  // namespace enc
}  // namespace hca
}  // namespace vga
