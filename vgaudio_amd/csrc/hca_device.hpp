// hca_device.hpp -- device-side tables and the exact 128-point DCT-IV shared by the HCA kernels.
//
// Tables: generated data header (tools/gen_hca_tables.py <- tests/golden/hca_tables.json), the values
// the reference's own tests pin (CriHcaTableTests.cs:8-115, MdctTests.cs:18-59); f64 entries are IEEE
// bit patterns.  DCT-IV: the staged butterflies of VGAudio/Utilities/Mdct.cs:126-181 in the same
// operation order (no FMA contraction), so spectra are bit-identical to the reference's f64 results.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#define HCA_TABLE_QUAL __device__ const
#include "hca_tables_data.h"
#include "hca_info.hpp"

namespace vga {
namespace hca {

// The encoder's input stream as CriHcaEncoder.Encode assembles it in its 1024-sample buffer
// (VGAudio/Codecs/CriHca/CriHcaEncoder.cs:170-254): [frames of the cleared buffer][copies of sample 0]
// [the PCM up to HcaInfo.SampleCount][loop audio replayed from the loop start][zeros].  Stream index u
// (0 = first sample of frame 0's buffer) -> raw sample index or "zero".
struct PcmMap {
    int zero_pre;    // u < zero_pre: whole pre-audio frames, encoded from the cleared buffer (:175-180)
    int pre_end;     // u < pre_end: pcm[0] (:182-188); pre_end = InsertedSamples - 128
    int main_end;    // u < main_end: pcm[u - pre_end] (:192-207)
    int post_end;    // u < post_end: _postAudio[u - main_end] (:209-232), else 0 (:234-240)
    int loop_start;  // raw index _postAudio was saved from (SaveLoopAudio :244-254)
    int last_chunk;  // index of the last 1024-sample chunk CriHcaFormat.EncodeFromPcm16 feeds (:52-68)
    int raw_len;     // samples per channel in the caller's PCM
};

__device__ __forceinline__ int16_t fetch_pcm(const PcmMap &m, const int16_t *__restrict__ src, int64_t u)
{
    if (u < m.zero_pre || u >= m.post_end) return 0;      // also u < 0: the MDCT starts from a cleared state
    if (u < m.pre_end) return m.raw_len > 0 ? src[0] : (int16_t)0;
    if (u < m.main_end) return src[u - m.pre_end];
    // _postAudio[k] was copied from the caller's chunk buffer while chunks 0..last_chunk went by; a chunk
    // the PCM does not fill keeps the previous chunk's samples in its tail (the reference reuses one
    // buffer, CriHcaFormat.cs:50-56), and what no chunk reached stays 0
    const int a = m.loop_start + (int)(u - m.main_end);
    const int chunk = a >> 10;
    if (a < 0 || chunk > m.last_chunk) return 0;
    if (a < m.raw_len) return src[a];
    return chunk >= 1 ? src[a - 1024] : (int16_t)0;
}

__device__ __forceinline__ double f64_bits(uint64_t b) { return __longlong_as_double((long long)b); }

// sqrt(2.0 / 128) (CriHcaChannel.cs:19): exactly 0.125
constexpr double MDCT_SCALE = 0.125;

// The hot tables, copied into LDS once per workgroup (a table lookup from HBM-backed constant data
// costs an L1/L2 round trip per dependent access; from LDS ~64 cycles).  LDS size sets the occupancy of the HCA
// kernels, so each side carries only what it reads (ENC / DEC) and the trig tables only the entries the transform
// uses: [0, 63) = the stage tables of sizes 1..32 (size 2^b starts at 2^b - 1), [63, 127) = the first half of the
// size-128 table (the pre-rotation).
template <bool ENC, bool DEC>
struct LdsTablesT {
    double sin_t[127], cos_t[127];
    double window[128];
    double dequant_scale[64];                       // DequantizerScalingTable (encoder: FindScaleFactor)
    double quant_scale[ENC ? 64 : 1];               // QuantizerScalingTable
    double inv_step[ENC ? 16 : 1];                  // QuantizerInverseStepSize
    double step[DEC ? 16 : 1];                      // QuantizerStepSize
    double dead_zone[ENC ? 16 : 1];                 // QuantizerDeadZone (CriHcaTables.cs:68-78)
    uint8_t shuffle[128];
    uint8_t enc_bits[ENC ? 8 : 1][16], enc_value[ENC ? 8 : 1][16];   // QuantizeSpectrumBits / Value   (index q+8)
    uint8_t dec_bits[DEC ? 8 : 1][16];                               // QuantizedSpectrumBits          (index code)
    int8_t dec_value[DEC ? 8 : 1][16];                               // QuantizedSpectrumValue
    uint8_t max_bits[16];
    uint8_t res_curve[64];                          // ScaleToResolutionCurve (both sides: CalculateResolution)
};
using EncTables = LdsTablesT<true, false>;
using DecTables = LdsTablesT<false, true>;

template <bool ENC, bool DEC>
__device__ __forceinline__ void load_tables(LdsTablesT<ENC, DEC> &t, int tid, int nthreads)
{
    for (int i = tid; i < 127; i += nthreads) {
        const int src = i < 63 ? i : i + 64;        // 63.. -> entries 127.. of the reference's table
        t.sin_t[i] = f64_bits(MDCT_SinBits[src]);
        t.cos_t[i] = f64_bits(MDCT_CosBits[src]);
    }
    for (int i = tid; i < 128; i += nthreads) {
        t.window[i] = (double)__uint_as_float(HCA_MdctWindowF32Bits[i]);
        t.shuffle[i] = MDCT_Shuffle128[i];
        if constexpr (ENC) {
            (&t.enc_bits[0][0])[i] = (&HCA_QuantizeSpectrumBits[0][0])[i];
            (&t.enc_value[0][0])[i] = (&HCA_QuantizeSpectrumValue[0][0])[i];
        }
        if constexpr (DEC) {
            (&t.dec_bits[0][0])[i] = (&HCA_QuantizedSpectrumBits[0][0])[i];
            (&t.dec_value[0][0])[i] = (&HCA_QuantizedSpectrumValue[0][0])[i];
        }
    }
    for (int i = tid; i < 64; i += nthreads) {
        t.dequant_scale[i] = f64_bits(HCA_DequantizerScalingTableBits[i]);
        t.res_curve[i] = i < 59 ? HCA_ScaleToResolutionCurve[i] : 0;
        if constexpr (ENC) t.quant_scale[i] = f64_bits(HCA_QuantizerScalingTableBits[i]);
    }
    for (int i = tid; i < 16; i += nthreads) {
        const double st = f64_bits(HCA_QuantizerStepSizeBits[i]);
        if constexpr (ENC) {
            t.inv_step[i] = f64_bits(HCA_QuantizerInverseStepSizeBits[i]);
            t.dead_zone[i] = __longlong_as_double(__double_as_longlong(st / 2) - (long long)(HCA_ResolutionMaxValue[i] + 1));
        }
        if constexpr (DEC) t.step[i] = st;
        t.max_bits[i] = HCA_QuantizedSpectrumMaxBits[i];
    }
}

// CriHcaPacking.cs:60-69
template <class Tables>
__device__ __forceinline__ int calculate_resolution(const Tables &t, int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int curve_position = noise_level - 5 * scale_factor / 2 + 2;
    curve_position = min(max(curve_position, 0), 58);
    return t.res_curve[curve_position];
}

// The 32 lanes of one transform live in ONE wave, whose LDS operations execute in program order: between
// butterfly stages only the LDS counter has to drain (and the compiler must not reorder), no s_barrier.
struct WaveSync {
    __device__ __forceinline__ void operator()() const { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
};
constexpr WaveSync wave_sync{};

// DCT-IV of one 128-vector, executed by the 32 lanes `t` = 0..31 of a (sub)group: the staged butterflies of
// Mdct.cs:126-181 with the six stages held in registers.  Lane t owns the complex pair (t, t + 32) of the
// pre-rotation and, in stage k, the pair at distance h = 32 >> k inside its block; between stages each lane keeps one
// of its two results and swaps the other with lane t ^ (h / 2) through ds_swizzle (the crossbar: no LDS memory, no
// write -> read round trip; staging every stage through LDS was 3 % slower).  Every butterfly is the reference's
// arithmetic operand for operand, no FMA contraction, so the spectra are bit-identical; LDS is touched once, for
// the final Gray-code/bit-reverse permutation (:177-180).
// in: 128 doubles in LDS, written before the caller's last barrier; tmp: 128 doubles of LDS scratch, which MAY BE
// `in` itself (a wave's LDS operations execute in order: every lane has read its inputs before any lane writes);
// out: anywhere.
// group_sync(): makes the 32 lanes' LDS writes visible to each other (they share a wave: wave_sync).
template <int XOR>
__device__ __forceinline__ double swizzle_xor(double v)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)bits, (XOR << 10) | 0x1F);
    const int hi = __builtin_amdgcn_ds_swizzle((int)(bits >> 32), (XOR << 10) | 0x1F);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <class Tables, class Sync>
__device__ __forceinline__ void dct4_128(const Tables &T, const double *in, double *tmp, double *out, int t,
                                         Sync group_sync)
{
    // Mdct.cs:137-147: the two pre-rotations of this lane ARE its stage-0 operands
    double f0, f1, b0, b1;
    {
        const double a0 = in[2 * t], c0 = in[127 - 2 * t];
        const double s0 = T.sin_t[63 + t], k0 = T.cos_t[63 + t];
        f0 = a0 * k0 + c0 * s0;
        f1 = a0 * s0 - c0 * k0;
        const int i = t + 32;
        const double a1 = in[2 * i], c1 = in[127 - 2 * i];
        const double s1 = T.sin_t[63 + i], k1 = T.cos_t[63 + i];
        b0 = a1 * k1 + c1 * s1;
        b1 = a1 * s1 - c1 * k1;
    }
    // Mdct.cs:150-175
#define VGA_DCT_STAGE(H, NEXT)                                                                  \
    {                                                                                           \
        const int i = t & ((H) - 1);                                                            \
        const double s = T.sin_t[(H) - 1 + i], c = T.cos_t[(H) - 1 + i];                        \
        const double a = f0 - b0, b = f1 - b1;                                                  \
        const double nf0 = f0 + b0, nf1 = f1 + b1;                                              \
        const double nb0 = a * c + b * s, nb1 = a * s - b * c;                                  \
        if ((NEXT) > 0) {                                                                       \
            const bool upper = (t & (NEXT)) != 0;     /* keeps its back, hands its front over */ \
            const double r0 = swizzle_xor<(NEXT)>(upper ? nf0 : nb0);                           \
            const double r1 = swizzle_xor<(NEXT)>(upper ? nf1 : nb1);                           \
            f0 = upper ? r0 : nf0; f1 = upper ? r1 : nf1;                                       \
            b0 = upper ? nb0 : r0; b1 = upper ? nb1 : r1;                                       \
        } else {                                                                                \
            f0 = nf0; f1 = nf1; b0 = nb0; b1 = nb1;                                             \
        }                                                                                       \
    }
    VGA_DCT_STAGE(32, 16)
    VGA_DCT_STAGE(16, 8)
    VGA_DCT_STAGE(8, 4)
    VGA_DCT_STAGE(4, 2)
    VGA_DCT_STAGE(2, 1)
    VGA_DCT_STAGE(1, 0)
#undef VGA_DCT_STAGE
    // after the last stage lane t holds the complex values 2t and 2t + 1: four consecutive doubles
    tmp[4 * t] = f0;
    tmp[4 * t + 1] = f1;
    tmp[4 * t + 2] = b0;
    tmp[4 * t + 3] = b1;
    group_sync();
    // Mdct.cs:177-180
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + 32 * k;
        out[i] = tmp[T.shuffle[i]] * MDCT_SCALE;
    }
}

}  // namespace hca
}  // namespace vga
