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

namespace vga {
namespace hca {

constexpr int SUBFRAMES = 8;
constexpr int SPSF = 128;     // SamplesPerSubFrame
constexpr int SPF = 1024;     // SamplesPerFrame

enum { CH_DISCRETE = 0, CH_STEREO_PRIMARY = 1, CH_STEREO_SECONDARY = 2 };

// HcaInfo subset + derived per-channel layout, passed by value to the kernels
struct DeviceInfo {
    int nch, frame_size, frame_count, sample_count, inserted_samples;
    int total_band_count, base_band_count, stereo_band_count, hfr_band_count, bands_per_hfr_group, hfr_group_count;
    int channel_type[8];
    int coded_count[8];
    uint8_t ath_curve[128];
};

__device__ __forceinline__ double f64_bits(uint64_t b) { return __longlong_as_double((long long)b); }

// sqrt(2.0 / 128) (CriHcaChannel.cs:19): exactly 0.125
constexpr double MDCT_SCALE = 0.125;

// The hot tables, copied into LDS once per workgroup (a table lookup from HBM-backed constant data
// costs an L1/L2 round trip per dependent access; from LDS ~64 cycles).
struct LdsTables {
    double sin_t[192], cos_t[192];      // size 2^b starts at (1<<b)-1; of the size-128 table only i < 64 is used
    double window[128];
    double dequant_scale[64];           // DequantizerScalingTable
    double quant_scale[64];             // QuantizerScalingTable
    double inv_step[16];                // QuantizerInverseStepSize
    double step[16];                    // QuantizerStepSize
    double dead_zone[16];               // QuantizerDeadZone (CriHcaTables.cs:68-78)
    uint8_t shuffle[128];
    uint8_t enc_bits[8][16], enc_value[8][16];   // QuantizeSpectrumBits / Value   (encoder, index q+8)
    uint8_t dec_bits[8][16];                     // QuantizedSpectrumBits          (decoder, index code)
    int8_t dec_value[8][16];                     // QuantizedSpectrumValue
    uint8_t max_bits[16];
    uint8_t res_curve[64];
};

__device__ __forceinline__ void load_tables(LdsTables &t, int tid, int nthreads)
{
    for (int i = tid; i < 191; i += nthreads) { t.sin_t[i] = f64_bits(MDCT_SinBits[i]); t.cos_t[i] = f64_bits(MDCT_CosBits[i]); }
    for (int i = tid; i < 128; i += nthreads) {
        t.window[i] = (double)__uint_as_float(HCA_MdctWindowF32Bits[i]);
        t.shuffle[i] = MDCT_Shuffle128[i];
        (&t.enc_bits[0][0])[i] = (&HCA_QuantizeSpectrumBits[0][0])[i];
        (&t.enc_value[0][0])[i] = (&HCA_QuantizeSpectrumValue[0][0])[i];
        (&t.dec_bits[0][0])[i] = (&HCA_QuantizedSpectrumBits[0][0])[i];
        (&t.dec_value[0][0])[i] = (&HCA_QuantizedSpectrumValue[0][0])[i];
    }
    for (int i = tid; i < 64; i += nthreads) {
        t.dequant_scale[i] = f64_bits(HCA_DequantizerScalingTableBits[i]);
        t.quant_scale[i] = f64_bits(HCA_QuantizerScalingTableBits[i]);
        t.res_curve[i] = i < 59 ? HCA_ScaleToResolutionCurve[i] : 0;
    }
    for (int i = tid; i < 16; i += nthreads) {
        t.inv_step[i] = f64_bits(HCA_QuantizerInverseStepSizeBits[i]);
        const double st = f64_bits(HCA_QuantizerStepSizeBits[i]);
        t.step[i] = st;
        t.dead_zone[i] = __longlong_as_double(__double_as_longlong(st / 2) - (long long)(HCA_ResolutionMaxValue[i] + 1));
        t.max_bits[i] = HCA_QuantizedSpectrumMaxBits[i];
    }
}

// CriHcaPacking.cs:60-69
__device__ __forceinline__ int calculate_resolution(const LdsTables &t, int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int curve_position = noise_level - 5 * scale_factor / 2 + 2;
    curve_position = min(max(curve_position, 0), 58);
    return t.res_curve[curve_position];
}

// DCT-IV of one 128-vector held in LDS, executed by the 32 lanes `t` = 0..31 of a (sub)group.
// in != tmp; out may be any 128-double LDS array.  Caller synchronises before (inputs written) and
// after (outputs read).  group_sync(): barrier among the lanes that cooperate on this transform.
template <class Sync>
__device__ __forceinline__ void dct4_128(const LdsTables &T, const double *in, double *tmp, double *out, int t,
                                         Sync group_sync)
{
    // Mdct.cs:137-147: 64 pre-rotations (2 per lane)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = t + 32 * k;
        const int i2 = i * 2;
        const double a = in[i2];
        const double b = in[127 - i2];
        const double s = T.sin_t[127 + i], c = T.cos_t[127 + i];
        tmp[i2] = a * c + b * s;
        tmp[i2 + 1] = a * s - b * c;
    }
    group_sync();
    // Mdct.cs:150-175: 6 stages x 32 butterflies (1 per lane)
#pragma unroll
    for (int stage = 0; stage < 6; stage++) {
        const int block_size_bits = 6 - stage;
        const int half_bits = block_size_bits - 1;
        const int block_size = 1 << block_size_bits;
        const int half = 1 << half_bits;
        const int block = t >> half_bits;
        const int i = t & (half - 1);
        const int front = (block * block_size + i) * 2;
        const int back = front + block_size;
        const double f0 = tmp[front], f1 = tmp[front + 1], b0 = tmp[back], b1 = tmp[back + 1];
        const double a = f0 - b0;
        const double b = f1 - b1;
        const double s = T.sin_t[half - 1 + i], c = T.cos_t[half - 1 + i];
        tmp[front] = f0 + b0;
        tmp[front + 1] = f1 + b1;
        tmp[back] = a * c + b * s;
        tmp[back + 1] = a * s - b * c;
        group_sync();
    }
    // Mdct.cs:177-180
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = t + 32 * k;
        out[i] = tmp[T.shuffle[i]] * MDCT_SCALE;
    }
}

}  // namespace hca
}  // namespace vga
