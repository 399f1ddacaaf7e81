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
__device__ __forceinline__ double mdct_window(int i) { return (double)__uint_as_float(HCA_MdctWindowF32Bits[i]); }
__device__ __forceinline__ double mdct_sin(int bits, int i) { return f64_bits(MDCT_SinBits[(1 << bits) - 1 + i]); }
__device__ __forceinline__ double mdct_cos(int bits, int i) { return f64_bits(MDCT_CosBits[(1 << bits) - 1 + i]); }

// sqrt(2.0 / 128) (CriHcaChannel.cs:19): exactly 0.125
constexpr double MDCT_SCALE = 0.125;

// CriHcaPacking.cs:60-69
__device__ __forceinline__ int calculate_resolution(int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int curve_position = noise_level - 5 * scale_factor / 2 + 2;
    curve_position = min(max(curve_position, 0), 58);
    return HCA_ScaleToResolutionCurve[curve_position];
}

// DCT-IV of one 128-vector held in LDS, executed by the 32 lanes `t` = 0..31 of a (sub)group.
// `in` and `out` may alias `tmp` only as documented: in != tmp; out may be any 128-double LDS array
// different from tmp.  Caller synchronises the group before (inputs written) and after (outputs read).
// GROUP_SYNC(): barrier among the lanes that cooperate on this transform.
template <class Sync>
__device__ __forceinline__ void dct4_128(const double *in, double *tmp, double *out, int t, Sync group_sync)
{
    // Mdct.cs:137-147: 64 pre-rotations (2 per lane)
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int i = t + 32 * k;
        const int i2 = i * 2;
        const double a = in[i2];
        const double b = in[127 - i2];
        const double s = mdct_sin(7, i), c = mdct_cos(7, i);
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
        const double s = mdct_sin(half_bits, i), c = mdct_cos(half_bits, i);
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
        out[i] = tmp[MDCT_Shuffle128[i]] * MDCT_SCALE;
    }
}

}  // namespace hca
}  // namespace vga
