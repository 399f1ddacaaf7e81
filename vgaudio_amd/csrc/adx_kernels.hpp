// adx_kernels.hpp -- launchers for the CRI ADX kernels (device pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vga {
namespace adx {

// CriAdxParameters (Codecs/CriAdx/CriAdxParameters.cs:5-12) with the two predictor
// coefficients already resolved on the host (Fixed: Coefs[Filter]; else CalculateCoefficients).
struct AdxDeviceParams {
    int frame_size;
    int version;
    int type;       // 2 Fixed, 3 Linear, 4 Exponential
    int filter;
    int padding;
    int history;
    int coef0, coef1;
};

// d_own_frames / d_own_samples (device, one int per channel; nullptr: every channel is pcm_length / sample_count long): the
// channels are shorter streams zero-padded to the launch's length (the ragged entry points' length buckets) -- what lies past
// a channel's own ceil(length / 32) frames / own samples is not output, and the seams there are left alone.
int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length, const AdxDeviceParams &p,
                  uint8_t *d_out, int64_t out_pitch, int16_t *d_history_out, hipStream_t stream, const int *d_own_frames = nullptr);
int launch_decode(const uint8_t *d_adpcm, int64_t in_pitch, int nch, int sample_count, const AdxDeviceParams &p,
                  int16_t *d_pcm, int64_t pcm_pitch, int *d_status, hipStream_t stream, const int *d_own_samples = nullptr);

}  // namespace adx
}  // namespace vga
