// gcadpcm_kernels.hpp -- launchers for the GC-ADPCM kernels (all pointers are device pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vga {
namespace gc {

int launch_coefs(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int16_t *d_coefs,
                 void *d_workspace, hipStream_t stream);
int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                  const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                  hipStream_t stream);
int launch_encode_v1(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                     const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                     hipStream_t stream);
int launch_decode(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch, int sample_count,
                  const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int64_t pcm_pitch, int *d_status,
                  hipStream_t stream);
int launch_synth(int16_t *d_pcm, int64_t pitch, int nch, int length, int first_channel, const uint32_t *d_params,
                 hipStream_t stream);

}  // namespace gc
}  // namespace vga
