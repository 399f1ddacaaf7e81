// gcadpcm_kernels.hpp -- launchers for the GC-ADPCM kernels (all pointers are device pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vga {
namespace gc {

int launch_coefs(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int16_t *d_coefs,
                 void *d_workspace, hipStream_t stream);
// d_scratch (optional, encode_scratch_bytes(nch) bytes): where the time pieces' states go; allocated stream-ordered when null
int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                  const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                  hipStream_t stream, void *d_scratch = nullptr, size_t scratch_bytes = 0);
size_t encode_scratch_bytes(int nch);
// gc_decode_kernel.hip (serial wave + helper waves)
int launch_decode(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch, int sample_count,
                  const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int64_t pcm_pitch, int *d_status,
                  hipStream_t stream);
// gc_channel_kernels.hip (channel metadata)
int launch_align_gather(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int loop_start, int loop_end,
                        int samples_to_keep, int samples_to_encode, int16_t *d_new_pcm, int64_t new_pitch,
                        int16_t *d_hist1, int16_t *d_hist2, hipStream_t stream);
int launch_channel_meta(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_pcm, int64_t pcm_pitch, int nch,
                        int loop_start, int samples_per_entry, int entries, int16_t *d_seek, int64_t seek_pitch,
                        int16_t *d_loop_context, hipStream_t stream);
int launch_dsp_image(const uint8_t *d_adpcm, int64_t adpcm_pitch, int adpcm_len, const int16_t *d_coefs,
                     const int16_t *d_gain, const int16_t *d_start_context, const int16_t *d_loop_context, int nch,
                     int sample_count, int nibble_count, int sample_rate, int looping, int start_addr, int end_addr,
                     int cur_addr, int bytes_per_interleave, int frames_per_interleave, int audio_data_size,
                     int mono_bytes, uint8_t *d_file, size_t file_size, hipStream_t stream);
int launch_synth(int16_t *d_pcm, int64_t pitch, int nch, int length, int first_channel, const uint32_t *d_params,
                 hipStream_t stream);

}  // namespace gc
}  // namespace vga
