// hca_kernels.hpp -- launchers for the CRI HCA kernels (device pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#include "hca_device.hpp"

namespace vga {
namespace hca {

// bytes of decoder workspace per (stream, frame): the scan's hand-over record (hca_decode_core.hpp)
size_t decode_record_bytes(const DeviceInfo &info);

// frames: stream s at d_frames + s*frames_pitch (frame_count*frame_size bytes + >= 8 bytes slack);
// pcm: stream s channel c at d_pcm + s*stream_pitch + c*ch_pitch (samples)
int launch_decode(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, const DeviceInfo &info, int16_t *d_pcm,
                  int64_t stream_pitch, int64_t ch_pitch, void *d_workspace, int *d_status, hipStream_t stream);

// d_crc_pow: uint16[4096], x^(8k) mod 0x18005 (built by the host)
int launch_encode(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, const PcmMap &map,
                  const DeviceInfo &info, uint8_t *d_frames, int64_t frames_pitch, const uint16_t *d_crc_pow,
                  int *d_status, hipStream_t stream, int first_frame = 0, int frame_limit = -1);

// hca_encode_wave_kernel.hip: one wave per run of frames, one or two channels (launch_encode hands such streams over)
bool encode_wave_kernel_takes(const DeviceInfo &info);
int launch_encode_wave(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, const PcmMap &map,
                       const DeviceInfo &info, uint8_t *d_frames, int64_t frames_pitch, const uint16_t *d_crc_pow,
                       int *d_status, hipStream_t stream, int first_frame, int end_frame, int frames_per_run_override);

}  // namespace hca
}  // namespace vga
