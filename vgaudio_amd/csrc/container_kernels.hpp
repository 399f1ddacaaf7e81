// container_kernels.hpp -- launchers for the container-image kernels
#pragma once
#include "common.hpp"

namespace vga {
namespace container {

// InterleaveExtensions.Interleave(byte[][], Stream, interleaveSize, outputSize) (Utilities/Interleave.cs:43-78) from
// `count` rows of `input_size` bytes (row r at src + r * pitch) into dst (output_size * count bytes, already zeroed).
int launch_interleave(const uint8_t *src, int64_t pitch, int input_size, int count, int interleave, int output_size,
                      uint8_t *dst, hipStream_t stream);

struct AdxHeaderArgs {
    int header_size, type, frame_size, nch, sample_rate, sample_count, highpass_frequency, version, encryption_type;
    int alignment_samples, looping, loop_start, loop_start_offset, loop_end, loop_end_offset;
    int footer_pos, footer_size, file_size;
};
// AdxWriter.WriteHeader (:81-117); the footer (:133-138) is a second launch AFTER the audio, as in the reference
int launch_adx_header(const AdxHeaderArgs &a, const int16_t *d_history, uint8_t *d_file, hipStream_t stream);
int launch_adx_footer(const AdxHeaderArgs &a, uint8_t *d_file, hipStream_t stream);

// copies one header_size-byte header (device) to the front of `count` images `pitch` bytes apart
int launch_replicate(const uint8_t *d_header, int header_size, uint8_t *d_files, int64_t pitch, int count, hipStream_t stream);

// WAVE 16-bit PCM <-> planar channels (Utilities/Interleave.cs:188-207 InterleavedByteToShort, :168-186
// ShortToInterleavedByte).  `interleaved` is a little-endian byte stream at ANY alignment; planar row c at
// pcm + c * pitch (in samples).
int launch_pcm16_deinterleave(const uint8_t *interleaved, int sample_count, int nch, int16_t *pcm, int64_t pitch, hipStream_t stream);
int launch_pcm16_interleave(const int16_t *pcm, int64_t pitch, int sample_count, int nch, uint8_t *interleaved, hipStream_t stream);

}  // namespace container
}  // namespace vga
