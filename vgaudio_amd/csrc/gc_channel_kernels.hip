// gc_channel_kernels.hip -- GC-ADPCM channel metadata on the device.
//
// Replaces what GcAdpcmChannelBuilder derives when a channel is built
// (VGAudio/Formats/GcAdpcm/GcAdpcmChannelBuilder.cs:148-202):
//   * GcAdpcmAlignment (GcAdpcmAlignment.cs:20-63): decode up to the loop end, build the tail from the
//     wrapped loop, re-encode it with the history of the last kept frame, decode it again;
//   * GcAdpcmLoopContext (GcAdpcmLoopContext.cs:17-26): pred/scale byte + two history samples at the loop start;
//   * GcAdpcmSeekTable.CreateSeekTable (GcAdpcmSeekTable.cs:25-38): history pairs every N samples.
// The decode / encode steps are the codec kernels (gcadpcm_kernels.hip, gc_encode_kernel.hip); the kernels
// here are the gathers in between.  Loop geometry is per batch (GcAdpcmFormat applies one loop to all of its
// channels, GcAdpcmFormat.cs:32-39).
#include "common.hpp"
#include "gcadpcm_kernels.hpp"
#include "container_kernels.hpp"

namespace vga {
namespace gc {

// newPcm of GcAdpcmAlignment.cs:44-51 and the re-encode history (:54-55)
__global__ __launch_bounds__(256) void gc_align_gather_kernel(
    const int16_t *__restrict__ pcm, int64_t pcm_pitch, int loop_start, int loop_end, int samples_to_keep,
    int samples_to_encode, int16_t *__restrict__ new_pcm, int64_t new_pitch, int16_t *__restrict__ hist1,
    int16_t *__restrict__ hist2)
{
    const int ch = blockIdx.y;
    const int16_t *src = pcm + (int64_t)ch * pcm_pitch;
    const int head = loop_end - samples_to_keep;          // samples of the last kept-from frame onwards (:46)
    const int loop_length = loop_end - loop_start;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < samples_to_encode) {
        const int16_t v = i < head ? src[samples_to_keep + i] : src[loop_start + (i - head) % loop_length];
        new_pcm[(int64_t)ch * new_pitch + i] = v;
    }
    if (i == 0) {
        hist1[ch] = samples_to_keep < 1 ? (int16_t)0 : src[samples_to_keep - 1];
        hist2[ch] = samples_to_keep < 2 ? (int16_t)0 : src[samples_to_keep - 2];
    }
}

// loop context (adpcm = the ORIGINAL stream, GcAdpcmChannelBuilder.cs:179) and seek table
__global__ __launch_bounds__(256) void gc_channel_meta_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ pcm, int64_t pcm_pitch,
    int loop_start, int samples_per_entry, int entries, int16_t *__restrict__ seek, int64_t seek_pitch,
    int16_t *__restrict__ loop_context)
{
    const int ch = blockIdx.y;
    const int16_t *p = pcm + (int64_t)ch * pcm_pitch;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (seek && i < entries) {
        int16_t *t = seek + (int64_t)ch * seek_pitch;
        t[2 * i] = i == 0 ? (int16_t)0 : p[(int64_t)i * samples_per_entry - 1];       // the first entry is always 0
        t[2 * i + 1] = i == 0 ? (int16_t)0 : p[(int64_t)i * samples_per_entry - 2];
    }
    if (loop_context && i == 0) {
        int16_t *c = loop_context + ch * 3;
        if (loop_start == 0) {                            // "current loop context is valid": the default context
            c[0] = c[1] = c[2] = 0;
        } else {
            c[0] = (int16_t)adpcm[(int64_t)ch * adpcm_pitch + loop_start / 14 * 8];
            c[1] = loop_start < 1 ? (int16_t)0 : p[loop_start - 1];
            c[2] = loop_start < 2 ? (int16_t)0 : p[loop_start - 2];
        }
    }
}

int launch_align_gather(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int loop_start, int loop_end,
                        int samples_to_keep, int samples_to_encode, int16_t *d_new_pcm, int64_t new_pitch,
                        int16_t *d_hist1, int16_t *d_hist2, hipStream_t stream)
{
    if (nch <= 0) return VGA_OK;
    const int bx = samples_to_encode > 0 ? (samples_to_encode + 255) / 256 : 1;
    hipLaunchKernelGGL(gc_align_gather_kernel, dim3(bx, nch), dim3(256), 0, stream, d_pcm, pcm_pitch, loop_start, loop_end,
                       samples_to_keep, samples_to_encode, d_new_pcm, new_pitch, d_hist1, d_hist2);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_channel_meta(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_pcm, int64_t pcm_pitch, int nch,
                        int loop_start, int samples_per_entry, int entries, int16_t *d_seek, int64_t seek_pitch,
                        int16_t *d_loop_context, hipStream_t stream)
{
    if (nch <= 0 || (!d_seek && !d_loop_context)) return VGA_OK;
    const int bx = entries > 0 ? (entries + 255) / 256 : 1;
    hipLaunchKernelGGL(gc_channel_meta_kernel, dim3(bx, nch), dim3(256), 0, stream, d_adpcm, adpcm_pitch, d_pcm, pcm_pitch,
                       loop_start, samples_per_entry, entries, d_seek, seek_pitch, d_loop_context);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace gc
}  // namespace vga

// ---------------------------------------------------------------- DSP container image
// Replaces VGAudio/Containers/Dsp/DspWriter.cs:38-97: one 0x60-byte big-endian header per channel, then the
// audio -- a single channel verbatim, several channels interleaved in blocks of BytesPerInterleave
// (Utilities/Interleave.cs:43-78).  The image is assembled in HBM next to the encoder's output, so a file
// leaves the device with one copy.
namespace vga {
namespace gc {

struct DspGeometry {
    int sample_count, nibble_count, sample_rate, looping;
    int start_addr, end_addr, cur_addr;
    int bytes_per_interleave, frames_per_interleave, audio_data_size;
};

__device__ __forceinline__ void put_be16(uint8_t *p, int v) { p[0] = (uint8_t)(v >> 8); p[1] = (uint8_t)v; }
__device__ __forceinline__ void put_be32(uint8_t *p, int v)
{
    p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v;
}

// WriteHeader (:52-80): one thread per channel (96 bytes each)
__global__ __launch_bounds__(64) void gc_dsp_header_kernel(
    const uint8_t *__restrict__ adpcm, int64_t adpcm_pitch, const int16_t *__restrict__ coefs,
    const int16_t *__restrict__ gain, const int16_t *__restrict__ start_context,
    const int16_t *__restrict__ loop_context, int nch, DspGeometry g, uint8_t *__restrict__ file)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nch) return;
    uint8_t *h = file + (size_t)0x60 * i;
    for (int k = 0; k < 0x60; k++) h[k] = 0;
    put_be32(h + 0x00, g.sample_count);
    put_be32(h + 0x04, g.nibble_count);
    put_be32(h + 0x08, g.sample_rate);
    put_be16(h + 0x0c, g.looping ? 1 : 0);
    put_be16(h + 0x0e, 0);                                 // Format: 0 for ADPCM (:23)
    put_be32(h + 0x10, g.start_addr);
    put_be32(h + 0x14, g.end_addr);
    put_be32(h + 0x18, g.cur_addr);
    for (int k = 0; k < 16; k++) put_be16(h + 0x1c + 2 * k, coefs[i * 16 + k]);
    put_be16(h + 0x3c, gain ? gain[i] : 0);
    if (start_context) {
        for (int k = 0; k < 3; k++) put_be16(h + 0x3e + 2 * k, start_context[i * 3 + k]);
    } else {                                               // GcAdpcmChannel.cs:45: (Adpcm[0], 0, 0) for a fresh channel
        put_be16(h + 0x3e, (int)adpcm[(int64_t)i * adpcm_pitch]);
    }
    if (g.looping && loop_context)
        for (int k = 0; k < 3; k++) put_be16(h + 0x44 + 2 * k, loop_context[i * 3 + k]);
    put_be16(h + 0x4a, nch == 1 ? 0 : nch);
    put_be16(h + 0x4c, nch == 1 ? 0 : g.frames_per_interleave);
}

int launch_dsp_image(const uint8_t *d_adpcm, int64_t adpcm_pitch, int adpcm_len, const int16_t *d_coefs,
                     const int16_t *d_gain, const int16_t *d_start_context, const int16_t *d_loop_context, int nch,
                     int sample_count, int nibble_count, int sample_rate, int looping, int start_addr, int end_addr,
                     int cur_addr, int bytes_per_interleave, int frames_per_interleave, int audio_data_size,
                     int mono_bytes, uint8_t *d_file, size_t file_size, hipStream_t stream)
{
    VGA_HIP_TRY(hipMemsetAsync(d_file, 0, file_size, stream));
    DspGeometry g{sample_count, nibble_count, sample_rate, looping, start_addr, end_addr, cur_addr,
                  bytes_per_interleave, frames_per_interleave, audio_data_size};
    hipLaunchKernelGGL(gc_dsp_header_kernel, dim3((nch + 63) / 64), dim3(64), 0, stream, d_adpcm, adpcm_pitch, d_coefs, d_gain,
                       d_start_context, d_loop_context, nch, g, d_file);
    VGA_HIP_TRY(hipGetLastError());
    uint8_t *data = d_file + (size_t)0x60 * nch;
    // WriteData (:82-94): one channel verbatim (an "interleave" of one row), several in BytesPerInterleave blocks
    if (nch == 1) return container::launch_interleave(d_adpcm, adpcm_pitch, mono_bytes, 1, mono_bytes, mono_bytes, data, stream);
    return container::launch_interleave(d_adpcm, adpcm_pitch, adpcm_len, nch, bytes_per_interleave, audio_data_size, data, stream);
}

}  // namespace gc
}  // namespace vga
