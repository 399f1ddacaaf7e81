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
