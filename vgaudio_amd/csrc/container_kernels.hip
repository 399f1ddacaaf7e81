// container_kernels.hip -- file images assembled in HBM.  All of this is byte movement:
// HBM-bound, every byte read once and written once.
#include "container_kernels.hpp"

#include <algorithm>
#include <type_traits>

namespace vga {
namespace container {

template <int G> struct Granule;
template <> struct Granule<1> { using type = uint8_t; };
template <> struct Granule<2> { using type = uint16_t; };
template <> struct Granule<4> { using type = uint32_t; };
template <> struct Granule<8> { using type = uint2; };
template <> struct Granule<16> { using type = uint4; };

// One thread per G-byte granule of the OUTPUT (coalesced stores; loads are contiguous inside one interleave block).
// G divides every segment size, so a granule never straddles two rows; a short input leaves zero gaps.
template <int G>
__global__ __launch_bounds__(256) void interleave_kernel(const uint8_t *__restrict__ src, int64_t pitch, uint32_t input_size,
                                                         uint32_t count, uint32_t interleave, uint32_t output_size,
                                                         uint8_t *__restrict__ dst)
{
    const uint64_t o64 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * G;
    if (o64 >= (uint64_t)output_size * count) return;
    const uint32_t o = (uint32_t)o64;                      // images are < 2 GiB (the reference's FileSize is an int)
    const uint32_t in_blocks = (input_size + interleave - 1) / interleave, out_blocks = (output_size + interleave - 1) / interleave;
    const uint32_t stride = interleave * count;
    uint32_t b = o / stride;
    if (b > out_blocks - 1) b = out_blocks - 1;            // the (short) last block's rows are packed more tightly
    const uint32_t r = o - b * stride;
    const uint32_t cur_out = b == out_blocks - 1 ? output_size - (out_blocks - 1) * interleave : interleave;
    const uint32_t i = r / cur_out, within = r - i * cur_out;
    if (b >= in_blocks) return;                            // blocksToCopy = min(inBlockCount, outBlockCount)
    const uint32_t cur_in = b == in_blocks - 1 ? input_size - (in_blocks - 1) * interleave : interleave;
    const uint32_t n = cur_in < cur_out ? cur_in : cur_out;
    if (within >= n) return;
    const uint8_t *s = src + (int64_t)i * pitch + (uint64_t)interleave * b + within;
    uint8_t *d = dst + o;
    using T = typename Granule<G>::type;
    if (within + G <= n) *reinterpret_cast<T *>(d) = *reinterpret_cast<const T *>(s);
    else for (uint32_t k = 0; within + k < n; k++) d[k] = s[k];
}

int launch_interleave(const uint8_t *src, int64_t pitch, int input_size, int count, int interleave, int output_size,
                      uint8_t *dst, hipStream_t stream)
{
    if (input_size <= 0 || output_size <= 0 || count <= 0) return VGA_OK;
    if (count == 1)                                       // a plain copy of min(input, output) bytes: one block, any granule
        interleave = (int)std::min<int64_t>(round_up(std::max(input_size, output_size), 16), 0x7FFFFFF0);
    const int out_blocks = (output_size + interleave - 1) / interleave;
    const int last_out = output_size - (out_blocks - 1) * interleave;
    const uint64_t align = (uint64_t)(uintptr_t)src | (uint64_t)(uintptr_t)dst | (uint64_t)pitch | (uint64_t)interleave |
                           (uint64_t)(count == 1 ? 0 : last_out);
    const uint64_t total = (uint64_t)output_size * count;
    auto go = [&](auto g) {
        constexpr int G = decltype(g)::value;
        const uint64_t granules = (total + G - 1) / G;
        hipLaunchKernelGGL(interleave_kernel<G>, dim3((unsigned)((granules + 255) / 256)), dim3(256), 0, stream, src, pitch,
                           (uint32_t)input_size, (uint32_t)count, (uint32_t)interleave, (uint32_t)output_size, dst);
    };
    if (!(align & 15)) go(std::integral_constant<int, 16>{});
    else if (!(align & 7)) go(std::integral_constant<int, 8>{});
    else if (!(align & 3)) go(std::integral_constant<int, 4>{});
    else if (!(align & 1)) go(std::integral_constant<int, 2>{});
    else go(std::integral_constant<int, 1>{});
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

// A positioned big-endian writer over the image, as the reference's BinaryWriter over MemoryStream(byte[FileSize]).
struct Cursor {
    uint8_t *buf;
    int size, pos;
    __device__ void put8(int v) { if (pos < size) buf[pos] = (uint8_t)v; pos++; }
    __device__ void put16(int v) { put8(v >> 8); put8(v); }
    __device__ void put32(int v) { put16(v >> 16); put16(v); }
};

// AdxWriter.WriteHeader (:81-117).  Fields are written one after the other whatever HeaderSize is; "(c)CRI" and
// then the audio overwrite what ran past it -- the later launches on the same stream reproduce that order.
__global__ void adx_header_kernel(AdxHeaderArgs a, const int16_t *__restrict__ history, uint8_t *__restrict__ file)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Cursor c{file, a.file_size, 0};
    c.put16(0x8000);
    c.put16(a.header_size);
    c.put8(a.type);
    c.put8(a.frame_size);
    c.put8(4);                                             // bit depth
    c.put8(a.nch);
    c.put32(a.sample_rate);
    c.put32(a.sample_count);
    c.put16(a.type != 2 ? a.highpass_frequency : 0);       // CriAdxType.Fixed
    c.put8(a.version);
    c.put8(a.encryption_type);
    if (a.version == 4) {
        c.put32(0);
        for (int i = 0; i < a.nch; i++) { c.put16(history[i]); c.put16(history[i]); }
        if (a.nch == 1) c.put32(0);
    }
    c.put16(a.alignment_samples);
    c.put16(a.looping ? 1 : 0);
    c.put32(a.looping ? 1 : 0);
    c.put32(a.loop_start);
    c.put32(a.loop_start_offset);
    c.put32(a.loop_end);
    c.put32(a.loop_end_offset);
    c.pos = a.header_size - 2;
    const char sig[6] = {'(', 'c', ')', 'C', 'R', 'I'};
    for (int k = 0; k < 6; k++) c.put8(sig[k]);
}

__global__ void adx_footer_kernel(AdxHeaderArgs a, uint8_t *__restrict__ file)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Cursor c{file, a.file_size, a.footer_pos};
    c.put16(0x8001);
    c.put16(a.footer_size - 4);
}

int launch_adx_header(const AdxHeaderArgs &a, const int16_t *d_history, uint8_t *d_file, hipStream_t stream)
{
    hipLaunchKernelGGL(adx_header_kernel, dim3(1), dim3(64), 0, stream, a, d_history, d_file);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_adx_footer(const AdxHeaderArgs &a, uint8_t *d_file, hipStream_t stream)
{
    hipLaunchKernelGGL(adx_footer_kernel, dim3(1), dim3(64), 0, stream, a, d_file);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

__global__ __launch_bounds__(256) void replicate_kernel(const uint8_t *__restrict__ header, int header_size,
                                                        uint8_t *__restrict__ files, int64_t pitch)
{
    uint8_t *dst = files + (int64_t)blockIdx.x * pitch;
    for (int k = threadIdx.x; k < header_size; k += 256) dst[k] = header[k];
}

int launch_replicate(const uint8_t *d_header, int header_size, uint8_t *d_files, int64_t pitch, int count, hipStream_t stream)
{
    if (count <= 0 || header_size <= 0) return VGA_OK;
    hipLaunchKernelGGL(replicate_kernel, dim3(count), dim3(256), 0, stream, d_header, header_size, d_files, pitch);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

// ---------------------------------------------------------------- WAVE: interleaved 16-bit PCM <-> planar channels
// A 256-sample x 32-channel tile goes through LDS so that both the interleaved side (runs of nch shorts per sample)
// and the planar side (runs of samples per channel) are accessed in contiguous runs.  Pure transposition: HBM-bound.
constexpr int PCM_TS = 256, PCM_TC = 32;

template <bool ALIGNED>
__global__ __launch_bounds__(256) void pcm16_deinterleave_kernel(const uint8_t *__restrict__ in, int n, int nch,
                                                                 int16_t *__restrict__ out, int64_t pitch)
{
    __shared__ int16_t tile[PCM_TS][PCM_TC + 1];
    const int i0 = blockIdx.x * PCM_TS, c0 = blockIdx.y * PCM_TC;
    const int st = min(PCM_TS, n - i0), ct = min(PCM_TC, nch - c0);
    for (int e = threadIdx.x; e < st * ct; e += 256) {
        const int s = e / ct, c = e - s * ct;
        const int64_t off = ((int64_t)(i0 + s) * nch + c0 + c) * 2;
        tile[s][c] = ALIGNED ? *reinterpret_cast<const int16_t *>(in + off) : (int16_t)(in[off] | (in[off + 1] << 8));
    }
    __syncthreads();
    for (int e = threadIdx.x; e < st * ct; e += 256) {
        const int c = e / st, s = e - c * st;
        out[(int64_t)(c0 + c) * pitch + i0 + s] = tile[s][c];
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(256) void pcm16_interleave_kernel(const int16_t *__restrict__ in, int64_t pitch, int n, int nch,
                                                               uint8_t *__restrict__ out)
{
    __shared__ int16_t tile[PCM_TS][PCM_TC + 1];
    const int i0 = blockIdx.x * PCM_TS, c0 = blockIdx.y * PCM_TC;
    const int st = min(PCM_TS, n - i0), ct = min(PCM_TC, nch - c0);
    for (int e = threadIdx.x; e < st * ct; e += 256) {
        const int c = e / st, s = e - c * st;
        tile[s][c] = in[(int64_t)(c0 + c) * pitch + i0 + s];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < st * ct; e += 256) {
        const int s = e / ct, c = e - s * ct;
        const int64_t off = ((int64_t)(i0 + s) * nch + c0 + c) * 2;
        const int16_t v = tile[s][c];
        if (ALIGNED) *reinterpret_cast<int16_t *>(out + off) = v;
        else { out[off] = (uint8_t)v; out[off + 1] = (uint8_t)(v >> 8); }
    }
}

int launch_pcm16_deinterleave(const uint8_t *interleaved, int sample_count, int nch, int16_t *pcm, int64_t pitch, hipStream_t stream)
{
    if (sample_count <= 0 || nch <= 0) return VGA_OK;
    const dim3 grid((sample_count + PCM_TS - 1) / PCM_TS, (nch + PCM_TC - 1) / PCM_TC);
    if (((uintptr_t)interleaved & 1) == 0)
        hipLaunchKernelGGL(pcm16_deinterleave_kernel<true>, grid, dim3(256), 0, stream, interleaved, sample_count, nch, pcm, pitch);
    else
        hipLaunchKernelGGL(pcm16_deinterleave_kernel<false>, grid, dim3(256), 0, stream, interleaved, sample_count, nch, pcm, pitch);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_pcm16_interleave(const int16_t *pcm, int64_t pitch, int sample_count, int nch, uint8_t *interleaved, hipStream_t stream)
{
    if (sample_count <= 0 || nch <= 0) return VGA_OK;
    const dim3 grid((sample_count + PCM_TS - 1) / PCM_TS, (nch + PCM_TC - 1) / PCM_TC);
    if (((uintptr_t)interleaved & 1) == 0)
        hipLaunchKernelGGL(pcm16_interleave_kernel<true>, grid, dim3(256), 0, stream, pcm, pitch, sample_count, nch, interleaved);
    else
        hipLaunchKernelGGL(pcm16_interleave_kernel<false>, grid, dim3(256), 0, stream, pcm, pitch, sample_count, nch, interleaved);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace container
}  // namespace vga
