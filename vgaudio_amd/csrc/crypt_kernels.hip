// crypt_kernels.hip -- ADX and HCA encryption passes: byte passes directly behind the codecs.
//   ADX  VGAudio/Codecs/CriAdx/CriAdxEncryption.cs:8-94   XOR of each frame's scale with a 15-bit LCG stream
//   HCA  VGAudio/Codecs/CriHca/CriHcaEncryption.cs:12-33  byte substitution + CRC-16 refresh per frame
// HBM-bound: ADX touches 2 of every frame_size bytes (but reads the frame for the emptiness test), HCA reads and
// writes every byte once.
#include "crypt_kernels.hpp"

namespace vga {
namespace crypt {

// The reference steps xor = (xor * mult + inc) & 0x7fff once per (frame, channel) slot, serially.  The k-th state is
// an affine map of the seed modulo 2^15, obtained here by square-and-multiply so every slot is independent.
struct Affine { unsigned a, c; };      // x -> a * x + c  (mod 2^15)
__device__ __forceinline__ Affine lcg_power(unsigned mult, unsigned inc, uint64_t k)
{
    Affine r{1u, 0u};
    unsigned a = mult & 0x7fffu, c = inc & 0x7fffu;
    while (k) {
        if (k & 1) { r.a = (r.a * a) & 0x7fffu; r.c = (r.c * a + c) & 0x7fffu; }
        c = (c * (a + 1)) & 0x7fffu;
        a = (a * a) & 0x7fffu;
        k >>= 1;
    }
    return r;
}

// EncryptDecryptChannel (:16-41): one thread per (frame, channel)
__global__ __launch_bounds__(256) void adx_crypt_kernel(uint8_t *__restrict__ audio, int64_t pitch, int frame_count, int nch,
                                                        int frame_size, AdxKey key, int encryption_type)
{
    const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;      // frame * nch + channel: the LCG's own order
    if (slot >= (int64_t)frame_count * nch) return;
    const int frame = (int)(slot / nch), ch = (int)(slot - (int64_t)frame * nch);
    uint8_t *f = audio + (int64_t)ch * pitch + (int64_t)frame * frame_size;
    bool not_empty = false;                                            // FrameNotEmpty (:96-107)
    for (int b = 0; b < frame_size; b++) not_empty |= f[b] != 0;
    if (!not_empty) return;
    const Affine p = lcg_power((unsigned)key.mult, (unsigned)key.inc, (uint64_t)slot);
    const unsigned x = (p.a * ((unsigned)key.seed & 0x7fffu) + p.c) & 0x7fffu;
    // the reference XORs with (byte)(xor >> 8) where xor is the int seed before the first step: an unmasked seed
    // only matters for slot 0, handled by using the caller's seed bits there
    const unsigned hi = slot == 0 ? ((unsigned)key.seed >> 8) & 0xffu : x >> 8;
    const unsigned lo = slot == 0 ? (unsigned)key.seed & 0xffu : x & 0xffu;
    uint8_t b0 = (uint8_t)(f[0] ^ hi);
    if (encryption_type == 9) b0 &= 0x1f;
    f[0] = b0;
    f[1] = (uint8_t)(f[1] ^ lo);
}

// GetScales + TestKey (:59-94): one workgroup per candidate key, valid[k] = 1 when the key explains every scale
__global__ __launch_bounds__(256) void adx_test_keys_kernel(const uint8_t *__restrict__ audio, int64_t pitch, int frame_count,
                                                            int nch, int frame_size, int encryption_type,
                                                            const AdxKey *__restrict__ keys, int *__restrict__ valid)
{
    __shared__ int bad;
    const AdxKey key = keys[blockIdx.x];
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const unsigned mask = encryption_type == 8 ? 0xE000u : 0x1000u;
    const int64_t total = (int64_t)frame_count * nch;
    const Affine stride = lcg_power((unsigned)key.mult, (unsigned)key.inc, 256);
    const Affine first = lcg_power((unsigned)key.mult, (unsigned)key.inc, threadIdx.x);
    unsigned x = (first.a * ((unsigned)key.seed & 0x7fffu) + first.c) & 0x7fffu;
    for (int64_t slot = threadIdx.x; slot < total; slot += 256) {
        const int frame = (int)(slot / nch), ch = (int)(slot - (int64_t)frame * nch);
        const uint8_t *f = audio + (int64_t)ch * pitch + (int64_t)frame * frame_size;
        const unsigned scale = ((unsigned)f[0] << 8) | f[1];
        const unsigned xr = slot == 0 ? (unsigned)key.seed : x;        // the first comparison uses the raw seed
        if (((scale ^ xr) & mask) != 0 && scale != 0) bad = 1;
        x = (stride.a * x + stride.c) & 0x7fffu;
        if ((slot >> 8) % 64 == 63) {                                  // leave early once any slot has failed
            __syncthreads();
            if (bad) break;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) valid[blockIdx.x] = bad ? 0 : 1;
}

// multiply in GF(2)[x] / (x^16 + x^15 + x^2 + 1)
__device__ __forceinline__ unsigned gf_mul16(unsigned a, unsigned b)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = ((r << 1) ^ ((r & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        if ((a >> i) & 1u) r ^= b;
    }
    return r;
}

// CryptFrame (:20-33): one wave per frame.  Each lane substitutes a contiguous chunk and CRCs it; the chunk CRCs
// are shifted to their place with x^(8k) mod P (crc_pow, the encoder's table) and XOR-reduced across the wave.
__global__ __launch_bounds__(64) void hca_crypt_kernel(uint8_t *__restrict__ frames, int64_t frames_pitch, int frame_count,
                                                       int frame_size, const uint8_t *__restrict__ table,
                                                       const uint16_t *__restrict__ crc_pow)
{
    __shared__ uint8_t sub[256];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) sub[i] = table[i];
    __syncthreads();
    const int64_t idx = blockIdx.x;
    const int stream = (int)(idx / frame_count), frame = (int)(idx - (int64_t)stream * frame_count);
    uint8_t *a = frames + (int64_t)stream * frames_pitch + (int64_t)frame * frame_size;
    const int nbytes = frame_size - 2;
    const int chunk = (nbytes + 63) / 64;
    const int begin = min(lane * chunk, nbytes), end = min(begin + chunk, nbytes);
    unsigned crc = 0;
    for (int i = begin; i < end; i++) {
        const unsigned byte = sub[a[i]];
        a[i] = (uint8_t)byte;
        crc ^= byte << 8;
#pragma unroll
        for (int j = 0; j < 8; j++) crc = ((crc << 1) ^ ((crc & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
    }
    unsigned part = begin < end ? gf_mul16(crc, crc_pow[nbytes - end]) : 0u;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part ^= (unsigned)__shfl_xor((int)part, o);
    if (lane == 0) {
        a[nbytes] = (uint8_t)(part >> 8);
        a[nbytes + 1] = (uint8_t)part;
    }
}

int launch_adx_crypt(uint8_t *d_audio, int64_t pitch, int frame_count, int nch, int frame_size, const AdxKey &key,
                     int encryption_type, hipStream_t stream)
{
    const int64_t total = (int64_t)frame_count * nch;
    if (total <= 0) return VGA_OK;
    hipLaunchKernelGGL(adx_crypt_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, d_audio, pitch, frame_count, nch,
                       frame_size, key, encryption_type);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_adx_test_keys(const uint8_t *d_audio, int64_t pitch, int frame_count, int nch, int frame_size, int encryption_type,
                         const AdxKey *d_keys, int nkeys, int *d_valid, hipStream_t stream)
{
    if (nkeys <= 0) return VGA_OK;
    hipLaunchKernelGGL(adx_test_keys_kernel, dim3(nkeys), dim3(256), 0, stream, d_audio, pitch, frame_count, nch, frame_size,
                       encryption_type, d_keys, d_valid);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_hca_crypt(uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                     const uint8_t *d_table, const uint16_t *d_crc_pow, hipStream_t stream)
{
    const int64_t total = (int64_t)nstreams * frame_count;
    if (total <= 0) return VGA_OK;
    hipLaunchKernelGGL(hca_crypt_kernel, dim3((unsigned)total), dim3(64), 0, stream, d_frames, frames_pitch, frame_count, frame_size,
                       d_table, d_crc_pow);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace crypt
}  // namespace vga
