// crypt_kernels.hip -- ADX and HCA encryption passes: byte passes directly behind the codecs.
//   ADX  VGAudio/Codecs/CriAdx/CriAdxEncryption.cs:8-94   XOR of each frame's scale with a 15-bit LCG stream
//   HCA  VGAudio/Codecs/CriHca/CriHcaEncryption.cs:12-33  byte substitution + CRC-16 refresh per frame
// HBM-bound: ADX touches 2 of every frame_size bytes (but reads the frame for the emptiness test), HCA reads and
// writes every byte once.
#include "crypt_kernels.hpp"
#include "hca_device.hpp"

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


// ---------------------------------------------------------------- VGAudio.Tools/CrackAdx/GuessAdx.cs:118-179
// The brute-force key search of the reference's `crackadx` tool for one file's frame scales: for every scale index
// (0..0xFFF -> the LCG state at the first non-empty frame), every candidate multiplier and every candidate increment,
// step the LCG along the scales until one disagrees.  4096 x 1024 x 1024 candidates for type 8 (4096 x 2048 x 4096
// for type 9); a candidate dies after 1.14 (type 9: 2) comparisons on average, so this is a few milliseconds of
// integer work: one thread per (index, multiplier), looping over the increments.  Survivors are appended to a list;
// FindStartingKey, the duplicate filter and KeyIsValid run on the host over that (short) list.
__global__ __launch_bounds__(256) void adx_guess_keys_kernel(
    const uint16_t *__restrict__ scales, int nscales, int start_frame, int validation_mask, int max_seed,
    const uint32_t *__restrict__ seed_bitmap /* 0x8000 bits, or null: every seed allowed */,
    const int *__restrict__ mults, int nmult, const int *__restrict__ incs, int ninc, int *__restrict__ out /* [cap][3] */,
    int cap, int *__restrict__ count)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)0x1000 * nmult) return;
    const int index = (int)(t / nmult), m = (int)(t - (int64_t)index * nmult);
    const unsigned seed = ((unsigned)scales[start_frame] ^ (unsigned)index) & (unsigned)(max_seed - 1);
    if (seed_bitmap && !((seed_bitmap[seed >> 5] >> (seed & 31)) & 1u)) return;   // TryScale :153
    const unsigned mult = (unsigned)mults[m];
    // the first comparison does not depend on the increment
    const unsigned s0 = scales[start_frame];
    if (((s0 ^ seed) & (unsigned)validation_mask) != 0 && s0 != 0) return;
    for (int n = 0; n < ninc; n++) {
        const unsigned inc = (unsigned)incs[n];
        unsigned x = (seed * mult + inc) & 0x7fffu;
        bool match = true;
        for (int i = start_frame + 1; i < nscales; i++) {
            const unsigned sc = scales[i];
            if (((sc ^ x) & (unsigned)validation_mask) != 0 && sc != 0) { match = false; break; }
            x = (x * mult + inc) & 0x7fffu;
        }
        if (match) {
            const int slot = atomicAdd(count, 1);
            if (slot < cap) { out[3 * slot] = (int)seed; out[3 * slot + 1] = (int)mult; out[3 * slot + 2] = (int)inc; }
        }
    }
}

// ---------------------------------------------------------------- CriHcaEncryption.TestKey (CriHcaEncryption.cs:48-63)
// One workgroup per candidate key, one lane per tested frame (at most FramesToTest = 10): the frame is decrypted on the
// fly (substitution table in LDS, the CRC CryptFrame refreshes is computed for the last two bytes) and walked with
// CriHcaPacking.UnpackFrame's bit reader for validity only (UnpackFrameHeader / DeltaDecode failures,
// UnpackingWasSuccessful; CriHcaPacking.cs:10-15, :71-237).  The reference tests a key's frames ONE AFTER THE OTHER and
// stops at the first that does not unpack; a wrong sync word throws InvalidDataException out of the whole search.  The
// lanes test all frames at once and the outcome is put back in order: valid[key] = 1 when every tested frame unpacks,
// 2 when a frame with a wrong sync word comes before the first frame that fails to unpack (the reference throws while
// testing this key), 0 otherwise (rejected at its first failing frame; what later frames hold is never looked at).
struct HcaTestReader {
    const uint8_t *frame;
    const uint8_t *sub;
    int frame_size, frame_bits, pos;
    unsigned crc;
    __device__ __forceinline__ unsigned byte_at(int i) const
    {
        if (i >= frame_size) return 0;
        if (i < frame_size - 2) return sub[frame[i]];
        return i == frame_size - 2 ? (crc >> 8) & 0xffu : crc & 0xffu;
    }
    __device__ __forceinline__ int peek(int bits) const         // BitReader.PeekInt (BitReader.cs:51-92)
    {
        if (bits == 0) return 0;
        const int remaining = frame_bits - pos;
        if (remaining <= 0) return 0;
        const int first = pos >> 3;
        uint64_t w = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) w = (w << 8) | byte_at(first + k);
        int v = (int)((w >> (40 - (pos & 7) - bits)) & ((1u << bits) - 1u));
        if (bits > remaining) v = (v >> (bits - remaining)) << (bits - remaining);
        return v;
    }
    __device__ __forceinline__ int read(int bits) { const int v = peek(bits); pos += bits; return v; }
};

__global__ __launch_bounds__(64) void hca_test_keys_kernel(
    const uint8_t *__restrict__ frames, int first_frame, int ntest, hca::DeviceInfo info, const uint8_t *__restrict__ tables,
    int *__restrict__ valid, int *__restrict__ flags)
{
    __shared__ hca::DecTables T;
    __shared__ uint8_t sub[256];
    __shared__ int first_fail, first_bad_sync;               // lowest tested frame that fails to unpack / has a wrong sync word
    const int lane = threadIdx.x;
    hca::load_tables(T, lane, 64);
    for (int i = lane; i < 256; i += 64) sub[i] = tables[(size_t)blockIdx.x * 256 + i];
    if (lane == 0) { first_fail = 64; first_bad_sync = 64; }
    __syncthreads();
    bool ok = true;
    if (lane < ntest) {
        HcaTestReader r;
        r.frame = frames + (size_t)(first_frame + lane) * info.frame_size;
        r.sub = sub;
        r.frame_size = info.frame_size;
        r.frame_bits = info.frame_size * 8;
        r.pos = 0;
        unsigned crc = 0;                                       // Crc16.Compute over the decrypted bytes (CryptFrame :30-32)
        for (int i = 0; i < info.frame_size - 2; i++) {
            crc ^= (unsigned)sub[r.frame[i]] << 8;
#pragma unroll
            for (int j = 0; j < 8; j++) crc = ((crc << 1) ^ ((crc & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        }
        r.crc = crc;
        const bool bad_sync = r.read(16) != 0xffff;
        if (bad_sync) { atomicMin(&first_bad_sync, lane); ok = false; }
        const int noise_level = r.read(9);
        const int eval_boundary = r.read(7);
        bool any_delta_bits = false;
        uint8_t res[8][128];
        for (int c = 0; c < info.nch && ok; c++) {
            const int count = info.coded_count[c];
            const int delta_bits = r.read(3);                   // ReadScaleFactors (:111-130)
            any_delta_bits |= delta_bits > 0;
            int prev = 0;
            const int max_delta = delta_bits > 0 ? 1 << (delta_bits - 1) : 0;
            for (int i = 0; i < 128; i++) {
                int sf = 0;
                if (i < count && delta_bits != 0) {
                    if (delta_bits >= 6 || i == 0) sf = r.read(6);
                    else {                                      // DeltaDecode (:185-211)
                        const int delta = r.read(delta_bits) - (max_delta - 1);
                        if (delta < max_delta) {
                            sf = prev + delta;
                            if (sf < 0 || sf > 63) { ok = false; break; }
                        } else sf = r.read(6);
                    }
                    prev = sf;
                }
                int rs = 0;
                if (i < count) rs = hca::calculate_resolution(T, sf, info.ath_curve[i] + noise_level - (i < eval_boundary ? 1 : 0));
                res[c][i] = (uint8_t)rs;
            }
            if (!ok) break;
            if (info.channel_type[c] == hca::CH_STEREO_SECONDARY) r.pos += 32;
            else if (info.hfr_group_count > 0) r.pos += 6 * info.hfr_group_count;
        }
        if (ok) {
            for (int sf = 0; sf < hca::SUBFRAMES; sf++)         // ReadSpectralCoefficients (:148-183): lengths only
                for (int c = 0; c < info.nch; c++)
                    for (int s = 0; s < info.coded_count[c]; s++) {
                        const int resolution = res[c][s];
                        int bits = T.max_bits[resolution];
                        const int code = r.peek(bits);
                        if (resolution < 8) bits = T.dec_bits[resolution][code];
                        else if (code / 2 == 0) bits--;
                        r.pos += bits;
                    }
            const int remaining = r.frame_bits - r.pos;        // UnpackingWasSuccessful (:213-237)
            const bool empty = noise_level <= 0 && !any_delta_bits;
            ok = (remaining >= 16 && remaining <= 128) || empty || (noise_level == 0 && remaining >= 16);
        }
        if (!ok && !bad_sync) atomicMin(&first_fail, lane);
    }
    __syncthreads();
    if (lane == 0) {
        const int outcome = first_bad_sync < first_fail ? 2 : (first_fail < 64 ? 0 : 1);
        valid[blockIdx.x] = outcome;
        if (outcome == 2) atomicOr(flags, 1);
    }
}

// FindFirstNonEmptyFrame (CriHcaEncryption.cs:65-88): smallest frame index with a non-zero byte in [2, size - 2)
__global__ __launch_bounds__(64) void hca_first_non_empty_kernel(const uint8_t *__restrict__ frames, int frame_count, int frame_size,
                                                                 int *__restrict__ first)
{
    const int f = blockIdx.x;
    if (f >= frame_count) return;
    bool any = false;
    for (int i = 2 + threadIdx.x; i < frame_size - 2; i += 64) any |= frames[(size_t)f * frame_size + i] != 0;
    if (__any(any) && threadIdx.x == 0) atomicMin(first, f);
}

// VGAudio.Tools/CrackHca/Crack.cs:43-80 (LoadFrequencies' counting): byte-value counts at the first `positions` bytes
// of every frame; one thread per frame, LDS histograms per workgroup
__global__ __launch_bounds__(256) void hca_byte_position_counts_kernel(const uint8_t *__restrict__ frames, int64_t frames_pitch,
                                                                       int nstreams, int frame_count, int frame_size, int positions,
                                                                       unsigned *__restrict__ counts)
{
    extern __shared__ unsigned s_hist[];                        // [positions][256]
    for (int i = threadIdx.x; i < positions * 256; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int64_t total = (int64_t)nstreams * frame_count;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int s = (int)(id / frame_count), f = (int)(id - (int64_t)s * frame_count);
        const uint8_t *a = frames + (int64_t)s * frames_pitch + (int64_t)f * frame_size;
        for (int p = 0; p < positions && p < frame_size; p++) atomicAdd(&s_hist[p * 256 + a[p]], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < positions * 256; i += 256)
        if (s_hist[i]) atomicAdd(&counts[i], s_hist[i]);
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

int launch_adx_guess_keys(const uint16_t *d_scales, int nscales, int start_frame, int encryption_type, const uint32_t *d_seed_bitmap,
                          const int *d_mults, int nmult, const int *d_incs, int ninc, int *d_out, int cap, int *d_count,
                          hipStream_t stream)
{
    const int64_t threads = (int64_t)0x1000 * nmult;
    if (threads <= 0 || ninc <= 0) return VGA_OK;
    hipLaunchKernelGGL(adx_guess_keys_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, d_scales, nscales,
                       start_frame, encryption_type == 8 ? 0xE000 : 0x1000, encryption_type == 8 ? 0x8000 : 0x2000, d_seed_bitmap,
                       d_mults, nmult, d_incs, ninc, d_out, cap, d_count);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_hca_find_key(const uint8_t *d_frames, int frame_count, const hca::DeviceInfo &info, const uint8_t *d_tables, int nkeys,
                        int *d_first /* scratch int */, int *d_valid, int *d_flags, hipStream_t stream)
{
    if (nkeys <= 0 || frame_count <= 0) return VGA_OK;
    VGA_HIP_TRY(hipMemsetAsync(d_first, 0x7f, sizeof(int), stream));
    VGA_HIP_TRY(hipMemsetAsync(d_flags, 0, sizeof(int), stream));
    hipLaunchKernelGGL(hca_first_non_empty_kernel, dim3(frame_count), dim3(64), 0, stream, d_frames, frame_count, info.frame_size, d_first);
    VGA_HIP_TRY(hipGetLastError());
    int first = 0;
    VGA_HIP_TRY(hipMemcpyAsync(&first, d_first, sizeof(int), hipMemcpyDeviceToHost, stream));
    VGA_HIP_TRY(hipStreamSynchronize(stream));
    if (first < 0 || first >= frame_count) first = 0;               // every frame empty: the reference tests from frame 0
    const int ntest = frame_count - first < 10 ? frame_count - first : 10;      // FramesToTest (:9)
    hipLaunchKernelGGL(hca_test_keys_kernel, dim3(nkeys), dim3(64), 0, stream, d_frames, first, ntest, info, d_tables, d_valid, d_flags);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

int launch_hca_byte_position_counts(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                                    int positions, unsigned *d_counts, hipStream_t stream)
{
    VGA_HIP_TRY(hipMemsetAsync(d_counts, 0, (size_t)positions * 256 * sizeof(unsigned), stream));
    const int64_t total = (int64_t)nstreams * frame_count;
    if (total <= 0) return VGA_OK;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(hca_byte_position_counts_kernel, dim3(blocks), dim3(256), (size_t)positions * 256 * sizeof(unsigned), stream,
                       d_frames, frames_pitch, nstreams, frame_count, frame_size, positions, d_counts);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace crypt
}  // namespace vga
