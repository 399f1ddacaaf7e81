// hca_decode_kernels.hip -- CRI HCA decoder for gfx950.
//
// Replaces VGAudio/Codecs/CriHca/CriHcaDecoder.cs:11-192 and the unpack half of
// VGAudio/Codecs/CriHca/CriHcaPacking.cs:10-229 (+ Utilities/BitReader.cs, Mdct.RunImdct Mdct.cs:94-119).
//
// Two kernels, because the two halves want opposite shapes:
//   hca_unpack_kernel : the frame bitstream is variable-length coded -> serial inside a frame, but
//                       every frame of every stream is independent.  lane = frame, 64 frames per wave;
//                       control flow is uniform across lanes (same band counts), only data differs.
//                       Output: scale factors, resolutions, intensity/HFR scales and the quantised
//                       spectra (int16) in a per-frame record in HBM.
//   hca_imdct_kernel  : workgroup = (stream, frame): dequantise, high-frequency reconstruction,
//                       intensity stereo, 128-point DCT-IV (exact staged butterflies), window +
//                       overlap-add, PCM16.  The IMDCT overlap (`_imdctPrevious`) is the only
//                       inter-frame state; instead of carrying it, each workgroup recomputes the
//                       previous frame's last sub-frame (9 transforms per channel instead of 8), so
//                       all frames stay independent.
// A frame whose scale-factor delta decoding fails keeps stale state in the reference
// (UnpackFrameHeader returns false, CriHcaPacking.cs:84); that is sequential state a lane-per-frame
// decoder cannot reproduce -- such frames (corrupt streams only) are flagged in *status (bit 1), an
// invalid sync word in bit 0 (the reference throws InvalidDataException).
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_kernels.hpp"

namespace vga {
namespace hca {

// ---- per-frame record written by the unpacker ------------------------------------------------
// [0] noise level (u16) [2] evaluation boundary (u8) [3] flags (u8); then per channel 272 bytes:
// scale factors[128], resolutions[128], intensity[8], hfr scales[8]; then int16 q[8][nch][128].
__host__ __device__ inline size_t record_channel_offset(int c) { return 16 + (size_t)c * 272; }
// the quantised spectra start on a 128-byte boundary of a record whose size is a multiple of 128: the unpacker hands
// them over in whole 128-byte runs (see flush_spectra)
__host__ __device__ inline size_t record_q_offset(int nch) { return (16 + (size_t)nch * 272 + 127) / 128 * 128; }
size_t unpack_record_bytes(int nch) { return record_q_offset(nch) + (size_t)8 * nch * 128 * 2; }

// MSB-first bit reader over the stream's (4-byte aligned) frame data: a 64-bit register window plus the next
// dword, which is fetched as soon as the position is known -- every symbol's length depends on the previous
// symbol, so without the look-ahead each peek is two dependent memory round trips.  The fetch is
// unconditional (same address again when no dword boundary was crossed: an L1 hit).
struct BitCursor {
    const uint32_t *base;   // 4-byte aligned start of the stream's frame data
    int64_t frame_bit0;     // absolute bit position of this frame inside the stream
    int64_t last_word;      // highest dword index that may be read (frames_pitch / 4 - 1)
    int frame_bits;
    int pos;                // bit position inside the frame (BitReader.Position)
    int64_t w;              // dword index of the window's first dword
    uint32_t hi, lo, nxt;   // base[w], base[w+1], base[w+2], byte-swapped

    __device__ __forceinline__ uint32_t fetch(int64_t i) const { return __builtin_bswap32(base[i < last_word ? i : last_word]); }
    __device__ __forceinline__ void start()
    {
        w = frame_bit0 >> 5;
        hi = fetch(w);
        lo = fetch(w + 1);
        nxt = fetch(w + 2);
    }
    // BitReader.PeekInt (BitReader.cs:51-92): MSB-first; bits past the end of the frame read as 0
    __device__ __forceinline__ int peek(int bits) const
    {
        if (bits == 0) return 0;
        const int sh = (int)((frame_bit0 + pos) & 31);
        const uint64_t win = ((uint64_t)hi << 32) | lo;
        int v = (int)((win << sh) >> (64 - bits));
        const int avail = frame_bits - pos;
        if (bits > avail) {
            if (avail <= 0) return 0;
            v = (v >> (bits - avail)) << (bits - avail);
        }
        return v;
    }
    __device__ __forceinline__ void skip(int bits)     // bits <= 32
    {
        pos += bits;
        const int64_t nw = (frame_bit0 + pos) >> 5;
        const bool adv = nw != w;
        hi = adv ? lo : hi;
        lo = adv ? nxt : lo;
        w = nw;
        nxt = fetch(nw + 2);
    }
    __device__ __forceinline__ int read(int bits)
    {
        const int v = peek(bits);
        skip(bits);
        return v;
    }
};

// Tried and dropped (round 2): copying the wave's 64 frames into LDS first, so that the bit readers walk LDS instead of
// issuing a vector load per symbol whose 64 lanes sit in 64 different cache lines.  The 44 KB of frames leave two waves
// per CU instead of six, and the chain per symbol (window shift, table look-up, advance) is long enough that the lost
// latency hiding costs more than the loads did: 67.6 ms instead of 39.3 at config 4.
__global__ __launch_bounds__(64) void hca_unpack_kernel(
    const uint8_t *__restrict__ frames, int64_t stream_pitch, int nstreams, DeviceInfo info,
    uint8_t *__restrict__ records, size_t record_bytes, int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_res[];   // [nch][128][64]
    __shared__ DecTables T;
    __shared__ uint4 s_stage[64][9];                                  // [lane][piece], 144-byte rows (bank spread)
    const int lane = threadIdx.x;
    load_tables(T, lane, 64);
    const int64_t gid = (int64_t)blockIdx.x * 64 + lane;
    const int64_t total = (int64_t)nstreams * info.frame_count;
    const bool live = gid < total;
    const int64_t id = live ? gid : total - 1;
    const int stream = (int)(id / info.frame_count);
    const int frame = (int)(id % info.frame_count);
    const int nch = info.nch;

    BitCursor r;
    r.base = reinterpret_cast<const uint32_t *>(frames + (int64_t)stream * stream_pitch);
    r.frame_bit0 = (int64_t)frame * info.frame_size * 8;
    r.last_word = stream_pitch / 4 - 1;
    __syncthreads();
    r.frame_bits = info.frame_size * 8;
    r.pos = 0;
    r.start();
    uint8_t *rec = records + (size_t)id * record_bytes;

    int flags = 0;
    if (r.read(16) != 0xffff) flags |= 1;
    const int noise_level = r.read(9);
    const int eval_boundary = r.read(7);

    for (int c = 0; c < nch; c++) {
        uint8_t *rc = rec + record_channel_offset(c);
        const int count = info.coded_count[c];
        // ReadScaleFactors / DeltaDecode (CriHcaPacking.cs:111-130, :185-211)
        const int delta_bits = r.read(3);
        int prev = 0;
        bool failed = false;
        const int max_delta = delta_bits > 0 ? 1 << (delta_bits - 1) : 0;
        uint32_t sf_pack[4] = {0, 0, 0, 0}, res_pack[4] = {0, 0, 0, 0};
        for (int i = 0; i < 128; i++) {
            int sf = 0;
            if (i < count && delta_bits != 0) {
                if (delta_bits >= 6 || i == 0) {
                    sf = r.read(6);
                } else if (!failed) {
                    const int delta = r.peek(delta_bits) - (max_delta - 1);   // ReadOffsetBinary, positive bias
                    r.skip(delta_bits);
                    if (delta < max_delta) {
                        sf = prev + delta;
                        if (sf < 0 || sf > 63) { failed = true; sf = 0; }
                    } else {
                        sf = r.read(6);
                    }
                }
                prev = sf;
            }
            // delta_bits == 0: Array.Clear of ALL 128 scale factors (:114-118); >= count stay 0 here
            int res = 0;
            if (i < count) {
                const int noise = info.ath_curve[i] + noise_level - (i < eval_boundary ? 1 : 0);
                res = calculate_resolution(T, sf, noise);
            }
            s_res[((size_t)c * 128 + i) * 64 + lane] = (uint8_t)res;
            // the record's byte arrays leave as 16-byte stores (one per 16 bands instead of 32 byte stores)
            sf_pack[(i >> 2) & 3] |= (uint32_t)sf << (8 * (i & 3));
            res_pack[(i >> 2) & 3] |= (uint32_t)res << (8 * (i & 3));
            if ((i & 15) == 15) {
                if (live) {
                    *reinterpret_cast<uint4 *>(rc + (i & ~15)) = make_uint4(sf_pack[0], sf_pack[1], sf_pack[2], sf_pack[3]);
                    *reinterpret_cast<uint4 *>(rc + 128 + (i & ~15)) = make_uint4(res_pack[0], res_pack[1], res_pack[2], res_pack[3]);
                }
#pragma unroll
                for (int k = 0; k < 4; k++) sf_pack[k] = res_pack[k] = 0;
            }
        }
        if (failed) flags |= 2;
        if (info.channel_type[c] == CH_STEREO_SECONDARY) {
            for (int i = 0; i < 8; i++) { const int v = r.read(4); if (live) rc[256 + i] = (uint8_t)v; }
        } else if (info.hfr_group_count > 0) {
            for (int i = 0; i < info.hfr_group_count; i++) { const int v = r.read(6); if (live) rc[264 + i] = (uint8_t)v; }
        }
    }

    // ReadSpectralCoefficients (:148-183).  A lane's eight int16 values are one 16-byte piece; written straight to its
    // record, the 64 lanes of a store hit 64 different records 4.7 KB apart (measured: 48 GB of HBM writes for 7.5 GB of
    // records).  The pieces go through LDS instead: after eight of them (64 coefficients) the wave writes 128 contiguous
    // bytes per record, eight lanes per record.
    const size_t q_off = record_q_offset(nch);
    const int64_t first = (int64_t)blockIdx.x * 64;
    auto flush_spectra = [&](size_t row_off, int s0, int pieces) {
        // the workgroup is one wave, whose LDS operations execute in program order: the hand-over needs the LDS counter
        // and a compiler ordering point, not s_barrier (and not __syncthreads(), which would also wait for the
        // bit reader's prefetched dword)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int rr = k * 8 + (lane >> 3), piece = lane & 7;
            if (piece < pieces && first + rr < total) {
                const uint4 v = s_stage[rr][piece];
                *reinterpret_cast<uint4 *>(records + (size_t)(first + rr) * record_bytes + q_off + row_off + (size_t)s0 * 2 +
                                           (size_t)piece * 16) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the pieces are in registers before the next ones overwrite them
    };
    for (int sf = 0; sf < SUBFRAMES; sf++) {
        for (int c = 0; c < nch; c++) {
            const int count = info.coded_count[c];
            const size_t row_off = ((size_t)sf * nch + c) * 128 * 2;
            uint32_t qp[4] = {0, 0, 0, 0};
            for (int s = 0; s < count; s++) {
                const int resolution = s_res[((size_t)c * 128 + s) * 64 + lane];
                int bits = T.max_bits[resolution];
                const int code = r.peek(bits);
                int value;
                if (resolution < 8) {
                    bits = T.dec_bits[resolution][code];
                    value = T.dec_value[resolution][code];
                } else {
                    value = code / 2 * (1 - (code % 2 * 2));
                    if (value == 0) bits--;
                }
                r.skip(bits);
                qp[(s >> 1) & 3] |= (uint32_t)(value & 0xFFFF) << (16 * (s & 1));
                const bool last = s == count - 1;
                if ((s & 7) == 7 || last) {                 // a piece is complete (the row's last one may be partly zeros)
                    s_stage[lane][(s >> 3) & 7] = make_uint4(qp[0], qp[1], qp[2], qp[3]);
#pragma unroll
                    for (int k = 0; k < 4; k++) qp[k] = 0;
                    if ((s & 63) == 63 || last) flush_spectra(row_off, s & ~63, ((s & 63) >> 3) + 1);
                }
            }
        }
    }
    if (live) {
        rec[0] = (uint8_t)(noise_level & 0xff);
        rec[1] = (uint8_t)(noise_level >> 8);
        rec[2] = (uint8_t)eval_boundary;
        rec[3] = (uint8_t)flags;
        if (flags && status) atomicOr(status, flags);
    }
}

// ---- dequantise + IMDCT ----------------------------------------------------------------------
// LDS: spec[nch][9][128] f64 (slot 0 = previous frame's sub-frame 7, slots 1..8 = this frame),
//      tmp[9][128] f64 butterfly scratch, dct[9][128] f64 per channel pass.
// 288 threads = nine 32-lane groups: a channel's nine transforms (eight sub-frames + the previous frame's last
// one for the overlap) run side by side.
constexpr int IMDCT_THREADS = 288;
__global__ __launch_bounds__(IMDCT_THREADS) void hca_imdct_kernel(
    const uint8_t *__restrict__ records, size_t record_bytes, int nstreams, DeviceInfo info,
    int16_t *__restrict__ pcm, int64_t stream_pitch, int64_t ch_pitch)
{
    extern __shared__ __attribute__((aligned(16))) double s_mem[];
    __shared__ DecTables T;
    const int nch = info.nch;
    // LDS decides the occupancy (4 workgroups per CU need <= 40 KB each): the transform permutes in its input
    double *spec = s_mem;                               // [nch][9][128]
    double *dct = spec + (size_t)nch * 9 * 128;         // [9][128]
    double *gain = dct + 9 * 128;                       // [2][nch][128]: previous frame, this frame

    const int tid = threadIdx.x;
    load_tables(T, tid, IMDCT_THREADS);
    __syncthreads();
    const int stream = blockIdx.x / info.frame_count;
    const int frame = blockIdx.x % info.frame_count;
    const uint8_t *rec_cur = records + ((size_t)stream * info.frame_count + frame) * record_bytes;
    const uint8_t *rec_prev = frame > 0 ? rec_cur - record_bytes : nullptr;

    // CalculateGain (CriHcaDecoder.cs:108-114)
    for (int i = tid; i < 2 * nch * 128; i += IMDCT_THREADS) {
        const int which = i / (nch * 128), c = (i / 128) % nch, s = i % 128;
        const uint8_t *rec = which == 0 ? rec_prev : rec_cur;
        double g = 0.0;
        if (rec && s < info.coded_count[c]) {
            const uint8_t *rc = rec + record_channel_offset(c);
            g = T.dequant_scale[rc[s]] * T.step[rc[128 + s]];
        }
        gain[i] = g;
    }
    __syncthreads();

    // DequantizeFrame (:83-100): spectra = q * gain; bands >= coded count are zero (:180)
    for (int i = tid; i < nch * 9 * 128; i += IMDCT_THREADS) {
        const int c = i / (9 * 128), slot = (i / 128) % 9, s = i % 128;
        const uint8_t *rec = slot == 0 ? rec_prev : rec_cur;
        const int sf = slot == 0 ? 7 : slot - 1;
        double v = 0.0;
        if (rec && s < info.coded_count[c]) {
            const int16_t *q = reinterpret_cast<const int16_t *>(rec + record_q_offset(nch));
            v = (double)(int)q[((size_t)sf * nch + c) * 128 + s] * gain[((slot == 0 ? 0 : 1) * nch + c) * 128 + s];
        }
        spec[i] = v;
    }
    __syncthreads();

    // ReconstructHighFrequency (:116-145)
    if (info.hfr_group_count > 0) {
        const int total_band_count = min(info.total_band_count, 127);
        const int hfr_start = info.base_band_count + info.stereo_band_count;
        const int hfr_bands = min(info.hfr_band_count, total_band_count - info.hfr_band_count);
        for (int i = tid; i < nch * 9 * hfr_bands; i += IMDCT_THREADS) {
            const int c = i / (9 * hfr_bands), slot = (i / hfr_bands) % 9, band = i % hfr_bands;
            if (info.channel_type[c] == CH_STEREO_SECONDARY) continue;
            const uint8_t *rec = slot == 0 ? rec_prev : rec_cur;
            if (!rec) continue;
            const int group = band / info.bands_per_hfr_group;
            if (group >= info.hfr_group_count) continue;
            const uint8_t *rc = rec + record_channel_offset(c);
            const int high = hfr_start + band, low = hfr_start - band - 1;
            const int index = (int)rc[264 + group] - (int)rc[low] + 64;
            double *sp = spec + ((size_t)c * 9 + slot) * 128;
            sp[high] = f64_bits(HCA_ScaleConversionTableBits[index & 127]) * sp[low];
        }
        __syncthreads();
    }
    // ApplyIntensityStereo (:147-166)
    if (info.stereo_band_count > 0) {
        const int nb = info.total_band_count - info.base_band_count;
        for (int i = tid; i < nch * 9 * nb; i += IMDCT_THREADS) {
            const int c = i / (9 * nb), slot = (i / nb) % 9, b = info.base_band_count + i % nb;
            if (info.channel_type[c] != CH_STEREO_PRIMARY) continue;
            const uint8_t *rec = slot == 0 ? rec_prev : rec_cur;
            if (!rec) continue;
            const int sf = slot == 0 ? 7 : slot - 1;
            const int iq = rec[record_channel_offset(c + 1) + 256 + sf];
            const double ratio_l = f64_bits(HCA_IntensityRatioTableBits[min(iq, 14)]);
            const double ratio_r = ratio_l - 2.0;
            double *l = spec + ((size_t)c * 9 + slot) * 128;
            double *rr = spec + ((size_t)(c + 1) * 9 + slot) * 128;
            const double lv = l[b];
            rr[b] = lv * ratio_r;
            l[b] = lv * ratio_l;
        }
        __syncthreads();
    }

    // RunImdct (:168-177 -> Mdct.cs:94-119) + PcmFloatToShort (:179-192) + CopyPcmToOutput (:31-45)
    const int grp = tid >> 5, t = tid & 31;
    for (int c = 0; c < nch; c++) {
        double *sp = spec + (size_t)c * 9 * 128;
        // 9 transforms, one per 32-lane group: slot 0 = the previous frame's sub-frame 7, slots 1..8 = this frame
        dct4_128(T, sp + (size_t)grp * 128, sp + (size_t)grp * 128, dct + (size_t)grp * 128, t, wave_sync);
        __syncthreads();
        // window + overlap-add: out(slot) needs `previous` produced from slot-1's transform
        int16_t *dst = pcm + (int64_t)stream * stream_pitch + (int64_t)c * ch_pitch;
        for (int i = tid; i < 8 * 128; i += IMDCT_THREADS) {
            const int slot = 1 + i / 128, j = i % 128;
            const double *dc = dct + (size_t)slot * 128;        // this sub-frame's dctOut
            const double *dp = dct + (size_t)(slot - 1) * 128;  // the one before
            const bool have_prev = slot > 1 || frame > 0;
            double out;
            if (j < 64) {
                const double prev = have_prev ? T.window[127 - j] * -dp[63 - j] : 0.0;       // _imdctPrevious[i]
                out = T.window[j] * dc[j + 64] + prev;
            } else {
                const int k = j - 64;
                const double prev = have_prev ? T.window[63 - k] * dp[k] : 0.0;              // _imdctPrevious[i+half]
                out = T.window[k + 64] * -dc[127 - k] - prev;
            }
            // (int)(x * 32768): RyuJIT cvttsd2si semantics, then Clamp16
            const double scaled = out * 32768.0;
            int sample = (scaled > -2147483649.0 && scaled < 2147483648.0) ? (int)scaled : (int)0x80000000;
            sample = min(max(sample, -32768), 32767);
            const int64_t tpos = (int64_t)frame * SPF + (slot - 1) * SPSF + j - info.inserted_samples;
            if (tpos >= 0 && tpos < info.sample_count) dst[tpos] = (int16_t)sample;
        }
        __syncthreads();
    }
}

int launch_decode(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, const DeviceInfo &info, int16_t *d_pcm,
                  int64_t stream_pitch, int64_t ch_pitch, void *d_workspace, int *d_status, hipStream_t stream)
{
    if (nstreams <= 0 || info.frame_count <= 0) return VGA_OK;
    const size_t rb = unpack_record_bytes(info.nch);
    const int64_t total = (int64_t)nstreams * info.frame_count;
    const size_t lds1 = (size_t)info.nch * 128 * 64;
    if (lds1 > 48 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_unpack_kernel, lds1));
    hipLaunchKernelGGL(hca_unpack_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), lds1, stream, d_frames,
                       frames_pitch, nstreams, info, reinterpret_cast<uint8_t *>(d_workspace), rb, d_status);
    const size_t lds2 = ((size_t)info.nch * 9 * 128 + 9 * 128 + (size_t)2 * info.nch * 128) * sizeof(double);
    if (lds2 > 64 * 1024)
        VGA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(hca_imdct_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(hca_imdct_kernel, dim3((unsigned)total), dim3(IMDCT_THREADS), lds2, stream,
                       reinterpret_cast<const uint8_t *>(d_workspace), rb, nstreams, info, d_pcm, stream_pitch, ch_pitch);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
