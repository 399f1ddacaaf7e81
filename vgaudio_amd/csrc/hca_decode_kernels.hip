// hca_decode_kernels.hip -- CRI HCA decoder for gfx950.
//
// Replaces VGAudio/Codecs/CriHca/CriHcaDecoder.cs:11-192 and the unpack half of
// VGAudio/Codecs/CriHca/CriHcaPacking.cs:10-229 (+ Utilities/BitReader.cs, Mdct.RunImdct Mdct.cs:94-119).
//
// Two kernels (per-lane logic in hca_decode_core.hpp, which the CPU suite also compiles into a lane emulator):
//   hca_scan_kernel   : lane = frame, 64 frames per wave.  A frame is variable-length coded, so where a code starts is a
//                       serial walk -- but only the code LENGTHS are needed for it.  The lane reads its frame's header
//                       (scale factors, intensity / HFR scales), then walks the 8 x nch x count spectral codes
//                       length-only and notes the bit offset of every 16th.  The bitstream reaches the lane through a
//                       ring of 16 dwords in LDS (16-byte global loads, issued a block of eight codes ahead) and a
//                       128-bit register window that is re-aligned every eight codes, so no memory access sits on the
//                       per-code dependency chain (round 2's unpacker: one dependent global load per code, 39 ms).
//                       Hand-over per frame: scale factors + chunk offsets, 576 bytes for a stereo frame of 128 bands
//                       (round 2: the quantised spectra as int16, 4.7 KB per frame = 17 GB written and read again).
//   hca_frames_kernel : workgroup = up to 16 consecutive frames of one stream, 128 threads.  Per frame: the frame's bytes
//                       and its record go to LDS; each 16-code chunk is decoded by its own lane (value, dequantised,
//                       stored in the transform's input layout); the 128-point DCT-IV (exact staged butterflies) runs
//                       on 8 lanes per transform with every lane-dependent twiddle in registers; window + overlap-add +
//                       PCM16.  The IMDCT overlap (`_imdctPrevious`) is carried from frame to frame in LDS rows; the
//                       first frame of a run recomputes the last sub-frame of the frame before it.
// A frame whose scale-factor delta decoding fails keeps stale state in the reference
// (UnpackFrameHeader returns false, CriHcaPacking.cs:84); that is sequential state a frame-parallel
// decoder cannot reproduce -- such frames (corrupt streams only) are flagged in *status (bit 1), an
// invalid sync word in bit 0 (the reference throws InvalidDataException).
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_decode_core.hpp"
#include "hca_kernels.hpp"

namespace vga {
namespace hca {

size_t decode_record_bytes(const DeviceInfo &info) { return (size_t)make_decode_layout(info).record_bytes; }

namespace {

struct __attribute__((packed, aligned(4))) Dwords4 {
    uint32_t v[4];
};

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ---- scan -------------------------------------------------------------------------------------
struct ScanTables {
    Symbol sym[16];
    uint8_t curve[64];
    __device__ __forceinline__ uint64_t symbol_len(int r) const { return sym[r].len; }
    __device__ __forceinline__ const uint8_t *res_curve() const { return curve; }
};

struct GlobalSrc {
    const uint32_t *base;              // the frame's first aligned dword
    int limit;                         // dwords readable from there (to the end of the stream's pitch)
    __device__ __forceinline__ void quad(int k, uint32_t out[4]) const
    {
        if (k + 3 < limit) {
            const Dwords4 v = *reinterpret_cast<const Dwords4 *>(base + k);
#pragma unroll
            for (int e = 0; e < 4; e++) out[e] = v.v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) out[e] = base[min(k + e, limit - 1)];
        }
    }
};
struct LdsRing {
    uint32_t *p;                       // ring[slot * 64 + lane]
    __device__ __forceinline__ void put(int slot, uint32_t v) { p[slot * 64] = v; }
    __device__ __forceinline__ uint32_t get(int slot) const { return p[slot * 64]; }
};
struct LdsRes {
    uint32_t *p;                       // res[(c * 16 + word) * 64 + lane]
    __device__ __forceinline__ void put(int c, int w, uint32_t v) { p[(c * 16 + w) * 64] = v; }
    __device__ __forceinline__ uint32_t get(int c, int w) const { return p[(c * 16 + w) * 64]; }
};
// Record pieces (16 bytes each, in record order) are staged per lane and leave four at a time: 64 contiguous bytes per
// record, four lanes per record (a lane's own 16-byte stores would touch 64 records 576 bytes apart per instruction).
struct StagedOut {
    uint4 (*stage)[5];                 // [lane][piece], 80-byte rows
    uint8_t *records;                  // record of the wave's first frame
    size_t record_bytes;
    int lane, live_records;            // records [0, live_records) of this wave exist
    int count;
    __device__ __forceinline__ void flush(int group, int pieces)
    {
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int rr = k * 16 + (lane >> 2), piece = lane & 3;
            if (piece < pieces && rr < live_records)
                *reinterpret_cast<uint4 *>(records + (size_t)rr * record_bytes + (size_t)group * 64 + piece * 16) = stage[rr][piece];
        }
        wave_lds_sync();
    }
    __device__ __forceinline__ void piece(const uint32_t v[4])
    {
        stage[lane][count & 3] = make_uint4(v[0], v[1], v[2], v[3]);
        if ((count & 3) == 3) flush(count >> 2, 4);
        count++;
    }
    __device__ __forceinline__ void finish()
    {
        if (count & 3) flush(count >> 2, count & 3);
    }
};

}  // namespace

__global__ __launch_bounds__(64) void hca_scan_kernel(const uint8_t *__restrict__ frames, int64_t stream_pitch, int nstreams,
                                                      DeviceInfo info, DecodeLayout lay, uint8_t *__restrict__ records,
                                                      int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];   // res[nch][16][64]
    __shared__ ScanTables T;
    __shared__ uint32_t s_ring[16 * 64];
    __shared__ uint4 s_stage[64][5];
    __shared__ int s_coded[8], s_type[8];
    __shared__ uint8_t s_ath[128];
    const int lane = threadIdx.x;
    if (lane < 16)
        T.sym[lane] = make_symbol(lane, HCA_QuantizedSpectrumBits[lane & 7], HCA_QuantizedSpectrumValue[lane & 7],
                                  HCA_QuantizedSpectrumMaxBits[lane]);
    T.curve[lane] = lane < 59 ? HCA_ScaleToResolutionCurve[lane] : 0;
    if (lane < 8) {
        s_coded[lane] = info.coded_count[lane];
        s_type[lane] = info.channel_type[lane];
    }
    s_ath[lane] = info.ath_curve[lane];
    s_ath[lane + 64] = info.ath_curve[lane + 64];
    __syncthreads();

    const int64_t first = (int64_t)blockIdx.x * 64;
    const int64_t total = (int64_t)nstreams * info.frame_count;
    const int64_t gid = first + lane;
    const bool live = gid < total;
    const int64_t id = live ? gid : total - 1;
    const int stream = (int)(id / info.frame_count);
    const int frame = (int)(id % info.frame_count);
    const int64_t a0 = (int64_t)frame * info.frame_size;               // byte offset of the frame inside its stream

    GlobalSrc src;
    src.base = reinterpret_cast<const uint32_t *>(frames + (int64_t)stream * stream_pitch) + (a0 >> 2);
    src.limit = (int)(stream_pitch / 4 - (a0 >> 2));
    LdsRing ring{s_ring + lane};
    LdsRes res{s_dyn + lane};
    StagedOut out;
    out.stage = s_stage;
    out.records = records + (size_t)first * lay.record_bytes;
    out.record_bytes = (size_t)lay.record_bytes;
    out.lane = lane;
    out.live_records = (int)min((int64_t)64, total - first);
    out.count = 0;
    ScanParams P;
    P.nch = info.nch;
    P.frame_bits = info.frame_size * 8;
    P.first_bit = (int)(a0 & 3) * 8;
    P.hfr_group_count = info.hfr_group_count;
    P.coded_count = s_coded;
    P.channel_type = s_type;
    P.ath_curve = s_ath;
    P.wide_offsets = lay.wide_offsets;
    const int flags = scan_frame(P, src, ring, res, out, T);
    out.finish();
    if (live && flags && status) atomicOr(status, flags);
}

// ---- frames -----------------------------------------------------------------------------------
namespace {

constexpr int FRAMES_THREADS = 128;
constexpr int MAX_FRAMES_PER_GROUP = 32;         // (configs[3]: 8 / 16 / 32 / 64 frames a workgroup = 24.5 / 23.6 / 23.1 / 23.0 ms for the decode)
// timing-only builds (tools/build_variants.sh dstopN:"-DVGA_HCA_DEC_STOP_AFTER=N", wrong output): the frame loop ends after
// 1 = the frame's dwords and record are in LDS, 2 = resolutions and gains, 3 = stage A (codes read, dequantised),
// 4 = stage B (the transforms); 99 = the product.  Round 6 at configs[3] (profiles/r06_y_hca_decode_stages.log), of the
// launch's 18.8 ms: load 3.2, resolutions + gains 1.0, stage A 6.2, stage B 3.7, stage C 4.7.
#ifndef VGA_HCA_DEC_STOP_AFTER
#define VGA_HCA_DEC_STOP_AFTER 99
#endif

struct FramesTables {
    Symbol sym[16];
    double dequant_scale[64];          // DequantizerScalingTable
    double step[16];                   // QuantizerStepSize
    uint8_t curve[64];
    uint8_t ath[128];
    int coded[8], type[8], chunk_base[8];
    __device__ __forceinline__ const Symbol &symbol(int r) const { return sym[r]; }
    __device__ __forceinline__ const uint8_t *res_curve() const { return curve; }
};
struct LdsFrame {
    const uint32_t *w;
    int zero_at;
    __device__ __forceinline__ uint32_t get(int k) const { return w[min(k, zero_at)]; }
};
struct Res16 {
    uint4 v;
    __device__ __forceinline__ int operator[](int e) const
    {
        const uint32_t w = e < 4 ? v.x : e < 8 ? v.y : e < 12 ? v.z : v.w;
        return (int)((w >> (8 * (e & 3))) & 255u);
    }
};

}  // namespace

__global__ __launch_bounds__(FRAMES_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void hca_frames_kernel(
    const uint8_t *__restrict__ frames, int64_t frames_pitch, DeviceInfo info, DecodeLayout lay,
    const uint8_t *__restrict__ records, int frames_per_group, int groups_per_stream, int16_t *__restrict__ pcm,
    int64_t stream_pitch, int64_t ch_pitch)
{
    extern __shared__ __attribute__((aligned(16))) char s_mem[];
    __shared__ FramesTables T;
    const int nch = info.nch;
    // LDS: rows[nch][9] of ROW_BYTES | gain[nch][128] f64 | res[nch][128] u8 | frame dwords (+1 zero) | record
    char *rows = s_mem;
    double *s_gain = reinterpret_cast<double *>(rows + (size_t)nch * 9 * ROW_BYTES);
    uint8_t *s_res = reinterpret_cast<uint8_t *>(s_gain + nch * 128);
    uint32_t *s_fb = reinterpret_cast<uint32_t *>(s_res + nch * 128);
    uint8_t *s_rec = reinterpret_cast<uint8_t *>(s_fb + (lay.frame_dwords + 1 + 3) / 4 * 4);

    const int tid = threadIdx.x;
    if (tid < 16)
        T.sym[tid] = make_symbol(tid, HCA_QuantizedSpectrumBits[tid & 7], HCA_QuantizedSpectrumValue[tid & 7],
                                 HCA_QuantizedSpectrumMaxBits[tid]);
    if (tid < 64) {
        T.dequant_scale[tid] = f64_bits(HCA_DequantizerScalingTableBits[tid]);
        T.curve[tid] = tid < 59 ? HCA_ScaleToResolutionCurve[tid] : 0;
    }
    if (tid < 16) T.step[tid] = f64_bits(HCA_QuantizerStepSizeBits[tid]);
    T.ath[tid] = info.ath_curve[tid];
    if (tid < 8) {
        T.coded[tid] = info.coded_count[tid];
        T.type[tid] = info.channel_type[tid];
        T.chunk_base[tid] = lay.chunk_base[tid];
    }
    // loop-invariant per lane: the transform's twiddles and the window values of its output sample
    const int L = tid & 7;
    const DctLane K = make_dct_lane(MDCT_SinBits, MDCT_CosBits, L);
    const DctUniform U = make_dct_uniform(MDCT_SinBits, MDCT_CosBits);
    const double w_cur = (double)__uint_as_float(HCA_MdctWindowF32Bits[tid]);
    const double w_prev = (double)__uint_as_float(HCA_MdctWindowF32Bits[127 - tid]);
    const int cur_at = 8 * imdct_cur_index(tid), prev_at = 8 * imdct_prev_index(tid);

    const int stream = blockIdx.x / groups_per_stream;
    const int f0 = (blockIdx.x % groups_per_stream) * frames_per_group;
    const int f1 = min(f0 + frames_per_group, info.frame_count);
    const uint32_t *sbase = reinterpret_cast<const uint32_t *>(frames + (int64_t)stream * frames_pitch);
    const int64_t last_word = frames_pitch / 4 - 1;
    const uint8_t *srec = records + (size_t)stream * info.frame_count * lay.record_bytes;
    int16_t *spcm = pcm + (int64_t)stream * stream_pitch;
    const int frame_bits = info.frame_size * 8;

    // the next frame's first dwords and record, loaded a frame ahead (unconditional loads at clamped positions)
    constexpr int FETCH_AHEAD = 2;
    uint32_t fa[FETCH_AHEAD], fb[FETCH_AHEAD];
    uint4 frec;
    auto fetch = [&](int f) {
        const int64_t w0 = ((int64_t)f * info.frame_size) >> 2;
#pragma unroll
        for (int i = 0; i < FETCH_AHEAD; i++) {
            const int k = min(tid + i * FRAMES_THREADS, lay.frame_dwords - 1);
            fa[i] = sbase[min(w0 + k, last_word)];
            fb[i] = sbase[min(w0 + k + 1, last_word)];
        }
        frec = reinterpret_cast<const uint4 *>(srec + (size_t)f * lay.record_bytes)[min(tid, lay.record_bytes / 16 - 1)];
    };
    fetch(f0 > 0 ? f0 - 1 : f0);
    __syncthreads();                                   // T is written
    // Inside the frame loop the stages talk through LDS only: lds_barrier() (LDS counter + s_barrier) instead of
    // __syncthreads(), which would also wait for the loads of the next frame and for stage C's stores at every stage boundary.
    for (int f = f0 > 0 ? f0 - 1 : f0; f < f1; f++) {
        const bool warm = f < f0;                      // the frame before the run: only its last sub-frame, no output
        const int base = (9 - f % 9) % 9;              // sub-frame sf of frame f lives in slot (base + sf) % 9 of its channel
        lds_barrier();                               // the previous frame's stage C has read its rows; T is written
        {   // the frame's bytes as big-endian dwords starting at its first bit, zero past its end; its record.  The first
            // FETCH_AHEAD x 128 dwords and 128 x 16 bytes of the record were loaded during the frame before (fetch below).
            const int64_t a0 = (int64_t)f * info.frame_size;
            const int64_t w0 = a0 >> 2;
            const int sh8 = (int)(a0 & 3) * 8;
            auto put = [&](int k, uint32_t r0, uint32_t r1) {
                uint32_t v = 0;
                if (k < lay.frame_dwords) {
                    const uint32_t x0 = bswap32(r0), x1 = bswap32(r1);
                    v = sh8 ? (x0 << sh8) | (x1 >> (32 - sh8)) : x0;
                    v = mask_past_end(v, k, frame_bits);
                }
                s_fb[k] = v;
            };
#pragma unroll
            for (int i = 0; i < FETCH_AHEAD; i++)
                if (tid + i * FRAMES_THREADS <= lay.frame_dwords) put(tid + i * FRAMES_THREADS, fa[i], fb[i]);
            for (int k = tid + FETCH_AHEAD * FRAMES_THREADS; k <= lay.frame_dwords; k += FRAMES_THREADS)
                put(k, sbase[min(w0 + k, last_word)], sbase[min(w0 + k + 1, last_word)]);
            const uint4 *rec = reinterpret_cast<const uint4 *>(srec + (size_t)f * lay.record_bytes);
            if (tid < lay.record_bytes / 16) reinterpret_cast<uint4 *>(s_rec)[tid] = frec;
            for (int k = tid + FRAMES_THREADS; k < lay.record_bytes / 16; k += FRAMES_THREADS) reinterpret_cast<uint4 *>(s_rec)[k] = rec[k];
            fetch(min(f + 1, f1 - 1));                 // in flight during this frame's stages
        }
        lds_barrier();
        if (VGA_HCA_DEC_STOP_AFTER == 1) continue;
        {   // resolutions (CriHcaPacking.cs:85-93) and gains (CriHcaDecoder.cs:108-114)
            const uint32_t head = *reinterpret_cast<const uint32_t *>(s_rec + lay.header_at);
            const int noise = (int)(head & 0xFFFFu), eval = (int)((head >> 16) & 0xFFu);
            for (int b = tid; b < nch * 128; b += FRAMES_THREADS) {
                const int c = b >> 7, s = b & 127;
                const int sf = s_rec[c * REC_CHANNEL_BYTES + s];
                const int rs = s < T.coded[c] ? resolution_for(T.curve, sf, (int)T.ath[s] + noise - (s < eval ? 1 : 0)) : 0;
                s_res[b] = (uint8_t)rs;
                s_gain[b] = T.dequant_scale[sf] * T.step[rs];
            }
            if (f == 0)                                // the start of the stream: _imdctPrevious is cleared
                for (int i = tid; i < nch * (ROW_BYTES / 8); i += FRAMES_THREADS) {
                    const int c = i / (ROW_BYTES / 8), k = i % (ROW_BYTES / 8);
                    reinterpret_cast<double *>(rows + (size_t)(c * 9 + (base + 8) % 9) * ROW_BYTES)[k] = 0.0;
                }
        }
        lds_barrier();
        if (VGA_HCA_DEC_STOP_AFTER == 2) continue;
        // stage A: ReadSpectralCoefficients + DequantizeFrame, one 16-code chunk per lane
        for (int id = tid; id < nch * 64; id += FRAMES_THREADS) {
            const int row = id >> 3, q = id & 7, c = row >> 3, sf = row & 7;
            if (warm && sf != 7) continue;
            const int nsym = min(max(T.coded[c] - 16 * q, 0), 16);
            uint32_t off = 0;
            if (nsym > 0) {
                const int k = sf * lay.chunks_per_subframe + T.chunk_base[c] + q;
                off = lay.wide_offsets ? reinterpret_cast<const uint32_t *>(s_rec + lay.offsets_at)[k]
                                       : reinterpret_cast<const uint16_t *>(s_rec + lay.offsets_at)[k];
            }
            Res16 r16;
            r16.v = *reinterpret_cast<const uint4 *>(s_res + c * 128 + 16 * q);
            int slot = base + sf;
            slot = slot >= 9 ? slot - 9 : slot;
            decode_chunk(LdsFrame{s_fb, lay.frame_dwords}, (int)off, nsym, 16 * q, r16, s_gain + c * 128 + 16 * q, T,
                         rows + (size_t)(c * 9 + slot) * ROW_BYTES);
        }
        // ReconstructHighFrequency (CriHcaDecoder.cs:116-145)
        if (info.hfr_group_count > 0) {
            lds_barrier();
            const int total_band_count = min(info.total_band_count, 127);
            const int hfr_start = info.base_band_count + info.stereo_band_count;
            const int hfr_bands = min(info.hfr_band_count, total_band_count - info.hfr_band_count);
            for (int i = tid; i < nch * 8 * hfr_bands; i += FRAMES_THREADS) {
                const int c = i / (8 * hfr_bands), sf = (i / hfr_bands) % 8, band = i % hfr_bands;
                if (T.type[c] == CH_STEREO_SECONDARY || (warm && sf != 7)) continue;
                const int group = band / info.bands_per_hfr_group;
                if (group >= info.hfr_group_count) continue;
                const int high = hfr_start + band, low = hfr_start - band - 1;
                const int index = (int)s_rec[c * REC_CHANNEL_BYTES + 136 + group] - (int)s_rec[c * REC_CHANNEL_BYTES + low] + 64;
                char *r = rows + (size_t)(c * 9 + (base + sf) % 9) * ROW_BYTES;
                *reinterpret_cast<double *>(r + spec_byte_offset(high)) =
                    f64_bits(HCA_ScaleConversionTableBits[index & 127]) * *reinterpret_cast<const double *>(r + spec_byte_offset(low));
            }
        }
        // ApplyIntensityStereo (:147-166)
        if (info.stereo_band_count > 0) {
            lds_barrier();
            const int nb = info.total_band_count - info.base_band_count;
            for (int i = tid; i < nch * 8 * nb; i += FRAMES_THREADS) {
                const int c = i / (8 * nb), sf = (i / nb) % 8, b = info.base_band_count + i % nb;
                if (T.type[c] != CH_STEREO_PRIMARY || (warm && sf != 7)) continue;
                const int iq = s_rec[(c + 1) * REC_CHANNEL_BYTES + 128 + sf];
                const double ratio_l = f64_bits(HCA_IntensityRatioTableBits[min(iq, 14)]);
                const double ratio_r = ratio_l - 2.0;
                double *l = reinterpret_cast<double *>(rows + (size_t)(c * 9 + (base + sf) % 9) * ROW_BYTES + spec_byte_offset(b));
                double *rr = reinterpret_cast<double *>(rows + (size_t)((c + 1) * 9 + (base + sf) % 9) * ROW_BYTES + spec_byte_offset(b));
                const double lv = *l;
                *rr = lv * ratio_r;
                *l = lv * ratio_l;
            }
        }
        lds_barrier();
        if (VGA_HCA_DEC_STOP_AFTER == 3) continue;
        // stage B: RunImdct's Dct4 (Mdct.cs:126-181), 8 lanes per transform; all of a transform's lanes share a wave
        for (int row = tid >> 3; row < nch * 8; row += FRAMES_THREADS / 8) {
            const int c = row >> 3, sf = row & 7;
            if (warm && sf != 7) continue;
            int slot = base + sf;
            slot = slot >= 9 ? slot - 9 : slot;
            char *r = rows + (size_t)(c * 9 + slot) * ROW_BYTES;
            dct_first_half(r, L, K);
            wave_lds_sync();
            double y[16];
            dct_second_half(r, L, U, y);
            wave_lds_sync();
            dct_store(r, K, y);
        }
        lds_barrier();
        if (VGA_HCA_DEC_STOP_AFTER == 4) continue;
        // stage C: window + overlap-add (Mdct.cs:112-118), PcmFloatToShort, CopyPcmToOutput (CriHcaDecoder.cs:31-45).
        // (Round 6: a wave's 8 x 64 samples of a channel turned through 1 KB of LDS into one 16-byte store per lane instead of
        // these eight 2-byte stores per lane -- same samples, 26.9 ms for the decode against 24.8: the stage is bound by the
        // rows the memory system keeps open (11.8 GB in 4.7 ms), as the ADPCM decoders' stores are, not by the instruction.)
        if (!warm) {
            for (int c = 0; c < nch; c++) {
                int16_t *dst = spcm + (int64_t)c * ch_pitch;
                const char *chrows = rows + (size_t)c * 9 * ROW_BYTES;
                int slot = base;
                int pslot = base + 8 >= 9 ? base - 1 : base + 8;
#pragma unroll
                for (int sf = 0; sf < 8; sf++) {
                    const double cur = *reinterpret_cast<const double *>(chrows + slot * ROW_BYTES + cur_at);
                    const double prev = *reinterpret_cast<const double *>(chrows + pslot * ROW_BYTES + prev_at);
                    const int sample = imdct_sample(tid < 64, w_cur, w_prev, cur, prev);
                    const int64_t tpos = (int64_t)f * SPF + sf * SPSF + tid - info.inserted_samples;
                    if (tpos >= 0 && tpos < info.sample_count) dst[tpos] = (int16_t)sample;
                    pslot = slot;
                    slot = slot + 1 >= 9 ? 0 : slot + 1;
                }
            }
        }
    }
}

int launch_decode(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, const DeviceInfo &info, int16_t *d_pcm,
                  int64_t stream_pitch, int64_t ch_pitch, void *d_workspace, int *d_status, hipStream_t stream)
{
    if (nstreams <= 0 || info.frame_count <= 0) return VGA_OK;
    const DecodeLayout lay = make_decode_layout(info);
    const int64_t total = (int64_t)nstreams * info.frame_count;
    uint8_t *records = reinterpret_cast<uint8_t *>(d_workspace);
    const size_t lds1 = (size_t)info.nch * 16 * 64 * sizeof(uint32_t);
    if (lds1 > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_scan_kernel, lds1));
    hipLaunchKernelGGL(hca_scan_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), lds1, stream, d_frames, frames_pitch,
                       nstreams, info, lay, records, d_status);
    // (test hook as in launch_encode: n > 0 = frames per group; 1000 + n means the same here.  A wave-per-frame form of this
    // launch, the encoder's scheme, was built and measured in round 6 and is not used: 20.0 ms against this kernel's 18.7 --
    // 7.5 G VALU instructions at 0.61 of the issue slots, 13.7 KB of LDS a wave = two waves per SIMD, and the 16-code chunks'
    // serial VLC chains have nothing to overlap with inside one wave (tools/experiments/hca_decode_wave_kernel.hip,
    // profiles/r06_l_sq_counters_hca_decode_wave_experiment.json).)
    // Also measured and not used (round 6): the batch in two halves with the second half's scan on a side stream underneath
    // the first half's frames launch -- 24.7 ms, the same as back to back: the frames launch leaves the scan's waves no room.
    const int hook = hca_frames_per_group_override();
    const int group_override = hook >= 1000 ? hook - 1000 : hook;
    // frames per workgroup: long runs amortise the table set-up and the recomputed sub-frame before the run, short ones
    // keep small inputs spread over the chip
    int per_group = (int)std::min<int64_t>(MAX_FRAMES_PER_GROUP, std::max<int64_t>(1, total / 8192));
    if (group_override > 0) per_group = std::min(group_override, 64);
    per_group = std::min(per_group, info.frame_count);
    const int groups = (info.frame_count + per_group - 1) / per_group;
    const size_t lds2 = (size_t)info.nch * 9 * ROW_BYTES + (size_t)info.nch * 128 * 9 +
                        (size_t)((lay.frame_dwords + 1 + 3) / 4 * 4) * 4 + (size_t)lay.record_bytes;
    if (lds2 > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_frames_kernel, lds2));
    hipLaunchKernelGGL(hca_frames_kernel, dim3((unsigned)((int64_t)nstreams * groups)), dim3(FRAMES_THREADS), lds2, stream,
                       d_frames, frames_pitch, info, lay, reinterpret_cast<const uint8_t *>(records), per_group, groups, d_pcm,
                       stream_pitch, ch_pitch);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
