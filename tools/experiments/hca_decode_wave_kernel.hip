// EXPERIMENT, NOT PART OF THE PRODUCT (not built into libvgaudio_hip.so): round 6 built the decoder's second launch in the
// encoder's wave-per-frame form and measured it at configs[3] -- 20.0 ms against hca_frames_kernel's 18.7 ms, byte-exact
// (tests/test_gpu_hca.py green with it) -- see vgaudio_amd/csrc/hca_decode_kernels.hip: launch_decode and LABNOTES 10.6.
// Kept for the record of what was tried; it compiled against csrc/ at commit time (include paths as in csrc/).
// hca_decode_wave_kernel.hip -- the second launch of the CRI HCA decoder for one- and two-channel streams without intensity
// stereo: ONE WAVE decodes a frame (round 6; the encoder's scheme, hca_encode_wave_kernel.hip).
//
// Replaces CriHcaDecoder.DecodeFrame's stages after the header (VGAudio/Codecs/CriHca/CriHcaDecoder.cs:83-192:
// DequantizeFrame, RestoreMissingBands' high-frequency part, RunImdct, PcmFloatToShort), CriHcaPacking.ReadSpectralCoefficients
// (CriHcaPacking.cs:148-183) and Mdct.RunImdct (Utilities/Mdct.cs:94-119) -- what hca_frames_kernel (hca_decode_kernels.hip)
// does with a workgroup of two waves and five to seven block-wide barriers a frame.  hca_scan_kernel's records are the input.
//
// A wave owns a run of consecutive frames of one stream; four waves share a workgroup only for the tables.  Per frame and
// channel, wave-synchronously: the channel's 64 chunks of 16 codes are decoded by a lane each (sub-frame lane / 8, chunk
// lane % 8) straight into the transform's input rows (8 rows of LDS); the eight 128-point DCT-IVs run at once on 8 lanes
// each, twiddles from a shared LDS table; window + overlap-add + PCM16: a lane per output sample pair (j, j + 64) of every
// sub-frame, 128-byte stores.  The overlap (`_imdctPrevious`) of a channel is its last transform's output, kept in a row of
// its own from frame to frame; the first frame of a run recomputes the last sub-frame of the frame before it.
// LDS per wave: 8 rows + per channel a row of overlap, the channel's gains and resolutions, the frame's bytes and record:
// ~13 KB -- three waves per SIMD, none of them ever waiting for another.
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_decode_core.hpp"
#include "hca_kernels.hpp"

namespace vga {
namespace hca {

namespace {

// bytes between two of the wave's eight transform rows (hca_decode_core.hpp's row of 1152 bytes + what keeps the eight rows,
// which the wave accesses in lockstep, off each other's banks)
#ifndef VGA_HCA_WAVE_ROW
#define VGA_HCA_WAVE_ROW 1152
#endif
constexpr int WROW_BYTES = VGA_HCA_WAVE_ROW;

#ifndef VGA_HCA_DEC_WAVES
#define VGA_HCA_DEC_WAVES 4
#endif
constexpr int DW_WAVES = VGA_HCA_DEC_WAVES;
constexpr int DW_THREADS = 64 * DW_WAVES;
constexpr int DW_MAX_FRAMES = 16;

struct DecWaveShared {
    Symbol sym[16];
    alignas(16) Twiddle tw[127];       // [0, 63): stage tables of sizes 1..32; [63, 127): the pre-rotation (size 128, i < 64)
    double window[128];
    double dequant_scale[64];
    double step[16];
    uint8_t curve[64];
    uint8_t ath[128];
    __device__ __forceinline__ const Symbol &symbol(int r) const { return sym[r]; }
    __device__ __forceinline__ const uint8_t *res_curve() const { return curve; }
};

struct DecWaveArgs {
    const uint8_t *frames;
    int64_t frames_pitch;
    const uint8_t *records;
    int16_t *pcm;
    int64_t stream_pitch, ch_pitch;
    int frames_per_run, runs_per_stream, total_runs, wave_bytes;
    DeviceInfo info;
    DecodeLayout lay;
};

struct LdsFrameW {
    const uint32_t *w;
    int zero_at;
    __device__ __forceinline__ uint32_t get(int k) const { return w[min(k, zero_at)]; }
};
struct Res16W {
    uint4 v;
    __device__ __forceinline__ int operator[](int e) const
    {
        const uint32_t w = e < 4 ? v.x : e < 8 ? v.y : e < 12 ? v.z : v.w;
        return (int)((w >> (8 * (e & 3))) & 255u);
    }
};

__device__ __forceinline__ void lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Dct4 of the wave's eight rows at once (8 lanes per row): hca_decode_core.hpp's staged butterflies, twiddles from LDS
__device__ __forceinline__ void dct8_rows(char *row, int L, const Twiddle *tw)
{
    {
        Cx z[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 p = *reinterpret_cast<const double2 *>(row + slot_byte_offset(L + 8 * k));
            const Twiddle t = tw[63 + L + 8 * k];
            z[k].re = p.x * t.c + p.y * t.s;           // Mdct.cs:145-146
            z[k].im = p.x * t.s - p.y * t.c;
        }
        stage_fence();
#pragma unroll
        for (int k = 0; k < 4; k++) butterfly(z[k], z[k + 4], tw[31 + L + 8 * k]);
        stage_fence();
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const Twiddle t = tw[15 + L + 8 * k];
            butterfly(z[k], z[k + 2], t);
            butterfly(z[k + 4], z[k + 6], t);
        }
        stage_fence();
        const Twiddle t = tw[7 + L];
#pragma unroll
        for (int k = 0; k < 8; k += 2) butterfly(z[k], z[k + 1], t);
#pragma unroll
        for (int k = 0; k < 8; k++)
            *reinterpret_cast<double2 *>(row + slot_byte_offset(L + 8 * k)) = make_double2(z[k].re, z[k].im);
    }
    lds_order();
    Cx z[8];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const double2 p = *reinterpret_cast<const double2 *>(row + slot_byte_offset(8 * L + m));
        z[m].re = p.x;
        z[m].im = p.y;
    }
    stage_fence();
#pragma unroll
    for (int m = 0; m < 4; m++) butterfly(z[m], z[m + 4], tw[3 + m]);
    stage_fence();
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const Twiddle t = tw[1 + m];
        butterfly(z[m], z[m + 2], t);
        butterfly(z[m + 4], z[m + 6], t);
    }
    {
        const Twiddle t = tw[0];
#pragma unroll
        for (int m = 0; m < 8; m += 2) butterfly(z[m], z[m + 1], t);
    }
    lds_order();                                       // every lane of the row has read its slots
    const int rev = ((L & 1) << 2) | (L & 2) | ((L >> 2) & 1);
    const int v = rev ^ (rev >> 1) ^ (rev >> 2);
    const int out_even = 8 * v, out_odd = 8 * (v ^ 7);
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double y = (j & 1) ? z[j >> 1].im : z[j >> 1].re;
        *reinterpret_cast<double *>(row + 64 * out_block_of(j) + (parity4(j) ? out_odd : out_even)) = y * 0.125;   // Scale = sqrt(2 / 128)
    }
}

}  // namespace

template <int NCH>
__global__ __launch_bounds__(DW_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void hca_frames_wave_kernel(const DecWaveArgs args)
{
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ DecWaveShared S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const DeviceInfo &info = args.info;
    const DecodeLayout &lay = args.lay;

    if (tid < 16) {
        S.sym[tid] = make_symbol(tid, HCA_QuantizedSpectrumBits[tid & 7], HCA_QuantizedSpectrumValue[tid & 7], HCA_QuantizedSpectrumMaxBits[tid]);
        S.step[tid] = f64_bits(HCA_QuantizerStepSizeBits[tid]);
    }
    if (tid < 64) {
        S.dequant_scale[tid] = f64_bits(HCA_DequantizerScalingTableBits[tid]);
        S.curve[tid] = tid < 59 ? HCA_ScaleToResolutionCurve[tid] : 0;
    }
    if (tid < 128) {
        S.ath[tid] = info.ath_curve[tid];
        S.window[tid] = (double)__uint_as_float(HCA_MdctWindowF32Bits[tid]);
    }
    if (tid < 127) {
        const int src = tid < 63 ? tid : tid + 64;
        S.tw[tid] = Twiddle{f64_bits(MDCT_SinBits[src]), f64_bits(MDCT_CosBits[src])};
    }
    __syncthreads();

    const int run = blockIdx.x * DW_WAVES + wave;
    if (run >= args.total_runs) return;
    const int stream = run / args.runs_per_stream;
    const int f0 = (run % args.runs_per_stream) * args.frames_per_run;
    const int f1 = min(f0 + args.frames_per_run, info.frame_count);

    // this wave's LDS: rows[8] | overlap[NCH][128] f64 | gain[128] f64 | res[128] u8 | frame dwords (+1 zero) | record
    char *rows = s_dyn + (size_t)wave * args.wave_bytes;
    double *overlap = reinterpret_cast<double *>(rows + 8 * WROW_BYTES);
    double *s_gain = overlap + NCH * 128;
    uint8_t *s_res = reinterpret_cast<uint8_t *>(s_gain + 128);
    uint32_t *s_fb = reinterpret_cast<uint32_t *>(s_res + 128);
    uint8_t *s_rec = reinterpret_cast<uint8_t *>(s_fb + (lay.frame_dwords + 1 + 3) / 4 * 4);

    const uint32_t *sbase = reinterpret_cast<const uint32_t *>(args.frames + (int64_t)stream * args.frames_pitch);
    const int64_t last_word = args.frames_pitch / 4 - 1;
    const uint8_t *srec = args.records + (size_t)stream * info.frame_count * lay.record_bytes;
    int16_t *spcm = args.pcm + (int64_t)stream * args.stream_pitch;
    const int frame_bits = info.frame_size * 8;
    const int L = lane & 7, sf_of_lane = lane >> 3;
    char *my_row = rows + sf_of_lane * WROW_BYTES;
    int coded[NCH], ctype[NCH], chunk_base[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        coded[c] = info.coded_count[c];
        ctype[c] = info.channel_type[c];
        chunk_base[c] = lay.chunk_base[c];
    }

    for (int f = f0 > 0 ? f0 - 1 : f0; f < f1; f++) {
        const bool warm = f < f0;                      // the frame before the run: only its last sub-frame, no output
        lds_order();
        {   // the frame's bytes as big-endian dwords starting at its first bit, zero past its end; its record
            const int64_t a0 = (int64_t)f * info.frame_size;
            const int64_t w0 = a0 >> 2;
            const int sh8 = (int)(a0 & 3) * 8;
            for (int k = lane; k <= lay.frame_dwords; k += 64) {
                uint32_t v = 0;
                if (k < lay.frame_dwords) {
                    const uint32_t x0 = bswap32(sbase[min(w0 + k, last_word)]);
                    const uint32_t x1 = bswap32(sbase[min(w0 + k + 1, last_word)]);
                    v = sh8 ? (x0 << sh8) | (x1 >> (32 - sh8)) : x0;
                    v = mask_past_end(v, k, frame_bits);
                }
                s_fb[k] = v;
            }
            const uint4 *rec = reinterpret_cast<const uint4 *>(srec + (size_t)f * lay.record_bytes);
            for (int k = lane; k < lay.record_bytes / 16; k += 64) reinterpret_cast<uint4 *>(s_rec)[k] = rec[k];
        }
        lds_order();
        const uint32_t head = *reinterpret_cast<const uint32_t *>(s_rec + lay.header_at);
        const int noise = (int)(head & 0xFFFFu), eval = (int)((head >> 16) & 0xFFu);
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            lds_order();                               // the channel before is done with the rows, gains and resolutions
            // resolutions (CriHcaPacking.cs:85-93) and gains (CriHcaDecoder.cs:108-114) of this channel's bands
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int s = lane + 64 * h;
                const int sfac = s_rec[c * REC_CHANNEL_BYTES + s];
                const int rs = s < coded[c] ? resolution_for(S.curve, sfac, (int)S.ath[s] + noise - (s < eval ? 1 : 0)) : 0;
                s_res[s] = (uint8_t)rs;
                s_gain[s] = S.dequant_scale[sfac] * S.step[rs];
            }
            if (f == 0) {                              // the start of the stream: _imdctPrevious is cleared
                overlap[c * 128 + lane] = 0.0;
                overlap[c * 128 + 64 + lane] = 0.0;
            }
            lds_order();
            // ReadSpectralCoefficients + DequantizeFrame: lane = (sub-frame lane / 8, chunk lane % 8), 16 codes each
            if (!warm || sf_of_lane == 7) {
                const int q = L;
                const int nsym = min(max(coded[c] - 16 * q, 0), 16);
                uint32_t off = 0;
                if (nsym > 0) {
                    const int k = sf_of_lane * lay.chunks_per_subframe + chunk_base[c] + q;
                    off = lay.wide_offsets ? reinterpret_cast<const uint32_t *>(s_rec + lay.offsets_at)[k]
                                           : reinterpret_cast<const uint16_t *>(s_rec + lay.offsets_at)[k];
                }
                Res16W r16;
                r16.v = *reinterpret_cast<const uint4 *>(s_res + 16 * q);
                decode_chunk(LdsFrameW{s_fb, lay.frame_dwords}, (int)off, nsym, 16 * q, r16, s_gain + 16 * q, S, my_row);
            }
            // ReconstructHighFrequency (CriHcaDecoder.cs:116-145): lane = sub-frame lane / 8, bands in turn
            if (info.hfr_group_count > 0 && ctype[c] != CH_STEREO_SECONDARY && (!warm || sf_of_lane == 7)) {
                lds_order();
                const int total_band_count = min(info.total_band_count, 127);
                const int hfr_start = info.base_band_count + info.stereo_band_count;
                const int hfr_bands = min(info.hfr_band_count, total_band_count - info.hfr_band_count);
                for (int band = L; band < hfr_bands; band += 8) {
                    const int group = band / info.bands_per_hfr_group;
                    if (group >= info.hfr_group_count) continue;
                    const int high = hfr_start + band, low = hfr_start - band - 1;
                    const int index = (int)s_rec[c * REC_CHANNEL_BYTES + 136 + group] - (int)s_rec[c * REC_CHANNEL_BYTES + low] + 64;
                    *reinterpret_cast<double *>(my_row + spec_byte_offset(high)) =
                        f64_bits(HCA_ScaleConversionTableBits[index & 127]) * *reinterpret_cast<const double *>(my_row + spec_byte_offset(low));
                }
            }
            lds_order();
            // RunImdct's Dct4 (Mdct.cs:126-181) of the eight sub-frames at once
            if (!warm || sf_of_lane == 7) dct8_rows(my_row, L, S.tw);
            lds_order();
            // window + overlap-add (Mdct.cs:112-118), PcmFloatToShort, CopyPcmToOutput (CriHcaDecoder.cs:31-45): samples
            // j = lane (first half) and j = lane + 64 of every sub-frame
            if (!warm) {
                int16_t *dst = spcm + (int64_t)c * args.ch_pitch;
                const double w_lo = S.window[lane], w_lo_p = S.window[127 - lane];
                const double w_hi = S.window[64 + lane], w_hi_p = S.window[63 - lane];
                const int cur_lo = imdct_cur_index(lane), prev_lo = imdct_prev_index(lane);
                const int cur_hi = imdct_cur_index(lane + 64), prev_hi = imdct_prev_index(lane + 64);
#pragma unroll
                for (int sf = 0; sf < 8; sf++) {
                    const double *cur = reinterpret_cast<const double *>(rows + sf * WROW_BYTES);
                    const double *prev = sf == 0 ? overlap + c * 128 : reinterpret_cast<const double *>(rows + (sf - 1) * WROW_BYTES);
                    const int s_lo = imdct_sample(true, w_lo, w_lo_p, cur[cur_lo], prev[prev_lo]);
                    const int s_hi = imdct_sample(false, w_hi, w_hi_p, cur[cur_hi], prev[prev_hi]);
                    const int64_t tpos = (int64_t)f * SPF + sf * SPSF + lane - info.inserted_samples;
                    if (tpos >= 0 && tpos < info.sample_count) dst[tpos] = (int16_t)s_lo;
                    if (tpos + 64 >= 0 && tpos + 64 < info.sample_count) dst[tpos + 64] = (int16_t)s_hi;
                }
            }
            lds_order();
            // the channel's overlap for the next frame: its last transform's output
            {
                const double *last = reinterpret_cast<const double *>(rows + 7 * WROW_BYTES);
                const double a = last[lane], b = last[64 + lane];
                overlap[c * 128 + lane] = a;
                overlap[c * 128 + 64 + lane] = b;
            }
        }
    }
}

static size_t dec_wave_lds_bytes(const DeviceInfo &info, const DecodeLayout &lay)
{
    const size_t b = (size_t)8 * WROW_BYTES + (size_t)info.nch * 128 * 8 + 128 * 8 + 128 + (size_t)((lay.frame_dwords + 1 + 3) / 4 * 4) * 4 +
                     (size_t)lay.record_bytes;
    return (b + 15) & ~(size_t)15;
}

bool decode_wave_kernel_takes(const DeviceInfo &info)
{
    if (!(info.nch == 1 || info.nch == 2) || info.stereo_band_count > 0) return false;
    const DecodeLayout lay = make_decode_layout(info);
    return DW_WAVES * dec_wave_lds_bytes(info, lay) + sizeof(DecWaveShared) + 64 <= (DW_WAVES > 4 ? 80 : 64) * 1024;
}

int launch_frames_wave(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, const DeviceInfo &info, const DecodeLayout &lay,
                       const uint8_t *d_records, int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, hipStream_t stream,
                       int frames_per_run_override)
{
    const int64_t total = (int64_t)nstreams * info.frame_count;
    int per_run = (int)std::min<int64_t>(DW_MAX_FRAMES, std::max<int64_t>(1, total / 16384));
    if (frames_per_run_override > 0) per_run = std::min(frames_per_run_override, 64);
    per_run = std::min(per_run, info.frame_count);
    const int runs = (info.frame_count + per_run - 1) / per_run;
    const int64_t total_runs = (int64_t)nstreams * runs;
    DecWaveArgs args{};
    args.frames = d_frames;
    args.frames_pitch = frames_pitch;
    args.records = d_records;
    args.pcm = d_pcm;
    args.stream_pitch = stream_pitch;
    args.ch_pitch = ch_pitch;
    args.frames_per_run = per_run;
    args.runs_per_stream = runs;
    args.total_runs = (int)total_runs;
    args.wave_bytes = (int)dec_wave_lds_bytes(info, lay);
    args.info = info;
    args.lay = lay;
    const size_t lds = (size_t)DW_WAVES * args.wave_bytes;
    const unsigned grid = (unsigned)((total_runs + DW_WAVES - 1) / DW_WAVES);
    if (info.nch == 2) {
        if (lds > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_frames_wave_kernel<2>, lds));
        hipLaunchKernelGGL(hca_frames_wave_kernel<2>, dim3(grid), dim3(DW_THREADS), lds, stream, args);
    } else {
        if (lds > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_frames_wave_kernel<1>, lds));
        hipLaunchKernelGGL(hca_frames_wave_kernel<1>, dim3(grid), dim3(DW_THREADS), lds, stream, args);
    }
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
