// hca_encode_kernel.hip -- CRI HCA frame encoder for gfx950.
//
// Replaces CriHcaEncoder.EncodeFrame and its stages (VGAudio/Codecs/CriHca/CriHcaEncoder.cs:271-286,
// :420-858), CriHcaPacking.PackFrame (CriHcaPacking.cs:17-58, :231-295), Mdct.RunMdct
// (VGAudio/Utilities/Mdct.cs:63-92) and the streaming shell (:126-269): frame k of a stream encodes samples
// [1024k, 1024k+1024) of the encoder's input stream (PcmMap, hca_device.hpp), with the previous 128
// samples as MDCT overlap -- so, unlike the reference's stateful encoder, every frame is independent.
//
// workgroup = a run of up to 16 consecutive frames of one stream, 128 threads (round 2: one frame per workgroup of 256
// threads, every frame paying for 5.5 KB of tables and 25 block-wide barriers).  Per frame:
//   * window + fold straight from the PCM in HBM into the transform's input layout (hca_decode_core.hpp: the pre-rotation's
//     operand pairs are adjacent, rows padded against bank conflicts), every lane with its four window values in registers;
//   * the 128-point DCT-IV on 8 lanes per transform (exact staged butterflies, the decoder's), twiddles in registers;
//   * a wave per channel, a lane per two bands: scale factor, scaling in place and the band's sixteen bit costs in ONE pass
//     over its eight coefficients; the frame header's length by DPP sums inside that wave -- no block-wide barrier;
//   * both binary searches of the bit allocation on wave 0, from the cost tables (4 KB through LDS);
//   * scale factors packed per channel inside a wave; spectra codes: 16 consecutive codes per lane, offsets from one
//     block-wide scan, whole dwords ORed into the frame in LDS; CRC-16 from per-lane partial CRCs folded with x^(8k);
//   * the frame leaves as aligned dwords (consecutive frames of a run complete each other's partial cache lines in L2).
// All arithmetic is the reference's f64 in the same operation order (-ffp-contract=off); order-dependent f64 sums
// (intensity-stereo energies, HFR group averages) are done by one lane each.
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_decode_core.hpp"
#include "hca_kernels.hpp"
#include "hca_encode_core.hpp"

namespace vga {
namespace hca {

using namespace enc;

namespace {

// Timing-only builds (tools/build_variants.sh stopN:"-DVGA_HCA_ENC_STOP_AFTER=N", tools/time_hca_decode.py) leave the frame
// loop after stage N to attribute the kernel's time; the product is built without the macro (never stops).
#ifndef VGA_HCA_ENC_STOP_AFTER
#define VGA_HCA_ENC_STOP_AFTER 99
#endif
#define ENC_STOP_AFTER(n) if (VGA_HCA_ENC_STOP_AFTER == (n)) continue

constexpr int ENC_THREADS = 128;
constexpr int MAX_ENC_FRAMES_PER_GROUP = 16;

}  // namespace

// LDS gives six workgroups = three waves per SIMD; the hint keeps hipcc at the 168 VGPRs that fit (it takes 171 otherwise)
__global__ __launch_bounds__(ENC_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void hca_encode_kernel(
    const int16_t *__restrict__ pcm, int64_t stream_pitch, int64_t ch_pitch, int frames_per_group, int groups_per_stream, PcmMap map,
    DeviceInfo info, uint8_t *__restrict__ frames, int64_t frames_pitch, const uint16_t *__restrict__ crc_pow,
    int *__restrict__ status, int first_frame, int end_frame)
{
    extern __shared__ __attribute__((aligned(16))) double s_mem[];
    __shared__ EncTab T;
    __shared__ CostLut Q;
    const int nch = info.nch;
    // LDS: spectra [nch][8] rows of RS doubles | cost tables uint4 [nch][128] | small arrays | frame bits | sfac, ires
    // (launch_encode computes the same sizes; six workgroups share a CU's 160 KB at two channels: every table here counts)
    double *spectra = s_mem;
    uint4 *costs = reinterpret_cast<uint4 *>(spectra + (size_t)nch * 8 * RS);
    double *hfr_avg = reinterpret_cast<double *>(costs + nch * 128);      // [nch][8]
    double *eratio = hfr_avg + nch * 8;                                   // [nch][8]
    int *red = reinterpret_cast<int *>(eratio + nch * 8);                 // [8]: slots of the block-wide reductions (0..3), search result (4, 5), CRC (6, 7)
    int *hlb = red + 8;                                                   // [8] header length bits
    int *dbits = hlb + 8;                                                 // [8] scale-factor delta bits
    int *empty = dbits + 8;                                               // [8]
    int *cand = empty + 8;                                                // [nch][8]
    int *intensity = cand + nch * 8;                                      // [nch][8]
    int *hfrs = intensity + nch * 8;                                      // [nch][8]
    uint32_t *fbuf = reinterpret_cast<uint32_t *>(hfrs + nch * 8);        // frame bits, big-endian words [fwords]
    const int fwords = ((info.frame_size + 3) / 4 + 3) & ~1;              // even: what follows stays 8-byte aligned
    uint8_t *sfac = reinterpret_cast<uint8_t *>(fbuf + fwords);           // [nch][128] scale factors (0..63)
    uint8_t *ires = sfac + nch * 128;                                     // [nch][128] resolutions (0..15)
    __shared__ int s_coded[8], s_ctype[8];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- once per workgroup: tables, per-lane constants
    for (int i = tid; i < 64; i += ENC_THREADS) {
        T.dequant_scale[i] = f64_bits(HCA_DequantizerScalingTableBits[i]);
        T.quant_scale[i] = f64_bits(HCA_QuantizerScalingTableBits[i]);
        T.res_curve[i] = i < 59 ? HCA_ScaleToResolutionCurve[i] : 0;
    }
    if (tid < 16) {
        T.inv_step[tid] = f64_bits(HCA_QuantizerInverseStepSizeBits[tid]);
        T.max_bits[tid] = HCA_QuantizedSpectrumMaxBits[tid];
    }
    (&T.enc_pair[0][0])[tid] = (uint8_t)(((&HCA_QuantizeSpectrumValue[0][0])[tid] << 4) | (&HCA_QuantizeSpectrumBits[0][0])[tid]);
    if (tid < 8) {
        int cc = 0, ct = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (tid == k) { cc = info.coded_count[k]; ct = info.channel_type[k]; }
        s_coded[tid] = cc;
        s_ctype[tid] = ct;
    }
    if (!cost_lut_build<ENC_THREADS>(Q, s_mem, tid)) {                      // (s_mem: nothing lives there before the first frame)
        if (tid == 0 && status) atomicOr(status, 16);
        return;
    }
    const int L = tid & 7;
    const DctUniform U = make_dct_uniform(MDCT_SinBits, MDCT_CosBits);
    DctLane out_bases;                                         // only the two store bases of dct_store are used
    {
        const int rev = ((L & 1) << 2) | (L & 2) | ((L >> 2) & 1);
        const int v = rev ^ (rev >> 1) ^ (rev >> 2);
        out_bases.out_even = 8 * v;
        out_bases.out_odd = 8 * (v ^ 7);
    }
    const int wi = tid & 63;                                   // this lane's position in the fold (Mdct.cs:80-89)
    const double w_a = (double)__uint_as_float(HCA_MdctWindowF32Bits[63 - wi]);
    const double w_b = (double)__uint_as_float(HCA_MdctWindowF32Bits[64 + wi]);
    const double w_c = (double)__uint_as_float(HCA_MdctWindowF32Bits[wi]);
    const double w_d = (double)__uint_as_float(HCA_MdctWindowF32Bits[127 - wi]);

    const int stream = blockIdx.x / groups_per_stream;
    const int f0 = first_frame + (blockIdx.x % groups_per_stream) * frames_per_group;     // frames [first_frame, end_frame) of every stream
    const int f1 = min(f0 + frames_per_group, end_frame);
    const int available = info.frame_size * 8;
    const bool small = nch <= 2;

    int red_par = 0;
    // block-wide sum / exclusive scan over the two waves: DPP inside the wave, ONE barrier for the two totals (the slot
    // pair alternates, so the next call cannot overwrite a total the other wave has not read yet)
    auto block_sum = [&](int v) __attribute__((always_inline)) -> int {
        const int w = wave_sum(v);
        int *slot = red + 2 * red_par;
        red_par ^= 1;
        if (lane == 0) slot[wave] = w;
        __syncthreads();
        return slot[0] + slot[1];
    };
    auto block_exclusive_scan = [&](int v) __attribute__((always_inline)) -> int {
        const int incl = wave_inclusive_scan(v);
        int *slot = red + 2 * red_par;
        red_par ^= 1;
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        return (wave == 1 ? slot[0] : 0) + incl - v;
    };
    auto put_bits = [&](int off, unsigned value, int nbits) __attribute__((always_inline)) {
        if (nbits <= 0) return;
        const uint64_t win = (uint64_t)value << (64 - nbits - (off & 31));
        const unsigned hi = (unsigned)(win >> 32), lo = (unsigned)win;
        if (hi) atomicOr(&fbuf[off >> 5], hi);
        if (lo) atomicOr(&fbuf[(off >> 5) + 1], lo);
    };

    // A frame's input: 9 x 128 samples per channel (its 1024 and the 128 before them), staged in LDS as int16 in the region
    // the cost tables and the small arrays use later in the frame.  Up to two channels a lane fetches sample tid + 128 k of
    // both channels (k = 0..8), packed into nine registers -- and does so for the NEXT frame while this frame's bit
    // allocation and packing run, so that the HBM latency is off the frame's critical path.
    int16_t *xin = reinterpret_cast<int16_t *>(costs);                        // [nch][9 * 128]
    const int16_t *spcm = pcm + (int64_t)stream * stream_pitch;
    // 9 x 128 int16 per channel = 576 dwords; lane `tid` moves dwords tid + 128 k of the [channel][576] array (k = 0..8 for
    // two channels, 0..4 for one): nine independent dword loads, nine registers, nothing to unpack
    uint32_t pk[9];
    bool pk_valid = false;
    auto prefetch = [&](int frame) __attribute__((always_inline)) {
        const int64_t u0 = (int64_t)frame * SPF - SPSF;                        // stream index of the overlap's first sample
        // only when the whole window lies inside the caller's PCM (every frame but a stream's first and last few) and
        // starts on a dword: straight-line loads.  Other frames go through the stream map at the top of their iteration.
        const int64_t first = u0 - map.pre_end;
        pk_valid = small && u0 >= map.pre_end && u0 + SPF + SPSF <= map.main_end && ((first | ch_pitch) & 1) == 0 &&
                   (reinterpret_cast<uintptr_t>(spcm) & 3) == 0;
        if (pk_valid) {
#pragma unroll
            for (int k = 0; k < 9; k++) {
                const int d = tid + 128 * k;                                   // dword of the [nch][576] array
                if (d < nch * 576) {
                    const int c = d >= 576 ? 1 : 0;
                    pk[k] = *reinterpret_cast<const uint32_t *>(spcm + (int64_t)c * ch_pitch + first + 2 * (d - 576 * c));
                }
            }
        }
    };
    prefetch(f0);

    for (int frame = f0; frame < f1; frame++) {
        __syncthreads();                                   // the previous frame is stored; T / s_coded are written
        if (pk_valid) {
#pragma unroll
            for (int k = 0; k < 9; k++)
                if (tid + 128 * k < nch * 576) reinterpret_cast<uint32_t *>(xin)[tid + 128 * k] = pk[k];
        } else {
            const int64_t u0 = (int64_t)frame * SPF - SPSF;
            for (int i = tid; i < nch * 9 * 128; i += ENC_THREADS)
                xin[i] = fetch_pcm(map, spcm + (int64_t)(i / (9 * 128)) * ch_pitch, u0 + i % (9 * 128));
        }
        __syncthreads();

        // ---- PcmToFloat (:845-858) + the fold of RunMdct (Mdct.cs:78-89), straight into the transform's input layout
        {
            constexpr double KQ = 1.0 / 32768.0;
            for (int item = tid >> 6; item < nch * 8; item += ENC_THREADS / 64) {
                const int c = item >> 3, sf = item & 7;
                const int16_t *p = xin + c * (9 * 128) + sf * SPSF;            // previous sub-frame, then this sub-frame
                const int x_pv_lo = p[wi];
                const int x_pv_hi = p[127 - wi];
                const int x_in_lo = p[SPSF + 63 - wi];
                const int x_in_hi = p[SPSF + 64 + wi];
                const double a = w_a * -(x_in_hi * KQ);
                const double b = w_b * (x_in_lo * KQ);
                const double cc = w_c * (x_pv_lo * KQ);
                const double d = w_d * (x_pv_hi * KQ);
                char *row = reinterpret_cast<char *>(spectra + (size_t)item * RS);
                *reinterpret_cast<double *>(row + spec_byte_offset(wi)) = a - b;
                *reinterpret_cast<double *>(row + spec_byte_offset(64 + wi)) = cc - d;
            }
        }
        __syncthreads();
        for (int i = tid; i < fwords; i += ENC_THREADS) fbuf[i] = 0;           // (the staged samples may have covered it)
        ENC_STOP_AFTER(1);
        // ---- Dct4 (Mdct.cs:126-181): 8 lanes per transform, in place; the output is the row's first 128 doubles
        for (int row = tid >> 3; row < nch * 8; row += ENC_THREADS / 8) {
            char *r = reinterpret_cast<char *>(spectra + (size_t)row * RS);
            // the lane's fifteen twiddles are fetched where they are used (L1 hits): kept in registers across the frame
            // loop they would cost every other stage 60 VGPRs (the pointers are laundered so that hipcc does not hoist them)
            const uint64_t *sin_bits = MDCT_SinBits, *cos_bits = MDCT_CosBits;
            asm volatile("" : "+s"(sin_bits), "+s"(cos_bits));
            dct_first_half_streamed(r, L, sin_bits, cos_bits);
            wave_lds_sync();
            double y[16];
            dct_second_half(r, L, U, y);
            wave_lds_sync();
            dct_store(r, out_bases, y);
        }
        __syncthreads();
        if (frame + 1 < f1) prefetch(frame + 1);           // lands while this frame is allocated and packed
        ENC_STOP_AFTER(2);

        // ---- EncodeIntensityStereo (:711-764)
        if (info.stereo_band_count > 0) {
            if (tid < nch * 8) {
                const int c = tid / 8, sf = tid % 8;
                if (s_ctype[c] == CH_STEREO_PRIMARY) {
                    const double *l = spectra + ((size_t)c * 8 + sf) * RS;
                    const double *r = spectra + ((size_t)(c + 1) * 8 + sf) * RS;
                    double energy_l = 0, energy_r = 0, energy_total = 0;
                    for (int b = info.base_band_count; b < info.total_band_count; b++) {
                        energy_l += fabs(l[b]);
                        energy_r += fabs(r[b]);
                        energy_total += fabs(l[b] + r[b]);
                    }
                    energy_total *= 2;
                    const double energy_lr = energy_r + energy_l;
                    const double stored = 2 * energy_l / energy_lr;
                    double ratio = energy_lr / energy_total;
                    ratio = clampd(ratio, 0.5, 1.4142135623730951 / 2);
                    int quantized = 1;
                    if (energy_r > 0 || energy_l > 0) {
                        while (quantized < 13 && f64_bits(HCA_IntensityRatioBoundsTableBits[quantized]) >= stored) quantized++;
                    } else {
                        quantized = 0;
                        ratio = 1;
                    }
                    intensity[(c + 1) * 8 + sf] = quantized;
                    eratio[c * 8 + sf] = ratio;
                }
            }
            __syncthreads();
            const int nb = info.total_band_count - info.base_band_count;
            for (int i = tid; i < nch * 8 * nb; i += ENC_THREADS) {
                const int c = i / (8 * nb), sf = (i / nb) % 8, b = info.base_band_count + i % nb;
                if (s_ctype[c] != CH_STEREO_PRIMARY) continue;
                double *l = spectra + ((size_t)c * 8 + sf) * RS;
                double *r = spectra + ((size_t)(c + 1) * 8 + sf) * RS;
                l[b] = (l[b] + r[b]) * eratio[c * 8 + sf];
                r[b] = 0;
            }
            __syncthreads();
        }

        // ---- CalculateScaleFactors (:673-689), ScaleSpectra (:651-671) in place and the band's bit costs at all sixteen
        // resolutions (CalculateUsedBits :554-597), in one pass: wave = channel (mod 2), lane = bands `lane` and `lane + 64`.
        // Bands >= the coded count keep their unscaled values (the HFR group averages below read exactly those).
        for (int c = wave; c < nch; c += 2) {
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                const int b = lane + 64 * h;
                int sfv = 0;
                uint4 ct = make_uint4(0, 0, 0, 0);
                if (b < s_coded[c]) {
                    double *col = spectra + (size_t)c * 8 * RS + b;
                    double x[8];
                    double mx = 0;
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) {
                        x[sf] = col[(size_t)sf * RS];
                        const double coeff = fabs(x[sf]);
                        mx = coeff > mx ? coeff : mx;
                    }
                    sfv = find_scale_factor(T, mx);
                    const double qs = T.quant_scale[sfv];
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) {
                        x[sf] = sfv != 0 ? clampd(x[sf] * qs, -0.999999999999, 0.999999999999) : 0.0;
                        col[(size_t)sf * RS] = x[sf];
                    }
                    ct = band_cost_table(Q, x);
                }
                sfac[c * 128 + b] = (uint8_t)sfv;
                costs[c * 128 + b] = ct;
            }
        }
        __syncthreads();
        ENC_STOP_AFTER(3);

        // ---- CalculateHfrGroupAverages (:766-793) + CalculateHfrScale (:795-832)
        if (info.hfr_group_count > 0) {
            if (tid < nch * 8) {
                const int c = tid / 8, group = tid % 8;
                if (group < info.hfr_group_count && s_ctype[c] != CH_STEREO_SECONDARY) {
                    const int hfr_start = info.stereo_band_count + info.base_band_count;
                    double sum = 0.0;
                    int count = 0;
                    int band = hfr_start + group * info.bands_per_hfr_group;
                    for (int i = 0; i < info.bands_per_hfr_group && band < SPSF; band++, i++) {
                        for (int sf = 0; sf < 8; sf++) sum += fabs(spectra[((size_t)c * 8 + sf) * RS + band]);
                        count += 8;
                    }
                    double avg = sum / count;
                    const int lim = min(info.hfr_band_count, info.total_band_count - info.hfr_band_count);
                    sum = 0.0;
                    count = 0;
                    band = group * info.bands_per_hfr_group;
                    for (int i = 0; i < info.bands_per_hfr_group && band < lim; band++, i++) {
                        for (int sf = 0; sf < 8; sf++) sum += fabs(spectra[((size_t)c * 8 + sf) * RS + (hfr_start - band - 1)]);
                        count += 8;
                    }
                    const double average = sum / count;
                    if (average > 0.0) {
                        const double inv = 1.0 / average;
                        avg *= inv < 1.4142135623730951 ? inv : 1.4142135623730951;
                    }
                    hfrs[c * 8 + group] = find_scale_factor(T, avg);
                }
            }
            __syncthreads();
        }

        // ---- CalculateFrameHeaderLength (:599-649): a channel's five candidate delta widths' lengths are 11-bit sums packed
        // three to a register and reduced with DPP inside the channel's wave
        auto header_lengths_fast = [&]() __attribute__((always_inline)) {
            for (int c = wave; c < nch; c += 2) {
                int a = 0, b = 0, e = 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int band = lane + 64 * h;
                    const bool in = band < s_coded[c];
                    const int sf = in ? sfac[c * 128 + band] : 0;
                    const bool counted = in && band >= 1;
                    const int delta = counted ? abs(sf - (int)sfac[c * 128 + band - 1]) : 0;
                    // per-lane costs <= 22, at most 64 lanes: 16-bit fields cannot carry into each other
                    if (counted) {
                        a += (delta > 0 ? 7 : 1) | ((delta > 1 ? 8 : 2) << 16);
                        b += (delta > 3 ? 9 : 3) | ((delta > 7 ? 10 : 4) << 16);
                        e += delta > 15 ? 11 : 5;
                    }
                    e += (in && sf != 0) ? 1 << 16 : 0;                       // non-zero scale factors: "empty channel" test
                }
                a = wave_sum(a);
                b = wave_sum(b);
                e = wave_sum(e);
                if (lane == 0) {
                    const int cand_len[6] = {0, 9 + (a & 0xFFFF), 9 + (a >> 16), 9 + (b & 0xFFFF), 9 + (b >> 16), 9 + (e & 0xFFFF)};
                    int len, db;
                    if ((e >> 16) == 0) { len = 3; db = 0; }
                    else {
                        db = 6;
                        len = 3 + 6 * s_coded[c];
#pragma unroll
                        for (int k = 1; k < 6; k++)
                            if (cand_len[k] < len) { len = cand_len[k]; db = k; }
                    }
                    if (s_ctype[c] == CH_STEREO_SECONDARY) len += 32;
                    else if (info.hfr_group_count > 0) len += 6 * info.hfr_group_count;
                    hlb[c] = len;
                    dbits[c] = db;
                }
            }
            __syncthreads();
        };
        // the serial form (after bands were dropped, CalculateNoiseLevel :469-484)
        auto header_lengths = [&]() __attribute__((always_inline)) {
            if (tid < nch * 5) {
                const int c = tid / 5, db = 1 + tid % 5;
                const int max_delta = (1 << (db - 1)) - 1;
                int length = 3 + 6;
                for (int band = 1; band < s_coded[c]; band++) {
                    const int delta = sfac[c * 128 + band] - sfac[c * 128 + band - 1];
                    length += abs(delta) > max_delta ? db + 6 : db;
                }
                cand[c * 8 + db] = length;
            } else if (tid >= 64 && tid < 64 + nch) {
                const int c = tid - 64;
                int e = 1;
                for (int i = 0; i < s_coded[c]; i++)
                    if (sfac[c * 128 + i] != 0) { e = 0; break; }
                empty[c] = e;
            }
            __syncthreads();
            if (tid < nch) {
                const int c = tid;
                int len, db;
                if (empty[c]) { len = 3; db = 0; }
                else {
                    db = 6;
                    len = 3 + 6 * s_coded[c];
                    for (int k = 1; k < 6; k++)
                        if (cand[c * 8 + k] < len) { len = cand[c * 8 + k]; db = k; }
                }
                if (s_ctype[c] == CH_STEREO_SECONDARY) len += 32;
                else if (info.hfr_group_count > 0) len += 6 * info.hfr_group_count;
                hlb[c] = len;
                dbits[c] = db;
            }
            __syncthreads();
        };
        header_lengths_fast();
        ENC_STOP_AFTER(4);

        // ---- CalculateUsedBits (:554-597), block-wide (more than two channels, or after bands were dropped)
        auto used_bits = [&](int noise_level, int eval_boundary) __attribute__((always_inline)) -> int {
            int partial = 0;
            for (int i = tid; i < nch * 128; i += ENC_THREADS) {
                const int c = i >> 7, b = i & 127;
                if (b >= s_coded[c]) continue;
                const int noise = b < eval_boundary ? noise_level - 1 : noise_level;
                partial += cost_at(costs[i], resolution_of(T, sfac[i], noise));
            }
            int total = block_sum(partial) + 16 + 16 + 16;
            for (int c = 0; c < nch; c++) total += hlb[c];
            return total;
        };

        // ---- CalculateNoiseLevel (:457-485) / BinarySearchLevel (:502-523) and CalculateEvaluationBoundary (:487-500) /
        // BinarySearchBoundary (:525-552)
        int level = 0, boundary = 0;
        bool too_low = false;
        bool searched = false;
        if (small) {
            // ONE wave runs both binary searches: each of its lanes owns four bands' cost tables and every probe is ~45
            // instructions and a DPP reduction -- no LDS round trip, no barrier.  The other wave waits once.
            if (wave == 0) {
                uint64_t clo[4], chi[4];
                int off[4], bnd[4];
                bool on[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int slot = lane + 64 * k;
                    const int sf = slot < nch * 128 ? sfac[slot] : 0;
                    on[k] = slot < nch * 128 && sf != 0 && (slot & 127) < s_coded[min(slot >> 7, nch - 1)];
                    off[k] = 2 - 5 * sf / 2;
                    bnd[k] = slot & 127;
                    const uint4 cw = slot < nch * 128 ? costs[slot] : make_uint4(0, 0, 0, 0);
                    clo[k] = ((uint64_t)cw.y << 32) | cw.x;
                    chi[k] = ((uint64_t)cw.w << 32) | cw.z;
                }
                int hsum = 48;
                for (int c = 0; c < nch; c++) hsum += hlb[c];
#define probe(NL, EB) (wave_sum(probe_partial(T, clo, chi, off, bnd, on, (NL), (EB))) + hsum)
                int low = 0, high = 255, mid_value = 0;
                while (low != high) {
                    const int mid = (low + high) / 2;
                    mid_value = probe(mid, 0);
                    if (mid_value > available) low = mid + 1;
                    else high = mid;
                }
                const int lv = (low == 255 && mid_value > available) ? -1 : low;
                int bd = 0;
                if (lv > 0) {
                    // BinarySearchBoundary (:525-552) probes CalculateUsedBits(level, boundary): the bands below the
                    // boundary at level - 1, the others at level.  That is the frame's bits at `level` plus, for every band
                    // below the boundary, what the band costs more at level - 1: ONE exclusive scan over the bands gives
                    // the value of every possible probe (band b: lane b & 63, first or second half), and the search's
                    // seven or eight dependent probes are lane reads -- the same probes, the same decisions, in the
                    // reference's order (the bits need not be monotone in the boundary and nothing here assumes it).
                    int at_level = 0, d_lo = 0, d_hi = 0;                      // bands `lane` and `lane + 64`, both channels
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int r1 = T.res_curve[min(max(lv + off[k], 0), 58)], r0 = T.res_curve[min(max(lv - 1 + off[k], 0), 58)];
                        const int c1 = (int)(((r1 >= 8 ? chi[k] : clo[k]) >> (8 * (r1 & 7))) & 0xFFu);
                        const int c0 = (int)(((r0 >= 8 ? chi[k] : clo[k]) >> (8 * (r0 & 7))) & 0xFFu);
                        at_level += on[k] ? c1 : 0;
                        const int more = on[k] ? c0 - c1 : 0;
                        if (k & 1) d_hi += more;
                        else d_lo += more;
                    }
                    const int base = wave_sum(at_level) + hsum;
                    const int in_lo = wave_inclusive_scan(d_lo), in_hi = wave_inclusive_scan(d_hi);
                    const int all_lo = __builtin_amdgcn_readlane(in_lo, 63);
                    const int ex_lo = base + in_lo - d_lo, ex_hi = base + all_lo + in_hi - d_hi;      // probe(lv, band)
                    auto probe_boundary = [&](int eb) {                       // eb is wave-uniform, 0 .. 127
                        const int a = __builtin_amdgcn_readlane(ex_lo, __builtin_amdgcn_readfirstlane(eb) & 63);
                        const int b = __builtin_amdgcn_readlane(ex_hi, __builtin_amdgcn_readfirstlane(eb) & 63);
                        return eb < 64 ? a : b;
                    };
                    int lo2 = 0, hi2 = 127;
                    while (abs(hi2 - lo2) > 1) {
                        const int mid = (lo2 + hi2) / 2;
                        const int mid_value2 = probe_boundary(mid);
                        if (available < mid_value2) hi2 = mid - 1;
                        else lo2 = mid;
                    }
                    if (lo2 == hi2) bd = lo2 < 127 ? lo2 : -1;
                    else bd = probe_boundary(hi2) > available ? lo2 : hi2;
                }
#undef probe
                if (lane == 0) { red[4] = lv; red[5] = bd; }
            }
            __syncthreads();
            level = red[4];
            boundary = red[5];
            searched = level >= 0;                     // level < 0 (bands must be dropped): the block-wide form below
        }
        if (!searched) {
            auto search_level = [&]() __attribute__((always_inline)) -> int {
                int low = 0, high = 255, mid_value = 0;
                while (low != high) {
                    const int mid = (low + high) / 2;
                    mid_value = used_bits(mid, 0);
                    if (mid_value > available) low = mid + 1;
                    else high = mid;
                }
                return (low == 255 && mid_value > available) ? -1 : low;
            };
            level = small ? -1 : search_level();
            int highest_band = info.base_band_count + info.stereo_band_count - 1;
            while (level < 0) {
                highest_band -= 2;
                if (highest_band < 0) { too_low = true; break; }
                if (tid < nch) {
                    sfac[tid * 128 + highest_band + 1] = 0;
                    sfac[tid * 128 + highest_band + 2] = 0;
                }
                __syncthreads();
                header_lengths();
                level = search_level();
            }
            if (too_low) {                   // InvalidDataException("Bitrate is set too low.")
                if (tid == 0 && status) atomicOr(status, 4);
                level = 255;
            }
            boundary = 0;
            if (level != 0) {
                int low = 0, high = 127;
                while (abs(high - low) > 1) {
                    const int mid = (low + high) / 2;
                    const int mid_value = used_bits(level, mid);
                    if (available < mid_value) high = mid - 1;
                    else low = mid;
                }
                if (low == high) boundary = low < 127 ? low : -1;
                else {
                    const int hi_value = used_bits(level, high);
                    boundary = hi_value > available ? low : high;
                }
            }
        }
        ENC_STOP_AFTER(5);
        if (boundary < 0) {                   // NotImplementedException in the reference
            if (tid == 0 && status) atomicOr(status, 8);
            boundary = 0;
        }

        // ---- CalculateFrameResolutions (:441-455)
        for (int i = tid; i < nch * 128; i += ENC_THREADS) {
            const int c = i >> 7, b = i & 127;
            ires[i] = b < s_coded[c] ? resolution_of(T, sfac[i], b < boundary ? level - 1 : level) : 0;
        }
        // ---- PackFrame (CriHcaPacking.cs:17-58); a frame the reference refuses ("Bitrate is set too low.")
        // is left zero -- its codes would not fit the frame
        if (tid == 0 && !too_low) atomicOr(&fbuf[0], 0xFFFF0000u | ((unsigned)level << 7) | (unsigned)boundary);
        int header_bits = 32;
        for (int k = 0; k < nch; k++) header_bits += hlb[k];
        // WriteScaleFactors (:262-295): wave = channel (mod 2), lane = bands 2 lane and 2 lane + 1; the variable-length
        // codes get their bit offsets from a scan inside the wave
        for (int c = wave; c < nch && !too_low; c += 2) {
            const int db = dbits[c];
            const uint8_t *sc = sfac + c * 128;
            unsigned code[2] = {0, 0};
            int nb2[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int band = 2 * lane + h;
                if (band == 0) {                               // the 3-bit delta width, then the first scale factor
                    code[h] = (unsigned)db;
                    nb2[h] = 3;
                    if (db != 0) { code[h] = (code[h] << 6) | (unsigned)sc[0]; nb2[h] = 9; }
                } else if (band < s_coded[c] && db != 0) {
                    if (db == 6) { code[h] = (unsigned)sc[band]; nb2[h] = 6; }
                    else {
                        const int max_delta = (1 << (db - 1)) - 1;
                        const int delta = sc[band] - sc[band - 1];
                        if (abs(delta) > max_delta) { code[h] = ((((1u << db) - 1)) << 6) | (unsigned)sc[band]; nb2[h] = db + 6; }
                        else { code[h] = (unsigned)(max_delta + delta); nb2[h] = db; }
                    }
                }
            }
            const int mine = nb2[0] + nb2[1];
            const int incl = wave_inclusive_scan(mine);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            int off = 32;
            for (int k = 0; k < c; k++) off += hlb[k];
            put_bits(off + incl - mine, code[0], nb2[0]);
            put_bits(off + incl - mine + nb2[0], code[1], nb2[1]);
            if (lane == 0) {                                   // intensity / HFR scales follow the scale factors
                off += total;
                if (s_ctype[c] == CH_STEREO_SECONDARY) {
                    for (int i = 0; i < 8; i++) { put_bits(off, (unsigned)intensity[c * 8 + i], 4); off += 4; }
                } else if (info.hfr_group_count > 0) {
                    for (int i = 0; i < info.hfr_group_count; i++) { put_bits(off, (unsigned)hfrs[c * 8 + i], 6); off += 6; }
                }
            }
        }
        __syncthreads();                               // ires is complete
        ENC_STOP_AFTER(6);
        // WriteSpectra (:238-260) in (sub-frame, channel, band) order: slot = (sf * nch + c) * 128 + band;
        // QuantizeSpectra (:420-439) on the fly.  A lane owns nch * 8 consecutive slots; 8 nch divides 128 for 1, 2, 4
        // and 8 channels, otherwise a lane's slots run on into the next channel's row.
        {
            const int per_thread = nch * 8;
            const int slot0 = tid * per_thread;
            // (sub-frame, channel, band) of the lane's first slot, then stepped (a division per code would cost more than
            // the code)
            int w_sf = slot0 / (nch * 128), w_c = (slot0 >> 7) - w_sf * nch, w_band = slot0 & 127;
            auto rewind = [&]() __attribute__((always_inline)) {
                w_sf = slot0 / (nch * 128);
                w_c = (slot0 >> 7) - w_sf * nch;
                w_band = slot0 & 127;
            };
            auto code_of = [&](unsigned &code, int &nbits) __attribute__((always_inline)) {
                const int sf = w_sf, c = w_c, band = w_band;
                w_band++;
                if (w_band == 128) {
                    w_band = 0;
                    w_c++;
                    if (w_c == nch) { w_c = 0; w_sf++; }
                }
                const int res = ires[c * 128 + band];
                code = 0;
                nbits = 0;
                if (res == 0) return;
                const double inv = T.inv_step[res];
                const double up = inv + 1;
                const int down = trunc_i(inv + 0.5);
                const int q = trunc_i(spectra[((size_t)c * 8 + sf) * RS + band] * inv + up) - down;
                if (res < 8) {
                    const unsigned pair = T.enc_pair[res][q + 8];   // value << 4 | bits
                    nbits = (int)(pair & 15u);
                    code = pair >> 4;
                } else {
                    nbits = T.max_bits[res] - 1;
                    code = (unsigned)abs(q);
                    if (q != 0) { code = (code << 1) | (q > 0 ? 0u : 1u); nbits++; }
                }
            };
            if (per_thread <= 16) {
                // up to two channels: a lane's 8 or 16 slots are consecutive bands of one (sub-frame, channel) row; their
                // resolutions arrive as one or two 64-bit LDS reads and the codes stay in registers (13 + 4 bits each).
                // The quantiser's constants need no table: QuantizerInverseStepSize[r] = ResolutionMaxValue[r] + 0.5
                // (CriHcaTables.cs:57), so inv = max + 0.5, shiftUp = inv + 1, shiftDown = (int)(inv + 0.5) = max + 1,
                // and QuantizedSpectrumMaxBits[r] = r - 3 from resolution 8 on.
                const double *xs = spectra + ((size_t)w_c * 8 + w_sf) * RS + w_band;
                const uint64_t *rs = reinterpret_cast<const uint64_t *>(ires + w_c * 128 + w_band);
                const uint64_t res_lo = rs[0], res_hi = per_thread > 8 ? rs[1] : 0;
                unsigned packed[16];
                int local = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    packed[k] = 0;
                    if (k < per_thread) {
                        const int res = (int)(((k < 8 ? res_lo : res_hi) >> (8 * (k & 7))) & 0xFFu);
                        const int maxv = res < 8 ? res : (1 << (res - 4)) - 1;
                        const double inv = (double)maxv + 0.5;
                        const int q = trunc_i(xs[k] * inv + (inv + 1)) - (maxv + 1);
                        const unsigned small_pair = T.enc_pair[min(res, 7)][(q + 8) & 15];   // value << 4 | bits
                        const unsigned mag = (unsigned)abs(q);
                        const unsigned large_pair = q != 0 ? ((((mag << 1) | (q > 0 ? 0u : 1u)) << 4) | (unsigned)(res - 3))
                                                           : (unsigned)(res - 4);
                        const unsigned pair = res == 0 ? 0u : (res < 8 ? small_pair : large_pair);
                        packed[k] = pair;
                        local += (int)(pair & 15u);
                    }
                }
                const int off = header_bits + block_exclusive_scan(local);
                Emitter e{fbuf, 0, off >> 5, off & 31};
                if (!too_low) {
#pragma unroll
                    for (int k = 0; k < 16; k++)
                        if (packed[k] & 15u) e.put(packed[k] >> 4, (int)(packed[k] & 15u));
                    e.finish();
                }
            } else {
                int local = 0;
                for (int k = 0; k < per_thread; k++) {
                    unsigned code;
                    int nbits;
                    code_of(code, nbits);
                    local += nbits;
                }
                rewind();
                const int off = header_bits + block_exclusive_scan(local);
                Emitter e{fbuf, 0, off >> 5, off & 31};
                for (int k = 0; k < per_thread && !too_low; k++) {
                    unsigned code;
                    int nbits;
                    code_of(code, nbits);
                    if (nbits > 0) e.put(code, nbits);
                }
                if (!too_low) e.finish();
            }
        }
        __syncthreads();
        ENC_STOP_AFTER(7);

        // ---- WriteChecksum (:231-236): CRC-16 (poly 0x8005, init 0) over the first frame_size-2 bytes
        {
            const int nbytes = info.frame_size - 2;
            const int chunk = (nbytes + ENC_THREADS - 1) / ENC_THREADS;
            const int begin = tid * chunk, end = min(begin + chunk, nbytes);
            unsigned crc = 0;
            for (int i = begin; i < end; i++) {
                const unsigned byte = (fbuf[i >> 2] >> (24 - 8 * (i & 3))) & 0xFFu;
                // eight shift-and-xor steps of x^16 + x^15 + x^2 + 1 at once: with t = the byte entering the register,
                // t * x^16 mod P = t << 1 ^ t << 2 ^ (parity(t) ? 0x8003 : 0)   (checked against the bitwise form
                // for every (crc, byte) pair: tests/test_oracle_hca.py)
                const unsigned t = ((crc >> 8) ^ byte) & 0xFFu;
                crc = ((crc << 8) & 0xFFFFu) ^ ((__popc(t) & 1) ? 0x8003u : 0u) ^ (t << 1) ^ (t << 2);
            }
            unsigned part = (begin < end) ? gf_mul(crc, crc_pow[nbytes - end]) : 0u;
            part = (unsigned)wave_xor((int)part);
            if (lane == 0) red[6 + wave] = (int)part;
            __syncthreads();
            if (tid == 0) {
                const unsigned total = (unsigned)(red[6] ^ red[7]) & 0xFFFFu;
                const int pos = nbytes;        // big-endian 16-bit value at the last two bytes
                fbuf[pos >> 2] |= (total >> 8) << (24 - 8 * (pos & 3));
                fbuf[(pos + 1) >> 2] |= (total & 0xFF) << (24 - 8 * ((pos + 1) & 3));
            }
            __syncthreads();
        }
        ENC_STOP_AFTER(8);

        // ---- store the frame: whole aligned dwords (the frame starts at any byte: its k-th dword is a funnel shift of two
        // big-endian words of fbuf), the few bytes before the first and after the last aligned dword one by one
        {
            uint8_t *dst = frames + (int64_t)stream * frames_pitch + (int64_t)frame * info.frame_size;
            const int lead = (int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);       // bytes before the first aligned dword
            const int ndw = (info.frame_size - lead) / 4;
            auto byte_at = [&](int b) { return (fbuf[b >> 2] >> (24 - 8 * (b & 3))) & 0xFFu; };
            uint32_t *dw = reinterpret_cast<uint32_t *>(dst + lead);
            for (int k = tid; k < ndw; k += ENC_THREADS) {
                const int b = lead + 4 * k;                                                   // frame byte of the dword's first byte
                const uint32_t hi = fbuf[b >> 2], lo = fbuf[(b >> 2) + 1];
                const int sh = 8 * (b & 3);
                const uint32_t be = sh ? (hi << sh) | (lo >> (32 - sh)) : hi;                 // frame bytes b .. b+3, first byte on top
                dw[k] = bswap32(be);
            }
            const int tail0 = lead + 4 * ndw;
            if (tid < lead) dst[tid] = (uint8_t)byte_at(tid);
            if (tid >= 64 && tid - 64 < info.frame_size - tail0) dst[tail0 + tid - 64] = (uint8_t)byte_at(tail0 + tid - 64);
        }
    }
}

int launch_encode(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, const PcmMap &map,
                  const DeviceInfo &info, uint8_t *d_frames, int64_t frames_pitch, const uint16_t *d_crc_pow,
                  int *d_status, hipStream_t stream, int first_frame, int frame_limit)
{
    // frames [first_frame, first_frame + frame_limit) of every stream (frame_limit < 0: to the stream's end) -- a frame is a
    // function of the stream's PCM alone (PcmMap), so the streaming shell (vga_hca_stream_encode) asks for the frames the
    // reference's encoder would have output by now
    const int end_frame = frame_limit < 0 ? info.frame_count : std::min(info.frame_count, first_frame + frame_limit);
    if (nstreams <= 0 || first_frame < 0 || end_frame <= first_frame) return VGA_OK;
    const int frame_span = end_frame - first_frame;
    const int nch = info.nch;
    // test hook (vga_testing_hca_frames_per_group_this_thread): n > 0 = frames per run / group, 1000 + n = this file's
    // workgroup-per-run kernel whatever the channel count (n = 0: its default run length)
    const int hook = hca_frames_per_group_override();
    if (hook < 1000 && encode_wave_kernel_takes(info))
        return launch_encode_wave(d_pcm, stream_pitch, ch_pitch, nstreams, map, info, d_frames, frames_pitch, d_crc_pow, d_status, stream,
                                  first_frame, end_frame, hook);
    const int group_override = hook >= 1000 ? hook - 1000 : hook;
    const size_t doubles = (size_t)nch * 8 * RS + (size_t)nch * 16;
    const size_t ints = 8 + 8 + 8 + 8 + 3 * (size_t)nch * 8;
    const size_t lds = doubles * 8 + (size_t)nch * 128 * 16 + ints * 4 + (size_t)((((info.frame_size + 3) / 4 + 3) & ~1) * 4) + (size_t)nch * 256;
    if (lds > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_encode_kernel, lds));
    // frames per workgroup: long runs amortise the per-workgroup set-up (tables, twiddles), short ones keep small inputs
    // spread over the chip
    const int64_t total = (int64_t)nstreams * frame_span;
    int per_group = (int)std::min<int64_t>(MAX_ENC_FRAMES_PER_GROUP, std::max<int64_t>(1, total / 8192));
    if (group_override > 0) per_group = std::min(group_override, 64);
    per_group = std::min(per_group, frame_span);
    const int groups = (frame_span + per_group - 1) / per_group;
    hipLaunchKernelGGL(hca_encode_kernel, dim3((unsigned)((int64_t)nstreams * groups)), dim3(ENC_THREADS), lds, stream,
                       d_pcm, stream_pitch, ch_pitch, per_group, groups, map, info, d_frames, frames_pitch, d_crc_pow, d_status,
                       first_frame, end_frame);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
