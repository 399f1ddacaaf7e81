// hca_encode_kernel.hip -- CRI HCA frame encoder for gfx950.
//
// Replaces CriHcaEncoder.EncodeFrame and its stages (VGAudio/Codecs/CriHca/CriHcaEncoder.cs:271-286,
// :420-858), CriHcaPacking.PackFrame (CriHcaPacking.cs:17-58, :231-295), Mdct.RunMdct
// (VGAudio/Utilities/Mdct.cs:63-92) and the non-looping streaming shell (:126-269): frame k of a
// stream encodes samples [1024k, 1024k+1024) of "input followed by zeros", with the previous 128
// samples as MDCT overlap -- so, unlike the reference's stateful encoder, every frame is independent.
//
// workgroup = (stream, frame), 256 threads.  All arithmetic is the reference's f64 in the same
// operation order (-ffp-contract=off); order-dependent f64 sums (intensity-stereo energies, HFR group
// averages) are done by one lane each; the bit-allocation searches (CalculateUsedBits, ~16 evaluations)
// are block reductions; packing is a block-wide prefix sum of code lengths + LDS atomic ORs; the
// CRC-16 is computed in parallel from per-chunk CRCs multiplied by x^(8*bytes_after) mod 0x18005.
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_kernels.hpp"

namespace vga {
namespace hca {

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// CriHcaEncoder.cs:691-709
__device__ __forceinline__ int find_scale_factor(const EncTables &T, double value)
{
    unsigned low = 0, high = 63;
    while (low < high) {
        const unsigned mid = (low + high) / 2;
        if (T.dequant_scale[mid] <= value) low = mid + 1;
        else high = mid;
    }
    return (int)low;
}

// (int)double for values known to be small
__device__ __forceinline__ int trunc_i(double d) { return (int)d; }

// multiply in GF(2)[x] / (x^16 + x^15 + x^2 + 1)
__device__ __forceinline__ unsigned gf_mul(unsigned a, unsigned b)
{
    unsigned r = 0;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        r = ((r << 1) ^ ((r & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        if ((a >> i) & 1u) r ^= b;
    }
    return r;
}

// CalculateUsedBits (:554-597) for one band: the bits its eight scaled coefficients cost at resolution `res`
// x: the band's eight coefficients, one per sub-frame, XS doubles apart
template <int XS>
__device__ __forceinline__ int band_cost(const EncTables &T, const double *x, int res)
{
    int cost = 0;
    if (res >= 8) {
        const int bits = T.max_bits[res] - 1;
        const double d = T.dead_zone[res];
#pragma unroll
        for (int sf = 0; sf < 8; sf++) cost += bits + (fabs(x[sf * XS]) >= d ? 1 : 0);
    } else {
        const double inv = T.inv_step[res];
        const double up = inv + 1;
        const int down = trunc_i(inv + 0.5 - 8);
#pragma unroll
        for (int sf = 0; sf < 8; sf++) {
            const int q = trunc_i(x[sf * XS] * inv + up) - down;
            cost += T.enc_bits[res][q];
        }
    }
    return cost;
}

// 16 costs (each <= 8 * 12 bits) packed into four dwords, plus a "known" bit per resolution
struct UsedBitsMemo {
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, known = 0;
};

// All sixteen costs of one band at once (the resolution is a compile-time constant in every unrolled step, so each
// takes one side of band_cost only and the eight coefficients are read from LDS once).  The binary searches touch
// a new resolution in some lane on nearly every step, which made the lazy variant evaluate band_cost -- both sides,
// the lanes' resolutions differ -- about fifteen times per frame.
__device__ __forceinline__ void build_cost_table(const EncTables &T, const double *xs, bool valid, UsedBitsMemo &m)
{
    double x[8];
#pragma unroll
    for (int sf = 0; sf < 8; sf++) x[sf] = xs[sf * 128];               // scaled spectra: [channel][sub-frame][band]
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 16; r++) w[r >> 2] |= (uint32_t)band_cost<1>(T, x, r) << (8 * (r & 3));
    m.w0 = valid ? w[0] : 0u; m.w1 = valid ? w[1] : 0u; m.w2 = valid ? w[2] : 0u; m.w3 = valid ? w[3] : 0u;
    m.known = 0xFFFFu;
}

// this thread's share of the frame's spectrum bits.  Up to two channels a thread owns one band and memoises
// its costs; with more it owns several bands and recomputes.
__device__ __forceinline__ int used_bits_partial(const EncTables &T, int tid, int nch, const int *s_coded, const uint8_t *sfac,
                                                 const double *scaled, int noise_level, int eval_boundary, UsedBitsMemo &m)
{
    int partial = 0;
    if (nch * 128 <= 256) {
        const int i = min(tid, nch * 128 - 1);
        const int c = i / 128, b = i % 128;
        const bool valid = tid < nch * 128 && b < s_coded[c];
        const int noise = b < eval_boundary ? noise_level - 1 : noise_level;
        const int res = calculate_resolution(T, sfac[i], noise);
        const int word = res >> 2, shift = 8 * (res & 3);
        const uint32_t wsel = word == 0 ? m.w0 : word == 1 ? m.w1 : word == 2 ? m.w2 : m.w3;
        partial = valid ? (int)((wsel >> shift) & 0xFFu) : 0;
    } else {
        for (int i = tid; i < nch * 128; i += 256) {
            const int c = i / 128, b = i % 128;
            if (b >= s_coded[c]) continue;
            const int noise = b < eval_boundary ? noise_level - 1 : noise_level;
            partial += band_cost<128>(T, scaled + (size_t)c * 1024 + b, calculate_resolution(T, sfac[i], noise));
        }
    }
    return partial;
}

// One probe of CalculateUsedBits for the four bands a lane of the searching wave owns: resolution from the noise level
// (CriHcaPacking.CalculateResolution), cost from the band's table.
// The sixteen byte costs of a band sit in two 64-bit halves (resolutions 0-7, 8-15): one select and one 64-bit shift
// (a four-way select over a uint4 makes hipcc spill the table to scratch and index it).
__device__ __forceinline__ int probe_partial(const EncTables &T, const uint64_t (&clo)[4], const uint64_t (&chi)[4],
                                             const int (&off)[4], const int (&bnd)[4], const bool (&on)[4], int noise_level,
                                             int eval_boundary)
{
    int partial = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int noise = bnd[k] < eval_boundary ? noise_level - 1 : noise_level;
        const int res = T.res_curve[min(max(noise + off[k], 0), 58)];
        const uint64_t half = res >= 8 ? chi[k] : clo[k];
        partial += on[k] ? (int)((half >> (8 * (res & 7))) & 0xFFu) : 0;
    }
    return partial;
}

// LDS limits the kernel to 3-4 workgroups (12-16 waves) per CU; without the occupancy hint hipcc aims for 10
// waves per SIMD, caps itself at 48 VGPRs and spills pointers to scratch
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void hca_encode_kernel(
    const int16_t *__restrict__ pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, PcmMap map,
    DeviceInfo info, uint8_t *__restrict__ frames, int64_t frames_pitch, const uint16_t *__restrict__ crc_pow,
    int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) double s_mem[];
    __shared__ EncTables T;
    const int nch = info.nch;
    double *spectra = s_mem;                               // [nch][8][128]
    // LDS decides the occupancy (5 workgroups per CU need <= 32 KB each): ScaleSpectra works IN PLACE -- the scaled
    // spectra keep the [channel][sub-frame][band] layout of the MDCT output (bands of one sub-frame are contiguous,
    // which is also what the cost tables and WriteSpectra read conflict-free); only the MDCT staging needs a second
    // region (dctin, which doubles as the transform's scratch, and xin; later the searching wave's cost tables)
    const size_t region_b = 11 * 128;
    double *scaled = spectra;                              // [nch][8][128], bands < coded count (the rest stays unscaled)
    double *dctin = spectra + (size_t)nch * 1024;          // [8][128]
    double *tmp = dctin;                                   // the transform permutes in place (hca_device.hpp)
    int16_t *xin = reinterpret_cast<int16_t *>(dctin + 8 * 128);   // [9][128] raw samples (2.3 KB of the 3 x 128 doubles)
    double *hfr_avg = dctin + region_b;                    // [nch][8]
    double *eratio = hfr_avg + nch * 8;                    // [nch][8]
    int *red = reinterpret_cast<int *>(eratio + nch * 8);  // [32] two alternating slot sets for the block reductions
    int *hlb = red + 32;                                   // [nch] header length bits
    int *dbits = hlb + 8;                                  // [nch] scale-factor delta bits
    int *cand = dbits + 8;                                 // [nch][8]
    int *empty = cand + 64;                                // [nch]
    int *intensity = empty + 8;                            // [nch][8]
    int *hfrs = intensity + 64;                            // [nch][8]
    uint32_t *fbuf = reinterpret_cast<uint32_t *>(hfrs + 64);    // frame bits, big-endian words [fwords]
    const int fwords = (info.frame_size + 3) / 4 + 2;
    uint8_t *sfac = reinterpret_cast<uint8_t *>(fbuf + fwords);  // [nch][128] scale factors (0..63)
    uint8_t *ires = sfac + nch * 128;                      // [nch][128] resolutions (0..15)

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int stream = blockIdx.x / info.frame_count;
    const int frame = blockIdx.x % info.frame_count;

    // block-wide sum / exclusive scan: wave-level shuffles, then ONE barrier for the four wave totals (the slot
    // set alternates, so the next call cannot overwrite totals a slower wave has not read yet)
    int red_par = 0;
    auto block_sum = [&](int v) __attribute__((always_inline)) -> int {
        // wave sum with DPP (quad swaps, half-row and row mirrors: every lane then holds its 16-lane row's sum) and
        // four v_readlane -- the binary searches wait for this fifteen times in a row, and six dependent
        // ds_bpermute (what __shfl_xor compiles to) cost several hundred cycles each time
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);     // row_half_mirror
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);     // row_mirror
        const int w = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
                      __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
        int *slot = red + 8 * red_par;
        red_par ^= 1;
        if (lane == 0) slot[wave] = w;
        __syncthreads();
        return slot[0] + slot[1] + slot[2] + slot[3];
    };
    auto wave_inclusive_scan = [&](int v) __attribute__((always_inline)) -> int {
        // Hillis-Steele inside each 16-lane row with row_shr, then the row totals are handed on with row_bcast
        v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);     // row_shr:1 (no source lane: + 0)
        v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);     // row_shr:2
        v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);     // row_shr:4
        v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);     // row_shr:8
        v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
        v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
        return v;
    };
    // exclusive scan over groups of `waves_per_group` consecutive waves (4: the whole block; 2: one channel's
    // 128 bands); total = the group's sum
    auto group_exclusive_scan = [&](int v, int waves_per_group, int &total) __attribute__((always_inline)) -> int {
        const int incl = wave_inclusive_scan(v);
        int *slot = red + 8 * red_par;
        red_par ^= 1;
        if (lane == 63) slot[wave] = incl;
        __syncthreads();
        const int first = wave / waves_per_group * waves_per_group;
        int base = 0;
        total = 0;
        for (int w = first; w < first + waves_per_group; w++) {
            const int t = slot[w];
            if (w < wave) base += t;
            total += t;
        }
        return base + incl - v;
    };
    auto put_bits = [&](int off, unsigned value, int nbits) __attribute__((always_inline)) {
        if (nbits <= 0) return;
        const uint64_t win = (uint64_t)value << (64 - nbits - (off & 31));
        const unsigned hi = (unsigned)(win >> 32), lo = (unsigned)win;
        if (hi) atomicOr(&fbuf[off >> 5], hi);
        if (lo) atomicOr(&fbuf[(off >> 5) + 1], lo);
    };

    // per-channel layout in LDS: indexing the by-value kernel argument with a per-lane channel number makes
    // hipcc copy the whole struct to scratch
    __shared__ int s_coded[8], s_ctype[8];
    if (tid < 8) {
        int cc = 0, ct = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (tid == k) { cc = info.coded_count[k]; ct = info.channel_type[k]; }
        s_coded[tid] = cc;
        s_ctype[tid] = ct;
    }
    for (int i = tid; i < fwords; i += 256) fbuf[i] = 0;
    load_tables(T, tid, 256);

    // ---- PcmToFloat (:845-858) + RunMdct (:834-843 -> Mdct.cs:63-92), channel by channel
    const int grp = tid >> 5, t = tid & 31;
    for (int c = 0; c < nch; c++) {
        const int16_t *src = pcm + (int64_t)stream * stream_pitch + (int64_t)c * ch_pitch;
        for (int i = tid; i < 9 * 128; i += 256) {
            const int64_t pos = (int64_t)frame * SPF - SPSF + i;
            xin[i] = fetch_pcm(map, src, pos);
        }
        __syncthreads();
        {
            // PcmToFloat: pcm * (1.0 / 32768.0), applied on the fly
            const int16_t *in = xin + (grp + 1) * 128, *prev = xin + grp * 128;
            double *din = dctin + grp * 128;
            constexpr double K = 1.0 / 32768.0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int i = t + 32 * k;
                const double a = T.window[63 - i] * -(in[64 + i] * K);
                const double b = T.window[64 + i] * (in[63 - i] * K);
                const double cc = T.window[i] * (prev[i] * K);
                const double d = T.window[127 - i] * (prev[127 - i] * K);
                din[i] = a - b;
                din[64 + i] = cc - d;
            }
        }
        __syncthreads();
        dct4_128(T, dctin + grp * 128, tmp + grp * 128, spectra + ((size_t)c * 8 + grp) * 128, t, wave_sync);
        __syncthreads();
    }

    // ---- EncodeIntensityStereo (:711-764)
    if (info.stereo_band_count > 0) {
        if (tid < nch * 8) {
            const int c = tid / 8, sf = tid % 8;
            if (s_ctype[c] == CH_STEREO_PRIMARY) {
                const double *l = spectra + ((size_t)c * 8 + sf) * 128;
                const double *r = spectra + ((size_t)(c + 1) * 8 + sf) * 128;
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
        for (int i = tid; i < nch * 8 * nb; i += 256) {
            const int c = i / (8 * nb), sf = (i / nb) % 8, b = info.base_band_count + i % nb;
            if (s_ctype[c] != CH_STEREO_PRIMARY) continue;
            double *l = spectra + ((size_t)c * 8 + sf) * 128;
            double *r = spectra + ((size_t)(c + 1) * 8 + sf) * 128;
            l[b] = (l[b] + r[b]) * eratio[c * 8 + sf];
            r[b] = 0;
        }
        __syncthreads();
    }

    // ---- CalculateScaleFactors (:673-689)
    for (int i = tid; i < nch * 128; i += 256) {
        const int c = i / 128, b = i % 128;
        int sfv = 0;
        if (b < s_coded[c]) {
            double mx = 0;
            for (int sf = 0; sf < 8; sf++) {
                const double coeff = fabs(spectra[((size_t)c * 8 + sf) * 128 + b]);
                mx = coeff > mx ? coeff : mx;
            }
            sfv = find_scale_factor(T, mx);
        }
        sfac[i] = sfv;
    }
    __syncthreads();
    // ---- ScaleSpectra (:651-671), in place: bands >= the coded count keep their unscaled values (the HFR group
    // averages below read exactly those), nothing reads a scaled value there
    for (int i = tid; i < nch * 1024; i += 256) {
        const int c = i / 1024, b = i % 128;
        const int sfv = sfac[c * 128 + b];
        if (b < s_coded[c])
            spectra[i] = sfv != 0 ? clampd(spectra[i] * T.quant_scale[sfv], -0.999999999999, 0.999999999999) : 0.0;
    }
    __syncthreads();

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
                    for (int sf = 0; sf < 8; sf++) sum += fabs(spectra[((size_t)c * 8 + sf) * 128 + band]);
                    count += 8;
                }
                double avg = sum / count;
                const int lim = min(info.hfr_band_count, info.total_band_count - info.hfr_band_count);
                sum = 0.0;
                count = 0;
                band = group * info.bands_per_hfr_group;
                for (int i = 0; i < info.bands_per_hfr_group && band < lim; band++, i++) {
                    for (int sf = 0; sf < 8; sf++) sum += fabs(scaled[((size_t)c * 8 + sf) * 128 + (hfr_start - band - 1)]);
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

    // ---- CalculateFrameHeaderLength (:599-649)
    // Up to two channels (every BASELINE shape): lane = (channel, band).  The five candidate delta widths' lengths are
    // 11-bit sums packed three to a register and reduced with DPP -- the serial form below walks 128 bands on ten lanes
    // while the other 246 wait at the barrier (~13 k cycles of a 66 k-cycle frame).
    const bool small = nch * 128 <= 256;
    auto wave_sum = [&](int v) __attribute__((always_inline)) -> int {
        v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);      // quad_perm [1,0,3,2]
        v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);      // quad_perm [2,3,0,1]
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);     // row_half_mirror
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);     // row_mirror
        return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
               __builtin_amdgcn_readlane(v, 48);
    };
    auto header_lengths_fast = [&]() __attribute__((always_inline)) {
        const int c = tid >> 7, band = tid & 127;
        const bool in = c < nch && band < s_coded[min(c, nch - 1)];
        const int sf = in ? sfac[c * 128 + band] : 0;
        const int delta = (in && band >= 1) ? abs(sf - (int)sfac[c * 128 + band - 1]) : 0;
        const bool counted = in && band >= 1;
        // per-lane costs <= 11, at most 127 lanes: 16-bit fields cannot carry into each other
        int a = 0, b = 0, e = 0;
        if (counted) {
            a = (delta > 0 ? 7 : 1) | ((delta > 1 ? 8 : 2) << 16);
            b = (delta > 3 ? 9 : 3) | ((delta > 7 ? 10 : 4) << 16);
            e = delta > 15 ? 11 : 5;
        }
        e |= (in && sf != 0) ? 1 << 16 : 0;                               // non-zero scale factors: "empty channel" test
        a = wave_sum(a);
        b = wave_sum(b);
        e = wave_sum(e);
        int *slot = red + 16;                                             // [3][4]: red[16..27]; red[28..29] hold the search result
        if (lane == 0) { slot[wave] = a; slot[4 + wave] = b; slot[8 + wave] = e; }
        __syncthreads();
        if (tid < nch) {
            const int cc = tid;
            const int sa = slot[2 * cc] + slot[2 * cc + 1], sb = slot[4 + 2 * cc] + slot[4 + 2 * cc + 1],
                      se = slot[8 + 2 * cc] + slot[8 + 2 * cc + 1];
            const int cand_len[6] = {0, 9 + (sa & 0xFFFF), 9 + (sa >> 16), 9 + (sb & 0xFFFF), 9 + (sb >> 16), 9 + (se & 0xFFFF)};
            int len, db;
            if ((se >> 16) == 0) { len = 3; db = 0; }
            else {
                db = 6;
                len = 3 + 6 * s_coded[cc];
#pragma unroll
                for (int k = 1; k < 6; k++)
                    if (cand_len[k] < len) { len = cand_len[k]; db = k; }
            }
            if (s_ctype[cc] == CH_STEREO_SECONDARY) len += 32;
            else if (info.hfr_group_count > 0) len += 6 * info.hfr_group_count;
            hlb[cc] = len;
            dbits[cc] = db;
        }
        __syncthreads();
    };
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
    if (small) header_lengths_fast();
    else header_lengths();

    // ---- CalculateUsedBits (:554-597)
    // The bit cost of a band's eight coefficients depends only on its resolution (the scaled spectra are fixed): every
    // (band, resolution) cost is computed once (build_cost_table: 16 costs of <= 96 bits in four dwords per band).
    UsedBitsMemo memo;
    if (small) {
        const int i = min(tid, nch * 128 - 1);
        build_cost_table(T, scaled + (size_t)(i >> 7) * 1024 + (i & 127), tid < nch * 128 && (i & 127) < s_coded[i >> 7], memo);
    }
    auto used_bits = [&](int noise_level, int eval_boundary) __attribute__((always_inline)) -> int {
        const int partial = used_bits_partial(T, tid, nch, s_coded, sfac, scaled, noise_level, eval_boundary, memo);
        int total = block_sum(partial) + 16 + 16 + 16;
        for (int c = 0; c < nch; c++) total += hlb[c];
        return total;
    };

    // ---- CalculateNoiseLevel (:457-485) / BinarySearchLevel (:502-523) and CalculateEvaluationBoundary (:487-500) /
    // BinarySearchBoundary (:525-552).
    const int available = info.frame_size * 8;
    int level = 0, boundary = 0;
    bool too_low = false;
    bool searched = false;
    if (small) {
        // ONE wave runs both binary searches: the cost tables of all bands travel to it through LDS (4 KB, in the dead
        // MDCT staging region), each of its lanes then owns four bands and every probe is ~45 instructions and a DPP
        // reduction -- no LDS round trip, no barrier -- where the block-wide form issued the same ~45 instructions on
        // four waves and met at a barrier sixteen times per frame.  The other three waves wait once.
        uint4 *costs = reinterpret_cast<uint4 *>(dctin);
        costs[tid] = make_uint4(memo.w0, memo.w1, memo.w2, memo.w3);
        __syncthreads();
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
                const uint4 cw = costs[slot];
                clo[k] = ((uint64_t)cw.y << 32) | cw.x;
                chi[k] = ((uint64_t)cw.w << 32) | cw.z;
            }
            int hsum = 48;
            for (int c = 0; c < nch; c++) hsum += hlb[c];
            // (a free function, not a lambda: hipcc keeps by-reference captures of register arrays in scratch)
#define probe(NL, EB) (wave_sum(probe_partial(T, clo, chi, off, bnd, on, (NL), (EB))) + hsum)
            int low = 0, high = 255, mid_value = 0;
            while (low != high) {
                const int mid = (low + high) / 2;
                mid_value = probe(mid, 0);
                if (mid_value > available) low = mid + 1;
                else high = mid;
            }
            int lv = (low == 255 && mid_value > available) ? -1 : low;
            int bd = 0;
            if (lv > 0) {
                int lo2 = 0, hi2 = 127;
                while (abs(hi2 - lo2) > 1) {
                    const int mid = (lo2 + hi2) / 2;
                    const int mid_value2 = probe(lv, mid);
                    if (available < mid_value2) hi2 = mid - 1;
                    else lo2 = mid;
                }
                if (lo2 == hi2) bd = lo2 < 127 ? lo2 : -1;
                else bd = probe(lv, hi2) > available ? lo2 : hi2;
            }
#undef probe
            if (lane == 0) { red[28] = lv; red[29] = bd; }
        }
        __syncthreads();
        level = red[28];
        boundary = red[29];
        searched = level >= 0;                         // level < 0 (bands must be dropped): the block-wide form below
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
        if (too_low) {                       // InvalidDataException("Bitrate is set too low.")
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
    if (boundary < 0) {                       // NotImplementedException in the reference
        if (tid == 0 && status) atomicOr(status, 8);
        boundary = 0;
    }

    // ---- CalculateFrameResolutions (:441-455)
    for (int i = tid; i < nch * 128; i += 256) {
        const int c = i / 128, b = i % 128;
        ires[i] = b < s_coded[c] ? calculate_resolution(T, sfac[i], b < boundary ? level - 1 : level) : 0;
    }
    __syncthreads();

    // ---- PackFrame (CriHcaPacking.cs:17-58); a frame the reference refuses ("Bitrate is set too low.")
    // is left zero -- its codes would not fit the frame
    if (tid == 0 && !too_low) fbuf[0] = 0xFFFF0000u | ((unsigned)level << 7) | (unsigned)boundary;
    // WriteScaleFactors (:262-295): lane = (channel, band), two channels per pass; the variable-length codes
    // get their bit offsets from a 128-lane exclusive scan
    for (int c0 = 0; c0 < nch; c0 += 2) {
        const int c = c0 + (tid >> 7), band = tid & 127;
        unsigned code = 0;
        int nbits = 0;
        if (c < nch && !too_low) {
            const int db = dbits[c];
            const uint8_t *sc = sfac + c * 128;
            if (band == 0) {                           // the 3-bit delta width, then the first scale factor
                code = (unsigned)db;
                nbits = 3;
                if (db != 0) { code = (code << 6) | (unsigned)sc[0]; nbits = 9; }
            } else if (band < s_coded[c] && db != 0) {
                if (db == 6) { code = (unsigned)sc[band]; nbits = 6; }
                else {
                    const int max_delta = (1 << (db - 1)) - 1;
                    const int delta = sc[band] - sc[band - 1];
                    if (abs(delta) > max_delta) { code = ((((1u << db) - 1)) << 6) | (unsigned)sc[band]; nbits = db + 6; }
                    else { code = (unsigned)(max_delta + delta); nbits = db; }
                }
            }
        }
        int total;
        const int rel = group_exclusive_scan(nbits, 2, total);
        if (c < nch && !too_low) {
            int off = 32;
            for (int k = 0; k < c; k++) off += hlb[k];
            put_bits(off + rel, code, nbits);
            if (band == 0) {                           // intensity / HFR scales follow the scale factors
                off += total;
                if (s_ctype[c] == CH_STEREO_SECONDARY) {
                    for (int i = 0; i < 8; i++) { put_bits(off, (unsigned)intensity[c * 8 + i], 4); off += 4; }
                } else if (info.hfr_group_count > 0) {
                    for (int i = 0; i < info.hfr_group_count; i++) { put_bits(off, (unsigned)hfrs[c * 8 + i], 6); off += 6; }
                }
            }
        }
    }
    // WriteSpectra (:238-260) in (sub-frame, channel, band) order: slot = (sf*nch + c)*128 + band;
    // QuantizeSpectra (:420-439) on the fly
    {
        const int per_thread = nch * 4;               // nch*8*128 / 256, divides 128
        // per_thread divides 128: a thread's slots share the sub-frame and the channel, the band runs on
        const int slot0 = tid * per_thread;
        const int sf = slot0 / (nch * 128), c = (slot0 / 128) % nch, band0 = slot0 % 128;
        const double *xs = scaled + ((size_t)c * 8 + sf) * 128 + band0;
        const uint8_t *rs = ires + c * 128 + band0;
        auto code_of = [&](int k, unsigned &code, int &nbits) __attribute__((always_inline)) {
            const int res = rs[k];
            code = 0;
            nbits = 0;
            if (res == 0) return;
            const double inv = T.inv_step[res];
            const double up = inv + 1;
            const int down = trunc_i(inv + 0.5);
            // a thread's slots run on across a channel boundary when 4 * nch does not divide 128 (3, 5, 6, 7 channels):
            // the sub-frame stays, the band wraps into the next channel
            const int bk = band0 + k;
            const int q = trunc_i(xs[(size_t)(bk >> 7) * 1024 + (bk & 127) - band0] * inv + up) - down;
            if (res < 8) {
                nbits = T.enc_bits[res][q + 8];
                code = T.enc_value[res][q + 8];
            } else {
                nbits = T.max_bits[res] - 1;
                code = (unsigned)abs(q);
                if (q != 0) { code = (code << 1) | (q > 0 ? 0u : 1u); nbits++; }
            }
        };
        // a thread's codes are consecutive in the stream: they are gathered in a 64-bit window and leave as whole
        // dwords (one LDS atomic per dword instead of up to two per code)
        struct Emitter {
            unsigned *buf;
            uint64_t acc;
            int word, p;
            __device__ __forceinline__ void put(unsigned value, int nbits)
            {
                acc |= (uint64_t)value << (64 - p - nbits);            // p < 32, nbits <= 13
                p += nbits;
                if (p >= 32) {
                    const unsigned hi = (unsigned)(acc >> 32);
                    if (hi) atomicOr(&buf[word], hi);
                    acc <<= 32;
                    word++;
                    p -= 32;
                }
            }
            __device__ __forceinline__ void finish()
            {
                const unsigned hi = (unsigned)(acc >> 32);
                if (hi) atomicOr(&buf[word], hi);
            }
        };
        int header_bits = 32;
        for (int k = 0; k < nch; k++) header_bits += hlb[k];
        if (per_thread <= 8) {                         // up to two channels: the codes stay in registers
            unsigned codes[8];
            int nb[8];
            int local = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                codes[k] = 0;
                nb[k] = 0;
                if (k < per_thread) code_of(k, codes[k], nb[k]);
                local += nb[k];
            }
            int all_bits;
            const int off = header_bits + group_exclusive_scan(local, 4, all_bits);
            Emitter e{fbuf, 0, off >> 5, off & 31};
            if (!too_low) {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (nb[k] > 0) e.put(codes[k], nb[k]);
                e.finish();
            }
        } else {
            int local = 0;
            for (int k = 0; k < per_thread; k++) {
                unsigned code;
                int nbits;
                code_of(k, code, nbits);
                local += nbits;
            }
            int all_bits;
            const int off = header_bits + group_exclusive_scan(local, 4, all_bits);
            Emitter e{fbuf, 0, off >> 5, off & 31};
            for (int k = 0; k < per_thread && !too_low; k++) {
                unsigned code;
                int nbits;
                code_of(k, code, nbits);
                if (nbits > 0) e.put(code, nbits);
            }
            if (!too_low) e.finish();
        }
    }
    __syncthreads();

    // ---- WriteChecksum (:231-236): CRC-16 (poly 0x8005, init 0) over the first frame_size-2 bytes
    {
        const int nbytes = info.frame_size - 2;
        const int chunk = (nbytes + 255) / 256;
        const int begin = tid * chunk, end = min(begin + chunk, nbytes);
        unsigned crc = 0;
        for (int i = begin; i < end; i++) {
            const unsigned byte = (fbuf[i >> 2] >> (24 - 8 * (i & 3))) & 0xFFu;
            crc ^= byte << 8;
#pragma unroll
            for (int j = 0; j < 8; j++) crc = ((crc << 1) ^ ((crc & 0x8000u) ? 0x8005u : 0u)) & 0xFFFFu;
        }
        unsigned part = (begin < end) ? gf_mul(crc, crc_pow[nbytes - end]) : 0u;
        {
            int v = (int)part;
            v ^= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);  // as block_sum, with xor
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
            v ^= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
            part = (unsigned)(__builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 16) ^
                              __builtin_amdgcn_readlane(v, 32) ^ __builtin_amdgcn_readlane(v, 48));
        }
        if (lane == 0) red[wave] = (int)part;
        __syncthreads();
        if (tid == 0) {
            const unsigned total = (unsigned)(red[0] ^ red[1] ^ red[2] ^ red[3]) & 0xFFFFu;
            const int pos = nbytes;        // big-endian 16-bit value at the last two bytes
            fbuf[pos >> 2] |= (total >> 8) << (24 - 8 * (pos & 3));
            fbuf[(pos + 1) >> 2] |= (total & 0xFF) << (24 - 8 * ((pos + 1) & 3));
        }
        __syncthreads();
    }

    // ---- store the frame (frame offsets are even: 2-byte stores)
    {
        uint16_t *dst = reinterpret_cast<uint16_t *>(frames + (int64_t)stream * frames_pitch + (int64_t)frame * info.frame_size);
        for (int i = tid; i < info.frame_size / 2; i += 256) {
            const int b = 2 * i;
            const unsigned b0 = (fbuf[b >> 2] >> (24 - 8 * (b & 3))) & 0xFFu;
            const unsigned b1 = (fbuf[(b + 1) >> 2] >> (24 - 8 * ((b + 1) & 3))) & 0xFFu;
            dst[i] = (uint16_t)(b0 | (b1 << 8));
        }
        if ((info.frame_size & 1) && tid == 0) {
            const int b = info.frame_size - 1;
            frames[(int64_t)stream * frames_pitch + (int64_t)frame * info.frame_size + b] =
                (uint8_t)((fbuf[b >> 2] >> (24 - 8 * (b & 3))) & 0xFFu);
        }
    }
}

int launch_encode(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, const PcmMap &map,
                  const DeviceInfo &info, uint8_t *d_frames, int64_t frames_pitch, const uint16_t *d_crc_pow,
                  int *d_status, hipStream_t stream)
{
    if (nstreams <= 0 || info.frame_count <= 0) return VGA_OK;
    const int nch = info.nch;
    const size_t region_b = 11 * 128;
    const size_t doubles = (size_t)nch * 1024 + region_b + (size_t)nch * 16;
    const size_t ints = 32 + 8 + 8 + 64 + 8 + 64 + 64;
    const size_t lds = doubles * 8 + ints * 4 + ((size_t)(info.frame_size + 3) / 4 + 2) * 4 + (size_t)nch * 256;
    if (lds > 64 * 1024)
        VGA_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(hca_encode_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(hca_encode_kernel, dim3((unsigned)((int64_t)nstreams * info.frame_count)), dim3(256), lds, stream,
                       d_pcm, stream_pitch, ch_pitch, nstreams, map, info, d_frames, frames_pitch, d_crc_pow,
                       d_status);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
