// hca_encode_wave_kernel.hip -- CRI HCA frame encoder for one- and two-channel streams: ONE WAVE encodes a frame.
//
// Replaces CriHcaEncoder.EncodeFrame and its stages (VGAudio/Codecs/CriHca/CriHcaEncoder.cs:271-286, :420-858),
// CriHcaPacking.PackFrame (CriHcaPacking.cs:17-58, :231-295) and Mdct.RunMdct (VGAudio/Utilities/Mdct.cs:63-92) for
// streams of up to two channels -- what hca_encode_kernel.hip did with a workgroup of two waves, ~20 block-wide barriers a
// frame, stages that ran on one of the two waves, and every intermediate (spectra, scale factors, cost tables,
// resolutions) parked in LDS.  HCA frames have no recurrence: frame k is a function of samples [1024 k - 128, 1024 k + 1024)
// of the encoder's input stream (PcmMap, hca_device.hpp).
//
// A wave owns a run of consecutive frames of one stream and never waits for another wave (four waves share a workgroup
// only for the tables: one barrier per workgroup, none per frame).  Per frame, wave-synchronously:
//   * samples: lane l loads sample 64 k + l (k = 0..17) of both channels as one packed register each, already one frame
//     ahead; the mirrored operand of the fold (sample 64 k + 63 - l) comes across the crossbar (ds_bpermute);
//   * per channel: window + fold into the transform's input rows (8 sub-frames = 8 rows of LDS), the 128-point DCT-IV on
//     8 lanes per transform -- all eight sub-frames of a channel at once --, twiddles from a shared LDS table;
//   * then the frame lives in REGISTERS in band order: lane l holds bands l and l + 64 of both channels, all eight
//     sub-frames (32 doubles), its four scale factors, four 128-bit cost tables and resolutions -- scale factors,
//     scaling, bit costs, the header's length, both binary searches of the bit allocation and the quantiser need no LDS
//     array and no barrier; the sums over bands are DPP reductions;
//   * the codes (value + length in 16 bits) are turned through LDS into stream order -- 32 consecutive codes per lane --,
//     one scan gives the lanes' bit offsets and each lane ORs whole dwords into the frame; CRC-16 from per-lane partial
//     CRCs; the frame leaves as aligned dwords.
// LDS per wave: 9 KB (the eight rows; the turned codes and the frame's bits reuse them) -- three to four waves per SIMD.
// Intensity stereo and HFR (the order-dependent f64 sums of :711-832) take a detour through the rows: one lane per sum.
// All arithmetic is the reference's f64 in the same operation order (-ffp-contract=off).
#include "common.hpp"
#include "hca_device.hpp"
#include "hca_decode_core.hpp"
#include "hca_kernels.hpp"
#include "hca_encode_core.hpp"

namespace vga {
namespace hca {

using namespace enc;

namespace {

// bytes between two of the wave's eight transform rows (hca_decode_core.hpp's row of 1152 bytes + what keeps the eight rows,
// which the wave accesses in lockstep, off each other's banks)
#ifndef VGA_HCA_WAVE_ROW
#define VGA_HCA_WAVE_ROW 1152
#endif
constexpr int WROW_BYTES = VGA_HCA_WAVE_ROW;

#ifndef VGA_HCA_ENC_STOP_AFTER
#define VGA_HCA_ENC_STOP_AFTER 99
#endif
// (the frame lives in registers: a timing-only build has to keep what it computed alive -- `sink` is folded into a store that
// never happens)
#define WAVE_STOP_AFTER(n, sink)                                          \
    if (VGA_HCA_ENC_STOP_AFTER == (n)) {                                  \
        if ((int)(sink) == 0x7A5C3E19 && cold_args()->status) atomicOr(cold_args()->status, 1 << 20); \
        continue;                                                         \
    }

// (tuning builds: tools/build_variants.sh)
// Eight waves a workgroup, two workgroups a CU (2 x (8 x 9 KB + 5.5 KB of tables) = 155 KB of the 160) = four waves per
// SIMD at 128 VGPRs: 21.0 ms at configs[3] against 22.2 for four-wave workgroups at three waves per SIMD
// (profiles/r06_c_hca_encode_variants.log).
#ifndef VGA_HCA_WG_WAVES
#define VGA_HCA_WG_WAVES 8
#endif
#ifndef VGA_HCA_WAVE_EU
#define VGA_HCA_WAVE_EU 4
#endif
constexpr int WG_WAVES = VGA_HCA_WG_WAVES;
constexpr int WG_THREADS = 64 * WG_WAVES;
constexpr int MAX_WAVE_FRAMES = 16;
constexpr int ROWS_BYTES = 8 * WROW_BYTES;              // one channel's eight transforms

// FindScaleFactor (:691-709) returns how many of the table's first 63 entries are <= value.  The table is geometric (ratio
// 2^(53/128) = 1.33), so the value's exponent and top three mantissa bits (a bucket 2^(1/8) = 1.09 wide) leave at most one
// entry undecided: a byte per bucket says how many entries lie at or below the bucket's lower edge and ONE exact compare with
// the next entry decides the rest -- two dependent LDS reads instead of the binary search's six (checked when the workgroup
// builds the table; entry 63 of the workgroup's copy of the table is +inf: the count stops at 63).
constexpr int SF_BUCKETS = 216;

// what the waves of a workgroup share (static LDS)
struct alignas(16) WaveShared {
    EncTab T;
    CostLut Q;
    alignas(16) Twiddle tw[127];                   // [0, 63): the stage tables of sizes 1..32 (size 2^b starts at 2^b - 1); [63, 127): the pre-rotation
    double window[128];                // MdctWindow / 32768 (exact: PcmToFloat's scaling folded in, see fold below)
    uint8_t sf_rank[SF_BUCKETS];       // FindScaleFactor by look-up (find_scale_factor_lut)
    int sf_key_base;
    int bad;
};

// Everything the kernel is told, as ONE by-value argument: the kernarg segment then IS this struct, and the stages that run
// once in a thousand frames (the stream's first and last frames through the stream map, intensity stereo, HFR, error flags)
// read their fields from it where they need them (cold_args) instead of keeping ~25 SGPRs of them alive through the frame
// loop, where they were spilled to VGPR lanes and read back lane by lane.
struct WaveArgs {
    const int16_t *pcm;
    int64_t stream_pitch, ch_pitch;
    uint8_t *frames;
    int64_t frames_pitch;
    const uint16_t *crc_pow;
    int *status;
    int frames_per_run, runs_per_stream, total_runs, first_frame, end_frame, wave_bytes;
    PcmMap map;
    DeviceInfo info;
};
typedef const __attribute__((address_space(4))) WaveArgs *ColdArgs;
__device__ __forceinline__ ColdArgs cold_args()
{
    ColdArgs p = (ColdArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                         // opaque: the loads stay where they are written
    return p;
}

template <int NCH>
struct Layout {
    static constexpr int PT = 16 * NCH;                // codes per lane in stream order: 8 sub-frames x NCH x 128 / 64
    static constexpr int STRIDE = 2 * PT + 16;         // bytes between two lanes' runs of codes: 16-byte reads, four lanes per bank group
    static constexpr int CODES_BYTES = 64 * STRIDE;
};

__device__ __forceinline__ int wave_shr1(int v)       // lane l receives lane l - 1's value, lane 0 receives 0
{
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);          // wave_shr:1
}

// The lane number, opaque to hipcc: what a stage of the frame derives from it (LDS addresses, band numbers, masks) is then
// computed where the stage starts and dies with it, instead of being hoisted out of the frame loop into registers that
// stay occupied through every other stage (45 VGPRs of such values at the first count).
__device__ __forceinline__ int fresh(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// v < lo ? lo : (v > hi ? hi : v) for a v that is not NaN, lo < 0 < hi (two instructions instead of two compares and four
// selects; the sign of a zero cannot come into it)
__device__ __forceinline__ double minmax_clamp(double v, double lo, double hi)
{
    return __builtin_fmin(__builtin_fmax(v, lo), hi);
}

// pre-rotation + stages 0..2 (hca_decode_core.hpp: dct_first_half) with the lane's twiddles from the shared table
__device__ __forceinline__ void dct_first_half_lds(char *row, int L, const Twiddle *tw)
{
    Cx z[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double2 p = *reinterpret_cast<const double2 *>(row + slot_byte_offset(L + 8 * k));
        const Twiddle t = tw[63 + L + 8 * k];
        z[k].re = p.x * t.c + p.y * t.s;               // Mdct.cs:145-146
        z[k].im = p.x * t.s - p.y * t.c;
        if ((k & 3) == 3) stage_fence();               // four operand pairs in flight, not eight
    }                                     // (keeps hipcc from fetching every stage's twiddles up front: 60 VGPRs)
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
    {
        const Twiddle t = tw[7 + L];
#pragma unroll
        for (int k = 0; k < 8; k += 2) butterfly(z[k], z[k + 1], t);
    }
#pragma unroll
    for (int k = 0; k < 8; k++)
        *reinterpret_cast<double2 *>(row + slot_byte_offset(L + 8 * k)) = make_double2(z[k].re, z[k].im);
}

// stages 3..5 (dct_second_half) + the output permutation and scale (dct_store)
__device__ __forceinline__ void dct_second_half_lds(char *row, int L, const Twiddle *tw, int out_even, int out_odd)
{
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every lane of the row has read its slots
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double y = (j & 1) ? z[j >> 1].im : z[j >> 1].re;
        const int base = parity4(j) ? out_odd : out_even;
        *reinterpret_cast<double *>(row + 64 * out_block_of(j) + base) = y * 0.125;     // Scale = sqrt(2 / 128)
    }
}

// value >= 0, not NaN
__device__ __forceinline__ int find_scale_factor_lut(const WaveShared &S, double value)
{
    const int key = (int)((uint32_t)__double2hiint(value) >> 17) - S.sf_key_base;
    const int rb = S.sf_rank[min(max(key, 0), SF_BUCKETS - 1)];
    return rb + (value >= S.T.dequant_scale[rb] ? 1 : 0);
}

}  // namespace

template <int NCH>
__global__ __launch_bounds__(WG_THREADS) __attribute__((amdgpu_waves_per_eu(VGA_HCA_WAVE_EU))) void hca_encode_wave_kernel(const WaveArgs args)
{
    using Lay = Layout<NCH>;
    constexpr int PT = Lay::PT;
    const int16_t *__restrict__ pcm = args.pcm;
    uint8_t *__restrict__ frames = args.frames;
    const int64_t stream_pitch = args.stream_pitch, ch_pitch = args.ch_pitch, frames_pitch = args.frames_pitch;
    const int frame_size = args.info.frame_size, hfr_group_count = args.info.hfr_group_count;
    const int map_pre_end = args.map.pre_end, map_main_end = args.map.main_end;
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ WaveShared S;
    const int tid = threadIdx.x;
    const int lane0 = tid & 63, wave = tid >> 6;

    // ---- once per workgroup: tables
    for (int i = tid; i < 64; i += WG_THREADS) {
        S.T.dequant_scale[i] = f64_bits(HCA_DequantizerScalingTableBits[i]);
        S.T.quant_scale[i] = f64_bits(HCA_QuantizerScalingTableBits[i]);
        S.T.res_curve[i] = i < 59 ? HCA_ScaleToResolutionCurve[i] : 0;
    }
    if (tid < 16) {
        S.T.inv_step[tid] = f64_bits(HCA_QuantizerInverseStepSizeBits[tid]);
        S.T.max_bits[tid] = HCA_QuantizedSpectrumMaxBits[tid];
    }
    if (tid < 128) {
        (&S.T.enc_pair[0][0])[tid] = (uint8_t)(((&HCA_QuantizeSpectrumValue[0][0])[tid] << 4) | (&HCA_QuantizeSpectrumBits[0][0])[tid]);
        // PcmToFloat (:845-858) multiplies every sample by 2^-15 and the fold (Mdct.cs:80-89) multiplies the result by a
        // window value: w * (x * 2^-15) and (w * 2^-15) * x round to the same double (a power of two scales exactly;
        // nothing here is near the subnormal range: |w| >= 6.9e-4, |x| >= 1 or x = 0)
        S.window[tid] = (double)__uint_as_float(HCA_MdctWindowF32Bits[tid]) * (1.0 / 32768.0);
    }
    if (tid < 127) {
        const int src = tid < 63 ? tid : tid + 64;                  // 63.. -> entries 127.. of the reference's table (size 128, i < 64)
        S.tw[tid] = Twiddle{f64_bits(MDCT_SinBits[src]), f64_bits(MDCT_CosBits[src])};
    }
    if (tid == 0) S.bad = 0;
    __syncthreads();
    S.sf_key_base = (int)(HCA_DequantizerScalingTableBits[0] >> 49) - 1;       // (every thread writes the same value)
    if (tid == 63) S.T.dequant_scale[63] = __longlong_as_double(0x7FF0000000000000ll);
    if (tid < SF_BUCKETS) {
        const int kb = (int)(HCA_DequantizerScalingTableBits[0] >> 49) - 1;
        // bucket 0: everything below the table's first entry; bucket b: keys kb + b (the last one: that and everything above)
        const double lower = tid == 0 ? 0.0 : __hiloint2double((kb + tid) << 17, 0);
        const double upper = __hiloint2double((kb + tid + 1) << 17, 0);
        int at_or_below = 0, inside = 0;
        for (int j = 0; j < 63; j++) {
            const double t = f64_bits(HCA_DequantizerScalingTableBits[j]);
            at_or_below += t <= lower ? 1 : 0;
            inside += (t > lower && t < upper) ? 1 : 0;
        }
        if (inside > 1 || (tid == 0 && (at_or_below | inside) != 0) || (tid == SF_BUCKETS - 1 && at_or_below != 63)) S.bad = 1;
        S.sf_rank[tid] = (uint8_t)at_or_below;
    }
    __syncthreads();
    if (S.bad || !cost_lut_build<WG_THREADS>(S.Q, reinterpret_cast<double *>(s_dyn), tid)) {       // (s_dyn: nothing lives there before the first frame)
        if (tid == 0 && cold_args()->status) atomicOr(cold_args()->status, 16);
        return;
    }
    __syncthreads();
    const EncTab &T = S.T;
    const CostLut &Q = S.Q;

    const int run = blockIdx.x * WG_WAVES + wave;
    if (run >= args.total_runs) return;
    const int stream = run / args.runs_per_stream;
    const int f0 = args.first_frame + (run % args.runs_per_stream) * args.frames_per_run;   // frames [first_frame, end_frame) of every stream
    const int f1 = min(f0 + args.frames_per_run, args.end_frame);

    char *rows = s_dyn + (size_t)wave * args.wave_bytes;                 // eight transform rows; later the turned codes + the frame's bits
    uint32_t *fbuf = reinterpret_cast<uint32_t *>(rows + Lay::CODES_BYTES);
    const int fwords = ((frame_size + 3) / 4 + 3) & ~1;
    const int available = frame_size * 8;

    int coded[NCH], ctype[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        coded[c] = args.info.coded_count[c];
        ctype[c] = args.info.channel_type[c];
    }
    const bool has_intensity = NCH == 2 && args.info.stereo_band_count > 0 && ctype[0] == CH_STEREO_PRIMARY;
    const int nbytes = frame_size - 2;
    const int crc_chunk = (nbytes + 63) / 64;
    // x^(8 k) mod P for the k bytes that follow this lane's chunk of the frame
    const unsigned crc_shift = lane0 * crc_chunk < nbytes ? args.crc_pow[nbytes - min(lane0 * crc_chunk + crc_chunk, nbytes)] : 0u;

    const int16_t *spcm = pcm + (int64_t)stream * stream_pitch;
    for (int frame = f0; frame < f1; frame++) {
        // sample 64 k + lane of the frame's window (its 1024 samples and the 128 before them), k = 0..17: channel 0 in the low
        // half, channel 1 in the high half -- straight from the caller's PCM when the whole window lies inside it (every
        // frame but a stream's first and last few).  (Loading them a frame ahead, under the previous frame's packing, was
        // measured: 18 registers held through the packing cost more than the exposed latency, 26.1 against 22.1 ms; touching
        // the next frame's lines with one load into an unused corner of LDS: 20.8 against 20.6 ms, nothing.)
        uint32_t pk[18];
        const int ln_in = fresh(lane0);
        const int64_t u0 = (int64_t)frame * SPF - SPSF;
        if (u0 >= map_pre_end && u0 + SPF + SPSF <= map_main_end) {
            const int16_t *p = spcm + (u0 - map_pre_end) + ln_in;
#pragma unroll
            for (int k = 0; k < 18; k++) {
                uint32_t v = (uint16_t)p[64 * k];
                if (NCH == 2) v |= (uint32_t)(uint16_t)p[ch_pitch + 64 * k] << 16;
                pk[k] = v;
            }
        } else {
            // through the stream map, staged in the rows (int16 [channel][18][64])
            int16_t *stage = reinterpret_cast<int16_t *>(rows);
            PcmMap map;
            {
                ColdArgs k = cold_args();
                map.zero_pre = k->map.zero_pre;
                map.pre_end = k->map.pre_end;
                map.main_end = k->map.main_end;
                map.post_end = k->map.post_end;
                map.loop_start = k->map.loop_start;
                map.last_chunk = k->map.last_chunk;
                map.raw_len = k->map.raw_len;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
            for (int i = 0; i < 18 * NCH; i++)
                stage[i * 64 + ln_in] = fetch_pcm(map, spcm + (int64_t)(i / 18) * ch_pitch, u0 + 64 * (i % 18) + ln_in);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < 18; k++) {
                uint32_t v = (uint16_t)stage[k * 64 + ln_in];
                if (NCH == 2) v |= (uint32_t)(uint16_t)stage[(18 + k) * 64 + ln_in] << 16;
                pk[k] = v;
            }
        }

        // ---- PcmToFloat (:845-858), RunMdct (Mdct.cs:63-92): per channel the fold of its eight sub-frames into the eight
        // rows, the eight transforms at once, and the spectra into registers in band order: x[c][h][sf] = band ln_in + 64 h.
        // (What only this stage needs -- window values, LDS addresses -- is derived from the ln_in number here, per channel,
        // instead of living in registers through the rest of the frame: `ln` is laundered so that hipcc does not hoist it.)
        double x[NCH][2][8];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            int ln = ln_in;
            asm volatile("" : "+v"(ln));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the previous user of the rows is done reading
            {
                const double w_a = S.window[63 - ln], w_b = S.window[64 + ln], w_c = S.window[ln], w_d = S.window[127 - ln];
                const int fold_lo = spec_byte_offset(ln), fold_hi = spec_byte_offset(64 + ln);
                const int mirror_addr = (63 - ln) * 4;
                auto half_of = [&](uint32_t v) { return c == 0 ? (int)(int16_t)(v & 0xFFFFu) : (int)v >> 16; };
#pragma unroll
                for (int sf = 0; sf < 8; sf++) {
                    // the fold's mirrored operands (sample 64 k + 63 - ln_in) come across the crossbar
                    const uint32_t m_odd = (uint32_t)__builtin_amdgcn_ds_bpermute(mirror_addr, (int)pk[2 * sf + 1]);
                    const uint32_t m_even = (uint32_t)__builtin_amdgcn_ds_bpermute(mirror_addr, (int)pk[2 * sf + 2]);
                    const int x_pv_lo = half_of(pk[2 * sf]);                 // p[wi]
                    const int x_pv_hi = half_of(m_odd);                      // p[127 - wi]
                    const int x_in_lo = half_of(m_even);                     // p[128 + 63 - wi]
                    const int x_in_hi = half_of(pk[2 * sf + 3]);             // p[128 + 64 + wi]
                    const double fa = w_a * -(double)x_in_hi;
                    const double fb = w_b * (double)x_in_lo;
                    const double fc = w_c * (double)x_pv_lo;
                    const double fd = w_d * (double)x_pv_hi;
                    *reinterpret_cast<double *>(rows + sf * WROW_BYTES + fold_lo) = fa - fb;
                    *reinterpret_cast<double *>(rows + sf * WROW_BYTES + fold_hi) = fc - fd;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            {
                const int L = ln & 7;
                char *my_row = rows + (ln >> 3) * WROW_BYTES;
                dct_first_half_lds(my_row, L, S.tw);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int rev = ((L & 1) << 2) | (L & 2) | ((L >> 2) & 1);
                const int v = rev ^ (rev >> 1) ^ (rev >> 2);
                dct_second_half_lds(my_row, L, S.tw, 8 * v, 8 * (v ^ 7));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int sf = 0; sf < 8; sf++)
                    x[c][h][sf] = *reinterpret_cast<const double *>(rows + sf * WROW_BYTES + 8 * (ln + 64 * h));
        }
        WAVE_STOP_AFTER(2, [&] { double t = 0; for (int c = 0; c < NCH; c++) for (int h = 0; h < 2; h++) for (int sf = 0; sf < 8; sf++) t += x[c][h][sf]; return __double2loint(t); }());

        // ---- EncodeIntensityStereo (:711-764): the energies are sums over the bands in order -- one ln_in per (sub-frame, sum)
        // walks them in a plain [sub-frame][band] array of the rows, one of the three terms at a time
        const int ln_is = fresh(lane0);
        uint32_t intensity_pack = 0;                   // eight 4-bit ratios of the secondary channel
        if (NCH == 2 && has_intensity) {
            const int base_band_count = cold_args()->info.base_band_count, total_band_count = cold_args()->info.total_band_count;
            double *plain = reinterpret_cast<double *>(rows);          // [8][128]
            double energy[3] = {0, 0, 0};              // lanes 0..7: sub-frame `ln_is`: L, R, total
#pragma unroll
            for (int term = 0; term < 3; term++) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) {
                        const double l = x[0][h][sf], r = x[NCH - 1][h][sf];
                        plain[sf * 128 + ln_is + 64 * h] = term == 0 ? fabs(l) : (term == 1 ? fabs(r) : fabs(l + r));
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (ln_is < 8) {
                    double e = 0;
                    for (int b = base_band_count; b < total_band_count; b++) e += plain[ln_is * 128 + b];
                    energy[term] = e;
                }
            }
            double ratio = 1;
            int quantized = 0;
            if (ln_is < 8) {
                const double energy_l = energy[0], energy_r = energy[1];
                const double energy_total = energy[2] * 2;
                const double energy_lr = energy_r + energy_l;
                const double stored = 2 * energy_l / energy_lr;
                ratio = energy_lr / energy_total;
                ratio = minmax_clamp(ratio, 0.5, 1.4142135623730951 / 2);
                quantized = 1;
                if (energy_r > 0 || energy_l > 0) {
                    while (quantized < 13 && f64_bits(HCA_IntensityRatioBoundsTableBits[quantized]) >= stored) quantized++;
                } else {
                    quantized = 0;
                    ratio = 1;
                }
            }
#pragma unroll
            for (int sf = 0; sf < 8; sf++) {
                const double rt = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ratio), sf),
                                                   __builtin_amdgcn_readlane(__double2loint(ratio), sf));
                intensity_pack |= (uint32_t)__builtin_amdgcn_readlane(quantized, sf) << (4 * sf);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int b = ln_is + 64 * h;
                    if (b >= base_band_count && b < total_band_count) {
                        x[0][h][sf] = (x[0][h][sf] + x[NCH - 1][h][sf]) * rt;
                        x[NCH - 1][h][sf] = 0;
                    }
                }
            }
        }

        // ---- CalculateScaleFactors (:673-689), ScaleSpectra (:651-671) and the band's bit costs at all sixteen resolutions
        // (CalculateUsedBits :554-597) in one pass over the band's eight coefficients.  Bands >= the coded count keep their
        // unscaled values (the HFR group averages read exactly those).
        const int ln_sc = fresh(lane0);
        int sfv[NCH][2];
        uint4 cost[NCH][2];
        uint32_t hfr_lo[NCH] = {}, hfr_hi[NCH] = {};   // the channel's HFR scales, 6 bits each: groups 0..4 | groups 5..7
#pragma unroll
        for (int c = 0; c < NCH; c++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int b = ln_sc + 64 * h;
                sfv[c][h] = 0;
                cost[c][h] = make_uint4(0, 0, 0, 0);
                if (b < coded[c]) {
                    double mx = 0;
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) {
                        const double coeff = fabs(x[c][h][sf]);
                        mx = coeff > mx ? coeff : mx;
                    }
                    const int s = find_scale_factor_lut(S, mx);
                    const double qs = T.quant_scale[s];
#pragma unroll
                    for (int sf = 0; sf < 8; sf++)
                        x[c][h][sf] = s != 0 ? minmax_clamp(x[c][h][sf] * qs, -0.999999999999, 0.999999999999) : 0.0;
                    sfv[c][h] = s;
                    cost[c][h] = band_cost_table(Q, x[c][h]);
                }
            }
            // ---- CalculateHfrGroupAverages (:766-793) + CalculateHfrScale (:795-832): one ln_sc per group walks the plain array
            if (hfr_group_count > 0 && ctype[c] != CH_STEREO_SECONDARY) {
                double *plain = reinterpret_cast<double *>(rows);
                struct { int stereo_band_count, base_band_count, bands_per_hfr_group, hfr_band_count, total_band_count; } info;
                {
                    ColdArgs k = cold_args();
                    info.stereo_band_count = k->info.stereo_band_count;
                    info.base_band_count = k->info.base_band_count;
                    info.bands_per_hfr_group = k->info.bands_per_hfr_group;
                    info.hfr_band_count = k->info.hfr_band_count;
                    info.total_band_count = k->info.total_band_count;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) plain[sf * 128 + ln_sc + 64 * h] = x[c][h][sf];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                int scale = 0;
                if (ln_sc < hfr_group_count) {
                    const int group = ln_sc;
                    const int hfr_start = info.stereo_band_count + info.base_band_count;
                    double sum = 0.0;
                    int count = 0;
                    int band = hfr_start + group * info.bands_per_hfr_group;
                    for (int i = 0; i < info.bands_per_hfr_group && band < SPSF; band++, i++) {
                        for (int sf = 0; sf < 8; sf++) sum += fabs(plain[sf * 128 + band]);
                        count += 8;
                    }
                    double avg = sum / count;
                    const int lim = min(info.hfr_band_count, info.total_band_count - info.hfr_band_count);
                    sum = 0.0;
                    count = 0;
                    band = group * info.bands_per_hfr_group;
                    for (int i = 0; i < info.bands_per_hfr_group && band < lim; band++, i++) {
                        for (int sf = 0; sf < 8; sf++) sum += fabs(plain[sf * 128 + (hfr_start - band - 1)]);
                        count += 8;
                    }
                    const double average = sum / count;
                    if (average > 0.0) {
                        const double inv = 1.0 / average;
                        avg *= inv < 1.4142135623730951 ? inv : 1.4142135623730951;
                    }
                    scale = find_scale_factor(T, avg);
                }
#pragma unroll
                for (int g = 0; g < 8; g++) {
                    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(scale, g);
                    if (g < 5) hfr_lo[c] |= v << (6 * g);
                    else hfr_hi[c] |= v << (6 * (g - 5));
                }
            }
        }
        WAVE_STOP_AFTER(3, [&] { double t = 0; for (int c = 0; c < NCH; c++) for (int h = 0; h < 2; h++) for (int sf = 0; sf < 8; sf++) t += x[c][h][sf]; return __double2loint(t); }() + (int)(cost[0][0].x ^ cost[0][1].y ^ cost[NCH - 1][0].z ^ cost[NCH - 1][1].w) + sfv[0][0] + sfv[NCH - 1][1]);

        // ---- CalculateFrameHeaderLength (:599-649): a channel's five candidate delta widths' lengths are sums packed two to a
        // register and reduced with DPP; everything after the sums is wave-uniform
        int hlb[NCH], dbits[NCH];
        auto header_lengths = [&]() __attribute__((always_inline)) {
            const int ln_h = fresh(lane0);
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                int pa = 0, pb = 0, pe = 0;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int band = ln_h + 64 * h;
                    const bool in = band < coded[c];
                    const int sf = sfv[c][h];
                    int prev = wave_shr1(sf);
                    if (h == 1) {
                        const int wrap = __builtin_amdgcn_readlane(sfv[c][0], 63);
                        prev = ln_h == 0 ? wrap : prev;
                    }
                    const bool counted = in && band >= 1;
                    const int delta = counted ? abs(sf - prev) : 0;
                    // per-lane costs <= 22, 64 lanes: 16-bit fields cannot carry into each other
                    if (counted) {
                        pa += (delta > 0 ? 7 : 1) | ((delta > 1 ? 8 : 2) << 16);
                        pb += (delta > 3 ? 9 : 3) | ((delta > 7 ? 10 : 4) << 16);
                        pe += delta > 15 ? 11 : 5;
                    }
                    pe += (in && sf != 0) ? 1 << 16 : 0;                       // non-zero scale factors: "empty channel" test
                }
                pa = wave_sum(pa);
                pb = wave_sum(pb);
                pe = wave_sum(pe);
                const int cand_len[6] = {0, 9 + (pa & 0xFFFF), 9 + (pa >> 16), 9 + (pb & 0xFFFF), 9 + (pb >> 16), 9 + (pe & 0xFFFF)};
                int len, db;
                if ((pe >> 16) == 0) { len = 3; db = 0; }
                else {
                    db = 6;
                    len = 3 + 6 * coded[c];
#pragma unroll
                    for (int k = 1; k < 6; k++)
                        if (cand_len[k] < len) { len = cand_len[k]; db = k; }
                }
                if (ctype[c] == CH_STEREO_SECONDARY) len += 32;
                else len += 6 * hfr_group_count;
                hlb[c] = len;
                dbits[c] = db;
            }
        };
        header_lengths();
        WAVE_STOP_AFTER(4, [&] { double t = 0; for (int c = 0; c < NCH; c++) for (int h = 0; h < 2; h++) for (int sf = 0; sf < 8; sf++) t += x[c][h][sf]; return __double2loint(t); }() + (int)(cost[0][0].x ^ cost[0][1].y ^ cost[NCH - 1][0].z ^ cost[NCH - 1][1].w) + hlb[0] + dbits[NCH - 1]);

        // ---- CalculateNoiseLevel (:457-485) / BinarySearchLevel (:502-523) and CalculateEvaluationBoundary (:487-500) /
        // BinarySearchBoundary (:525-552).  Slot k = (channel k >> 1, band ln_h + 64 (k & 1)).
        const int ln_se = fresh(lane0);
        constexpr int NS = 2 * NCH;
        uint64_t clo[NS], chi[NS];
        int off[NS], bnd[NS];
        bool on[NS];
#pragma unroll
        for (int k = 0; k < NS; k++) {
            const uint4 cw = cost[k >> 1][k & 1];
            clo[k] = ((uint64_t)cw.y << 32) | cw.x;
            chi[k] = ((uint64_t)cw.w << 32) | cw.z;
            bnd[k] = ln_se + 64 * (k & 1);
        }
        int level = 0, boundary = 0;
        bool too_low = false;
        int highest_band = 0;
        bool dropping = false;
        for (;;) {
            int hsum = 48;
#pragma unroll
            for (int c = 0; c < NCH; c++) hsum += hlb[c];
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int sf = sfv[k >> 1][k & 1];
                on[k] = sf != 0 && bnd[k] < coded[k >> 1];
                off[k] = 2 - 5 * sf / 2;
            }
            auto probe = [&](int noise_level) __attribute__((always_inline)) -> int {
                int partial = 0;
#pragma unroll
                for (int k = 0; k < NS; k++) {
                    const int res = T.res_curve[min(max(noise_level + off[k], 0), 58)];
                    const uint64_t half = res >= 8 ? chi[k] : clo[k];
                    partial += on[k] ? (int)((half >> (8 * (res & 7))) & 0xFFu) : 0;
                }
                return wave_sum(partial) + hsum;
            };
            int low = 0, high = 255, mid_value = 0;
            while (low != high) {
                const int mid = (low + high) / 2;
                mid_value = probe(mid);
                if (mid_value > available) low = mid + 1;
                else high = mid;
            }
            level = (low == 255 && mid_value > available) ? -1 : low;
            if (level >= 0) break;
            // CalculateNoiseLevel (:469-484): drop the two highest bands and try again
            if (!dropping) highest_band = cold_args()->info.base_band_count + cold_args()->info.stereo_band_count - 1;
            dropping = true;
            highest_band -= 2;
            if (highest_band < 0) { too_low = true; break; }
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int b = ln_se + 64 * h;
                    if (b == highest_band + 1 || b == highest_band + 2) sfv[c][h] = 0;
                }
            header_lengths();
        }
        if (too_low) {                                 // InvalidDataException("Bitrate is set too low.")
            if (ln_se == 0 && cold_args()->status) atomicOr(cold_args()->status, 4);
            level = 255;
        }
        if (level > 0 && !too_low) {
            // BinarySearchBoundary (:525-552) probes CalculateUsedBits(level, boundary): the bands below the boundary at
            // level - 1, the others at level -- the frame's bits at `level` plus, for every band below the boundary, what the
            // band costs more at level - 1: ONE exclusive scan over the bands gives the value of every possible probe and the
            // search's dependent probes are ln_se reads -- the same probes and decisions in the reference's order (the bits
            // need not be monotone in the boundary and nothing here assumes it)
            int hsum = 48;
#pragma unroll
            for (int c = 0; c < NCH; c++) hsum += hlb[c];
            int at_level = 0, d_lo = 0, d_hi = 0;
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int r1 = T.res_curve[min(max(level + off[k], 0), 58)], r0 = T.res_curve[min(max(level - 1 + off[k], 0), 58)];
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
            const int ex_lo = base + in_lo - d_lo, ex_hi = base + all_lo + in_hi - d_hi;      // probe(level, band)
            auto probe_boundary = [&](int eb) {                       // eb is wave-uniform, 0 .. 127
                const int pa = __builtin_amdgcn_readlane(ex_lo, __builtin_amdgcn_readfirstlane(eb) & 63);
                const int pb = __builtin_amdgcn_readlane(ex_hi, __builtin_amdgcn_readfirstlane(eb) & 63);
                return eb < 64 ? pa : pb;
            };
            int lo2 = 0, hi2 = 127;
            while (abs(hi2 - lo2) > 1) {
                const int mid = (lo2 + hi2) / 2;
                const int mid_value2 = probe_boundary(mid);
                if (available < mid_value2) hi2 = mid - 1;
                else lo2 = mid;
            }
            if (lo2 == hi2) boundary = lo2 < 127 ? lo2 : -1;
            else boundary = probe_boundary(hi2) > available ? lo2 : hi2;
        }
        WAVE_STOP_AFTER(5, [&] { double t = 0; for (int c = 0; c < NCH; c++) for (int h = 0; h < 2; h++) for (int sf = 0; sf < 8; sf++) t += x[c][h][sf]; return __double2loint(t); }() + level + boundary);
        if (boundary < 0) {                            // NotImplementedException in the reference
            if (ln_se == 0 && cold_args()->status) atomicOr(cold_args()->status, 8);
            boundary = 0;
        }

        // ---- CalculateFrameResolutions (:441-455)
        const int ln_q = fresh(lane0);
        int res[NCH][2];
#pragma unroll
        for (int c = 0; c < NCH; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int b = ln_q + 64 * h;
                res[c][h] = b < coded[c] ? resolution_of(T, sfv[c][h], b < boundary ? level - 1 : level) : 0;
            }

        // ---- QuantizeSpectra (:420-439): code and length of every coefficient in 16 bits (code << 4 | bits; 12 + 4 at most),
        // written where the ln_q that owns the code in stream order -- (sub-frame, channel, band): WriteSpectra
        // (CriHcaPacking.cs:238-260) -- will read it.  The quantiser's constants need no table: QuantizerInverseStepSize[r] =
        // ResolutionMaxValue[r] + 0.5 (CriHcaTables.cs:57), shiftUp = inv + 1, shiftDown = (int)(inv + 0.5) = max + 1, and
        // QuantizedSpectrumMaxBits[r] = r - 3 from resolution 8 on.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the rows' last readers are done
        {
            const int turn_addr = (ln_q / PT) * Lay::STRIDE + (ln_q % PT) * 2;
#pragma unroll
            for (int c = 0; c < NCH; c++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int r = res[c][h];
                    const int maxv = r < 8 ? r : (1 << max(r - 4, 0)) - 1;
                    const double inv = (double)maxv + 0.5;
                    const double up = inv + 1;
                    const uint8_t *pair_row = T.enc_pair[min(r, 7)];
#pragma unroll
                    for (int sf = 0; sf < 8; sf++) {
                        const int q = trunc_i(x[c][h][sf] * inv + up) - (maxv + 1);
                        const unsigned small_pair = pair_row[(q + 8) & 15];                   // value << 4 | bits
                        const unsigned mag = (unsigned)abs(q);
                        const unsigned large_pair = q != 0 ? ((((mag << 1) | (q > 0 ? 0u : 1u)) << 4) | (unsigned)(r - 3))
                                                           : (unsigned)(r - 4);
                        const unsigned pair = r == 0 ? 0u : (r < 8 ? small_pair : large_pair);
                        const int base_slot = ((sf * NCH + c) * 128 + 64 * h);          // + ln_q: a multiple of 64 (PT divides 64)
                        *reinterpret_cast<uint16_t *>(rows + (base_slot / PT) * Lay::STRIDE + turn_addr) = (uint16_t)pair;
                    }
                }
        }
        const int ln_pk = fresh(lane0);
        for (int i = ln_pk; i < fwords; i += 64) fbuf[i] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WAVE_STOP_AFTER(6, fbuf[ln_pk] + reinterpret_cast<uint32_t *>(rows)[ln_pk * 7] + level);

        const int ln_hd = fresh(lane0);
        auto put_bits = [&](int bit, unsigned value, int nbits) __attribute__((always_inline)) {
            const uint64_t win = (uint64_t)value << ((64 - nbits - (bit & 31)) & 63);      // nbits = 0: the value is 0
            const unsigned hi = (unsigned)(win >> 32), lo = (unsigned)win;
            atomicOr(&fbuf[bit >> 5], hi);
            atomicOr(&fbuf[(bit >> 5) + 1], lo);
        };
        int header_bits = 32;
#pragma unroll
        for (int c = 0; c < NCH; c++) header_bits += hlb[c];
        if (!too_low) {
            // ---- PackFrame (CriHcaPacking.cs:17-58): sync word, noise level, evaluation boundary
            if (ln_hd == 0) atomicOr(&fbuf[0], 0xFFFF0000u | ((unsigned)level << 7) | (unsigned)boundary);
            // WriteScaleFactors (:262-295): bands in order = lanes in order, first half then second half
            int bit0 = 32;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int db = dbits[c];
                unsigned code[2] = {0, 0};
                int nb2[2] = {0, 0};
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int band = ln_hd + 64 * h;
                    const int sf = sfv[c][h];
                    int prev = wave_shr1(sf);
                    if (h == 1) {
                        const int wrap = __builtin_amdgcn_readlane(sfv[c][0], 63);
                        prev = ln_hd == 0 ? wrap : prev;
                    }
                    if (band == 0) {                           // the 3-bit delta width, then the first scale factor
                        code[h] = (unsigned)db;
                        nb2[h] = 3;
                        if (db != 0) { code[h] = (code[h] << 6) | (unsigned)sf; nb2[h] = 9; }
                    } else if (band < coded[c] && db != 0) {
                        if (db == 6) { code[h] = (unsigned)sf; nb2[h] = 6; }
                        else {
                            const int max_delta = (1 << (db - 1)) - 1;
                            const int delta = sf - prev;
                            if (abs(delta) > max_delta) { code[h] = ((((1u << db) - 1)) << 6) | (unsigned)sf; nb2[h] = db + 6; }
                            else { code[h] = (unsigned)(max_delta + delta); nb2[h] = db; }
                        }
                    }
                }
                const int incl0 = wave_inclusive_scan(nb2[0]);
                const int total0 = __builtin_amdgcn_readlane(incl0, 63);
                const int incl1 = wave_inclusive_scan(nb2[1]);
                const int total1 = __builtin_amdgcn_readlane(incl1, 63);
                put_bits(bit0 + incl0 - nb2[0], code[0], nb2[0]);
                put_bits(bit0 + total0 + incl1 - nb2[1], code[1], nb2[1]);
                if (ln_hd < 8) {                                // intensity / HFR scales follow the scale factors
                    const int at = bit0 + total0 + total1;
                    if (ctype[c] == CH_STEREO_SECONDARY) put_bits(at + 4 * ln_hd, (intensity_pack >> (4 * ln_hd)) & 15u, 4);
                    else if (ln_hd < hfr_group_count)
                        put_bits(at + 6 * ln_hd, (ln_hd < 5 ? hfr_lo[c] >> (6 * ln_hd) : hfr_hi[c] >> (6 * (ln_hd - 5))) & 63u, 6);
                }
                bit0 += hlb[c];
            }
            const int ln_ws = fresh(lane0);
            // ---- WriteSpectra (:238-260): PT consecutive codes per ln_ws
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t cw[PT / 2];
            {
                const uint4 *src = reinterpret_cast<const uint4 *>(rows + ln_ws * Lay::STRIDE);
#pragma unroll
                for (int i = 0; i < PT / 8; i++) {
                    const uint4 v = src[i];
                    cw[4 * i] = v.x;
                    cw[4 * i + 1] = v.y;
                    cw[4 * i + 2] = v.z;
                    cw[4 * i + 3] = v.w;
                }
            }
            // A ln_ws's codes are consecutive in the stream.  They are concatenated inside the ln_ws as a tree -- pairs (at most 24
            // bits), then fours (at most 48 bits, with their lengths) -- and every four is ORed into the frame at its
            // bit position as three dwords, whatever they hold: no branch per code, and an LDS atomic that ORs zeros costs
            // less than finding out that it would (the first and last dword of a ln_ws are shared with its neighbours).
            constexpr int UNITS = PT / 4;
            uint64_t unit[UNITS];
            int ulen[UNITS];
            int local = 0;
#pragma unroll
            for (int j = 0; j < UNITS; j++) {
                uint32_t u2[2];
                int n2[2];
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint32_t w = cw[2 * j + e];
                    const uint32_t c0 = (w >> 4) & 0xFFFu, c1 = w >> 20;
                    const int n0 = (int)(w & 15u), n1 = (int)((w >> 16) & 15u);
                    u2[e] = (c0 << n1) | c1;
                    n2[e] = n0 + n1;
                }
                unit[j] = ((uint64_t)u2[0] << n2[1]) | u2[1];
                ulen[j] = n2[0] + n2[1];
                local += ulen[j];
            }
            const int incl = wave_inclusive_scan(local);
            int bit = header_bits + incl - local;
#pragma unroll
            for (int j = 0; j < UNITS; j++) {
                const uint64_t v = unit[j] << ((64 - ulen[j]) & 63);           // left-aligned (an empty unit is 0)
                const uint32_t v_hi = (uint32_t)(v >> 32), v_lo = (uint32_t)v;
                const int sh = bit & 31;
                uint32_t *dst = fbuf + (bit >> 5);
                atomicOr(dst, v_hi >> sh);
                atomicOr(dst + 1, (uint32_t)(v >> sh));
                atomicOr(dst + 2, (uint32_t)(((uint64_t)v_lo << 32) >> sh));
                bit += ulen[j];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WAVE_STOP_AFTER(7, fbuf[lane0]);

        const int ln_crc = fresh(lane0);
        // ---- WriteChecksum (:231-236): CRC-16 (poly 0x8005, init 0) over the first frame_size - 2 bytes: per-ln_crc partial
        // CRCs of consecutive chunks, shifted to their place with x^(8k) mod P and XORed
        {
            const int crc_begin = ln_crc * crc_chunk, crc_end = min(crc_begin + crc_chunk, nbytes);
            unsigned crc = 0;
            for (int i = crc_begin; i < crc_end; i++) {
                const unsigned byte = (fbuf[i >> 2] >> (24 - 8 * (i & 3))) & 0xFFu;
                // eight shift-and-xor steps of x^16 + x^15 + x^2 + 1 at once: with t = the byte entering the register,
                // t * x^16 mod P = t << 1 ^ t << 2 ^ (parity(t) ? 0x8003 : 0)   (tests/test_oracle_hca.py)
                const unsigned t = ((crc >> 8) ^ byte) & 0xFFu;
                crc = ((crc << 8) & 0xFFFFu) ^ ((__popc(t) & 1) ? 0x8003u : 0u) ^ (t << 1) ^ (t << 2);
            }
            unsigned part = (crc_begin < crc_end) ? gf_mul(crc, crc_shift) : 0u;
            const unsigned total = (unsigned)wave_xor((int)part) & 0xFFFFu;
            if (ln_crc == 0) {
                const int pos = nbytes;                // big-endian 16-bit value at the last two bytes
                fbuf[pos >> 2] |= (total >> 8) << (24 - 8 * (pos & 3));
                fbuf[(pos + 1) >> 2] |= (total & 0xFF) << (24 - 8 * ((pos + 1) & 3));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WAVE_STOP_AFTER(8, fbuf[ln_crc]);

        const int ln_st = fresh(lane0);
        // ---- store the frame: whole aligned dwords (the frame starts at any byte: its k-th dword is a funnel shift of two
        // big-endian words of fbuf), the few bytes before the first and after the last aligned dword one by one
        {
            uint8_t *dst = frames + (int64_t)stream * frames_pitch + (int64_t)frame * frame_size;
            const int lead = (int)((4 - (reinterpret_cast<uintptr_t>(dst) & 3)) & 3);       // bytes before the first aligned dword
            const int ndw = (frame_size - lead) / 4;
            auto byte_at = [&](int b) { return (fbuf[b >> 2] >> (24 - 8 * (b & 3))) & 0xFFu; };
            uint32_t *dw = reinterpret_cast<uint32_t *>(dst + lead);
            for (int k = ln_st; k < ndw; k += 64) {
                const int b = lead + 4 * k;                                                   // frame byte of the dword's first byte
                const uint32_t hi = fbuf[b >> 2], lo = fbuf[(b >> 2) + 1];
                const int sh = 8 * (b & 3);
                const uint32_t be = sh ? (hi << sh) | (lo >> (32 - sh)) : hi;                 // frame bytes b .. b+3, first byte on top
                dw[k] = bswap32(be);
            }
            const int tail0 = lead + 4 * ndw;
            if (ln_st < lead) dst[ln_st] = (uint8_t)byte_at(ln_st);
            if (ln_st >= 32 && ln_st - 32 < frame_size - tail0) dst[tail0 + ln_st - 32] = (uint8_t)byte_at(tail0 + ln_st - 32);
        }
    }
}

// bytes of dynamic LDS one wave of the kernel needs for streams of this shape
static size_t wave_lds_bytes(const DeviceInfo &info)
{
    const size_t fwords = (size_t)(((info.frame_size + 3) / 4 + 3) & ~1);
    const size_t codes = info.nch == 2 ? Layout<2>::CODES_BYTES : Layout<1>::CODES_BYTES;
    return (std::max<size_t>(ROWS_BYTES, codes + fwords * 4 + 8) + 15) & ~(size_t)15;
}

bool encode_wave_kernel_takes(const DeviceInfo &info)
{
    // one or two channels; frames whose bits fit next to the turned codes without pushing a workgroup past the LDS of a CU
    return (info.nch == 1 || info.nch == 2) && WG_WAVES * wave_lds_bytes(info) + sizeof(WaveShared) + 64 <= (WG_WAVES > 4 ? 80 : 64) * 1024;
}

int launch_encode_wave(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams, const PcmMap &map,
                       const DeviceInfo &info, uint8_t *d_frames, int64_t frames_pitch, const uint16_t *d_crc_pow,
                       int *d_status, hipStream_t stream, int first_frame, int end_frame, int frames_per_run_override)
{
    const int frame_span = end_frame - first_frame;
    const size_t wave_bytes = wave_lds_bytes(info);
    const size_t lds = WG_WAVES * wave_bytes;
    // frames per wave: long runs amortise the per-workgroup set-up (tables), short ones keep small inputs spread over the chip
    const int64_t total = (int64_t)nstreams * frame_span;
    int per_run = (int)std::min<int64_t>(MAX_WAVE_FRAMES, std::max<int64_t>(1, total / 16384));
    if (frames_per_run_override > 0) per_run = std::min(frames_per_run_override, 64);
    per_run = std::min(per_run, frame_span);
    const int runs = (frame_span + per_run - 1) / per_run;
    const int64_t total_runs = (int64_t)nstreams * runs;
    const unsigned grid = (unsigned)((total_runs + WG_WAVES - 1) / WG_WAVES);
    WaveArgs args{};
    args.pcm = d_pcm;
    args.stream_pitch = stream_pitch;
    args.ch_pitch = ch_pitch;
    args.frames = d_frames;
    args.frames_pitch = frames_pitch;
    args.crc_pow = d_crc_pow;
    args.status = d_status;
    args.frames_per_run = per_run;
    args.runs_per_stream = runs;
    args.total_runs = (int)total_runs;
    args.first_frame = first_frame;
    args.end_frame = end_frame;
    args.wave_bytes = (int)wave_bytes;
    args.map = map;
    args.info = info;
    if (info.nch == 2) {
        if (lds > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_encode_wave_kernel<2>, lds));
        hipLaunchKernelGGL(hca_encode_wave_kernel<2>, dim3(grid), dim3(WG_THREADS), lds, stream, args);
    } else {
        if (lds > 32 * 1024) VGA_HIP_TRY(allow_dynamic_lds(hca_encode_wave_kernel<1>, lds));
        hipLaunchKernelGGL(hca_encode_wave_kernel<1>, dim3(grid), dim3(WG_THREADS), lds, stream, args);
    }
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace hca
}  // namespace vga
