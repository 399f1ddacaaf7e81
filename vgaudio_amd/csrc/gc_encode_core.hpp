// gc_encode_core.hpp -- per-lane arithmetic of the GC-ADPCM frame encoder, shared by
// the gfx950 kernel (gcadpcm_kernels.hip) and the host-side lane emulator that the
// CPU test-suite uses to check these shortcuts against the literal reference formula
// (tests/host/gc_encode_emulator.cpp).  No codec result is ever produced on the CPU in
// the product path; the host build of this header exists for testing only.
//
// Reference: VGAudio/Codecs/GcAdpcm/GcAdpcmEncoder.cs:96-171 (DspEncodeCoef).
//
// Three exact reformulations of the reference's arithmetic are used; each is proven in
// LABNOTES.md 4.1 ("GC-ADPCM encode: exact shortcuts") and exercised exhaustively by the tests:
//  (S1) pre-scan (:107-124): the signed maxDistance only needs max(d), min(d) over the
//       frame unless +M and -M both occur (rare; sequential rescan then), and the
//       halving loop has a closed form in clz().
//  (S2) quantise (:140-144): (int)((double)((float)d / 2^k) +- 0.4999999f) equals
//       (r + 2^(k-1) - 1 + (d < 0)) >> k  (arithmetic shift) with r = (int)(float)d.
//  (S3) totalDistance (:161-162) is an exact integer; 32-bit accumulation is exact under the
//       overflow bound checked per frame (pass_fast); otherwise the pass is redone literally
//       with a 64-bit sum.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define VGA_HD __host__ __device__ __forceinline__
#else
#define VGA_HD inline
#endif

// 24-bit multiply (operands are 16/17-bit here) and an optimisation barrier that keeps hipcc from
// re-associating the negated coefficients back into "mad, then subtract" (one extra dependent op).
#if defined(__HIP_DEVICE_COMPILE__)
#define VGA_MUL24(a, b) __mul24((a), (b))
#define VGA_OPAQUE(v) asm("" : "+v"(v))
// acc + e*e as ONE instruction (hipcc otherwise re-associates the 14 sums into mul, mul, add3)
static __device__ __forceinline__ uint32_t vga_mad24_acc(int e, uint32_t acc)
{
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %1, %2" : "=v"(r) : "v"(e), "v"(acc));
    return r;
}
#define VGA_MAD24_ACC(e, acc) vga_mad24_acc((e), (acc))
// e * e for |e| <= 65535 as an unsigned 32-bit value.  NOT __mul24: the device library writes that one as a signed C product,
// the compiler may then assume it stays below 2^31 -- it folded the first two squares of a 64-bit error sum into one 32-bit
// mad (round 5: every frame of a full-scale square came out with the wrong predictor).
static __device__ __forceinline__ uint32_t vga_square24(int e)
{
    uint32_t r;
    asm("v_mul_i32_i24 %0, %1, %1" : "=v"(r) : "v"(e));
    return r;
}
#else
#define VGA_MAD24_ACC(e, acc) ((acc) + (uint32_t)(e) * (uint32_t)(e))
#define VGA_MUL24(a, b) ((a) * (b))
#define VGA_OPAQUE(v) ((void)0)
#endif

namespace vga {
namespace gc {

VGA_HD int imin(int a, int b) { return a < b ? a : b; }
VGA_HD int imax(int a, int b) { return a > b ? a : b; }
VGA_HD int clamp16i(int v) { return imin(imax(v, -32768), 32767); }
VGA_HD int clamp4i(int v) { return imin(imax(v, -8), 7); }

// p / 2048 with C#'s truncation toward zero, three ops: sign bit, mad, shift
VGA_HD int div2048(int p)
{
    int sgn = (int)((unsigned)p >> 31);
    VGA_OPAQUE(sgn);                                   // keeps it a 0/1 factor: v_lshrrev, v_mad_u32_u24, v_ashrrev
    return (int)((unsigned)p + (unsigned)VGA_MUL24(sgn, 2047)) >> 11;
}

VGA_HD int bit_length(unsigned v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return v ? 32 - __builtin_clz(v) : 0;
#else
    int n = 0;
    while (v) { n++; v >>= 1; }
    return n;
#endif
}

// number of `maxDistance /= 2` steps of GcAdpcmEncoder.cs:119-123 for a signed maxDistance
// in [-32768, 32767]  (closed form of: while (sp <= 12 && (md > 7 || md < -8)) { md /= 2; sp++; })
VGA_HD int halvings(int md)
{
    if (md >= 0) {
        const int n = bit_length((unsigned)md) - 3;
        return n > 0 ? n : 0;
    }
    const unsigned a = (unsigned)(-md);            // truncating /2 on a negative = halving the magnitude
    int n = bit_length(a) - 4;
    if (n < 0) n = 0;
    if ((a >> n) > 8u) n++;
    return n;
}

// literal sequential pre-scan (GcAdpcmEncoder.cs:107-115); used for the rare +M/-M tie
VGA_HD int prescan_sequential(const int (&x)[16], int c0, int c1)
{
    int max_distance = 0;
    for (int s = 0; s < 14; s++) {
        const int predicted = (x[s] * c1 + x[s + 1] * c0) / 2048;
        int distance = clamp16i(x[s + 2] - predicted);
        const int ad = distance < 0 ? -distance : distance;
        const int am = max_distance < 0 ? -max_distance : max_distance;
        if (ad > am) max_distance = distance;
    }
    return max_distance;
}

// running max/min of the unclamped pre-scan distances of samples [s_begin, s_end)
VGA_HD void prescan_range(const int (&x)[16], int c0, int c1, int s_begin, int s_end, int &dmax, int &dmin)
{
    for (int s = s_begin; s < s_end; s++) {
        const int predicted = (x[s] * c1 + x[s + 1] * c0) / 2048;
        const int d = x[s + 2] - predicted;
        dmax = imax(dmax, d);
        dmin = imin(dmin, d);
    }
}

// Scale power of the FIRST quantise pass (value after the do-loop's first ++), given the
// frame-wide max/min of the unclamped distances.  Returns -100 when the signed maximum is
// ambiguous (+M and -M both present, M > 0): caller must use prescan_sequential().
VGA_HD int first_scale_power_from_range(int dmax, int dmin)
{
    // straight-line version of halvings() for both signs (|1 keeps clz defined; bit_length(0|1) - 3 < 0 -> 0)
    const int pos = imax(clamp16i(dmax), 0);
    const int neg = imax(-clamp16i(dmin), 0);
    const int hp = imax(bit_length((unsigned)pos | 1u) - 3, 0);
    const int nn = imax(bit_length((unsigned)neg | 1u) - 4, 0);
    const int hn = nn + ((((unsigned)neg >> nn) > 8u) ? 1 : 0);
    // equal magnitudes with different halving counts: the sign of the reference's maxDistance depends on
    // which of +M / -M came first
    if (pos == neg && hp != hn) return -100;
    const int n = pos > neg ? hp : hn;
    return imax(n - 1, 0);                          // (n<=1 ? -1 : n-2) + 1
}

VGA_HD int first_scale_power_from_md(int md)
{
    const int n = halvings(md);
    return n <= 1 ? 0 : n - 1;
}

// scalePower after the overflow bump loop of GcAdpcmEncoder.cs:166-168
VGA_HD int apply_bumps(int scale_power, int max_overflow)
{
    for (int v = max_overflow + 8; v > 256; v >>= 1)
        if (++scale_power >= 12) scale_power = 11;
    return scale_power;
}

struct PassOut {
    int q[14];           // the quantised samples, -8..7 (adpcmOut of GcAdpcmEncoder.cs:155)
    uint64_t total;      // totalDistance
    int max_overflow;
    int o12, o13;        // reconstructed samples 12, 13
    unsigned hist_pair;  // (o12 & 0xFFFF) | (o13 << 16): the next frame's history, as the kernel hands it on
    bool exact;          // false: the fast pass could not prove itself exact -> redo with pass_literal
};

VGA_HD uint32_t bswap32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}

// The 8 frame bytes as two little-endian dwords: byte 0 = header (predictor<<4 | scale),
// bytes 1..7 = nibbles hi-first (GcAdpcmEncoder.cs:83-93).  The nibbles are accumulated SIGNED
// (w = w*16 + q, one shift-add each); adding 0x888..8 turns the sum into the packing of the biased
// nibbles q+8 and the XOR with 0x888..8 un-biases them.
VGA_HD void pack_frame(const int (&q)[14], int predictor, int scale_power, uint32_t &d0, uint32_t &d1)
{
    uint32_t wa = 0, wb = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 14; s++) {
        if (s < 6) wa = (wa << 4) + (uint32_t)q[s];
        else       wb = (wb << 4) + (uint32_t)q[s];
    }
    wa = ((wa + 0x00888888u) ^ 0x00888888u) & 0x00FFFFFFu;
    wb = (wb + 0x88888888u) ^ 0x88888888u;
    const uint32_t header = (uint32_t)((predictor << 4) | (scale_power & 0xF));
    d0 = bswap32((header << 24) | wa);
    d1 = bswap32(wb);
}
VGA_HD void frame_words(const PassOut &r, int predictor, int scale_power, uint32_t &d0, uint32_t &d1)
{
    pack_frame(r.q, predictor, scale_power, d0, d1);
}

// Literal quantise pass: the reference's own float/double formula (:127-164), 64-bit total.
VGA_HD PassOut pass_literal(const int (&x)[16], int c0, int c1, int scale_power)
{
    PassOut r;
    const int k = scale_power + 11;
    const int scale = 1 << k;
    union { uint32_t u; float f; } inv;
    inv.u = (uint32_t)(127 - k) << 23;             // exact 2^-k: the f32 divide by 2^k is this multiply
    uint64_t total = 0;
    int max_overflow = 0;
    int o0 = x[0], o1 = x[1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 14; s++) {
        const int predicted = o0 * c1 + o1 * c0;
        const int distance = x[s + 2] * 2048 - predicted;
        const float fq = (float)distance * inv.f;
        const double half = (distance > 0) ? (double)0.4999999f : -(double)0.4999999f;
        const int unclamped = (int)((double)fq + half);
        const int q = clamp4i(unclamped);
        const int ov = unclamped - q;
        max_overflow = imax(max_overflow, ov < 0 ? -ov : ov);
        r.q[s] = q;
        const int corrected = predicted + q * scale;
        const int recon = clamp16i((corrected + 1024) >> 11);
        const int d = x[s + 2] - recon;
        total += (uint64_t)(uint32_t)(d * d);
        o0 = o1;
        o1 = recon;
    }
    r.total = total; r.max_overflow = max_overflow; r.o12 = o0; r.o13 = o1;
    r.hist_pair = (unsigned)(o0 & 0xFFFF) | ((unsigned)o1 << 16);
    r.exact = true;
    return r;
}

// r = (int)(float)d : the reference's int -> float rounding (round-to-nearest-even to 24 bits),
// back as an integer.  |d| >= 2^31 - 64 rounds to 2^31, which does not fit: the hardware
// conversion saturates (host: same by hand) and the caller's overflow limit rejects the frame.
VGA_HD int round_through_f32(int d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(float)d;                       // v_cvt_f32_i32 + v_cvt_i32_f32 (saturating)
#else
    const float f = (float)d;
    if (f >= 2147483648.0f) return 2147483647;
    return (int)f;
#endif
}

// Fast quantise pass: integer-only, 15 VALU ops per sample of which 8 are on the dependent chain
// (mad, cvt, cvt, add3, ashr, med3, lshl_add, med3) instead of the f32/f64 detour.
//   d      = in*2048 - (o0*c1 + o1*c0)                       (two mads with negated coefs)
//   r      = (int)(float)d                                   the reference's float rounding
//   u      = (r + 2^(k-1) - 1 + (d<0)) >> k                  unclamped nibble  (S2)
//   q      = clamp(u, -8, 7)
//   recon  = clamp16(((in*2048 + 1024 - d) >> 11) + (q << (k-11)))  (in*2048 + 1024 - d == predicted + 1024)
// The nibbles leave the pass unpacked (r.q): the kernel's helper wave packs the winner's frame (pack_frame).
// The overflow is recovered from the running max/min of u (one max3/min3 per two samples).
// r.exact == false (frame must be redone with pass_literal) when the 32-bit sum of squared
// errors could overflow (S3): with |c0|+|c1| <= 32767 the predictor cannot wrap, and then
// |in - recon| <= (ov + 1/2) * 2^(k-11) + 2 where ov is the pass's max overflow; we require that
// bound to stay <= 17 500 (14 * 17500^2 < 2^32), which also rules out int32 overflow in u.
// in2048v[s] = x[s + 2] * 2048 and in2048p[s] = x[s + 2] * 2048 + 1024 are supplied by the caller (the
// kernel's helper wave precomputes them per tile).
// WIDE_TOTAL: the error sum in 64 bits (a multiply and an add with carry per sample instead of one mad) -- exact whatever
// the overflow as long as the predictor cannot wrap (|c0| + |c1| <= 32767: then |d| < 2^30 + 2^26, u cannot leave int32, and
// |in - recon| <= 65535 squares into 32 bits).  For the one case the 32-bit sum cannot serve: a pass at the cap (scale 12 ends
// the reference's loop whatever it overflowed, :170) whose overflow exceeds 3 -- loud noise, clipped waves.
// NO_ROUND (round 5): the pass without the detour through f32 -- r = d instead of r = (int)(float)d, two conversions a
// sample less.  (int)(float)d IS d while |d| < 2^24, and a sample with |d| >= 2^24 shows: its unclamped nibble has
// |u| >= 2^(24-k), so the pass's overflow is at least 2^(13 - scale_power) - 8.  A pass whose overflow stays BELOW that bound
// (pass_no_round_is_exact) therefore never met such a sample -- by induction over the samples it is the exact pass, nibble
// for nibble -- and one that does not must be run again with the conversions.  The kernel takes this form for a frame when
// every lane of the wave quantises at scale 9 or below (70 % of the synthetic set's wave-frames).
template <bool WIDE_TOTAL, bool NO_ROUND = false>
VGA_HD PassOut pass_fast_core_t(const int (&x)[16], const int (&in2048v)[14], const int (&in2048p)[14], int c0, int c1,
                                int scale_power)
{
    PassOut r;
    uint64_t total64 = 0;
    const int k = scale_power + 11;
    const int km11 = scale_power;
    int bias = (1 << (k - 1)) - 1;
    int nc0 = -c0, nc1 = -c1;
    VGA_OPAQUE(bias);
    VGA_OPAQUE(nc0);
    VGA_OPAQUE(nc1);
    uint32_t total = 0;
    int umax = 0, umin = 0;
    int u_prev = 0;
    int o0 = x[0], o1 = x[1];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 14; s++) {
        int base = VGA_MUL24(o0, nc1) + in2048v[s];                 // off the dependent chain (o0 is one step old)
        VGA_OPAQUE(base);
        const int d = VGA_MUL24(o1, nc0) + base;                    // == in2048 - predicted (mod 2^32)
        const int rd = NO_ROUND ? d : round_through_f32(d);
        const int u = (int)((uint32_t)rd + (uint32_t)bias + ((uint32_t)d >> 31)) >> k;
        const int q = imin(imax(u, -8), 7);
        if (s & 1) {
            umax = imax(imax(umax, u_prev), u);
            umin = imin(imin(umin, u_prev), u);
        }
        u_prev = u;
        r.q[s] = q;
        // (predicted + 1024 + q * 2^k) >> 11 with the shift taken off the dependent chain: q * 2^k is a
        // multiple of 2^11 (k >= 11), so it passes through the floor
        const int pr11 = (int)((uint32_t)in2048p[s] - (uint32_t)d) >> 11;
        const int recon = clamp16i(pr11 + (int)((uint32_t)q << km11));
        const int e = x[s + 2] - recon;
#if defined(__HIP_DEVICE_COMPILE__)
        if (WIDE_TOTAL) total64 += (uint64_t)vga_square24(e);           // |e| <= 65535: the product's low 32 bits are the square
#else
        if (WIDE_TOTAL) total64 += (uint64_t)((int64_t)e * (int64_t)e);
#endif
        else total = VGA_MAD24_ACC(e, total);                       // total += e * e, one v_mad_i32_i24
        o0 = o1;
        o1 = recon;
    }
    r.hist_pair = (unsigned)(o0 & 0xFFFF) | ((unsigned)o1 << 16);
    const int ov = imax(imax(umax - 7, -8 - umin), 0);
    const int ac0 = c0 < 0 ? -c0 : c0, ac1 = c1 < 0 ? -c1 : c1;
    r.exact = ac0 + ac1 <= 32767 && (WIDE_TOTAL || (ov <= 17497 && (((2 * ov + 1) << (k - 11)) <= 34996)));
    r.total = WIDE_TOTAL ? total64 : (uint64_t)total;
    r.max_overflow = ov;
    r.o12 = o0; r.o13 = o1;
    return r;
}
VGA_HD PassOut pass_fast_core(const int (&x)[16], const int (&in2048v)[14], const int (&in2048p)[14], int c0, int c1,
                              int scale_power)
{
    return pass_fast_core_t<false>(x, in2048v, in2048p, c0, c1, scale_power);
}
VGA_HD PassOut pass_fast_core_no_round(const int (&x)[16], const int (&in2048v)[14], const int (&in2048p)[14], int c0, int c1,
                                       int scale_power)
{
    return pass_fast_core_t<false, true>(x, in2048v, in2048p, c0, c1, scale_power);
}
// the bound under which a NO_ROUND pass is the exact pass (scale_power <= 12: the bound is positive up to scale 9)
VGA_HD bool pass_no_round_is_exact(int scale_power, int max_overflow)
{
    return scale_power <= 9 && max_overflow < (1 << (13 - scale_power)) - 8;
}
VGA_HD PassOut pass_fast_core_wide(const int (&x)[16], const int (&in2048v)[14], const int (&in2048p)[14], int c0, int c1,
                                   int scale_power)
{
    return pass_fast_core_t<true>(x, in2048v, in2048p, c0, c1, scale_power);
}

VGA_HD PassOut pass_fast(const int (&x)[16], int c0, int c1, int scale_power)
{
    int in2048v[14], in2048p[14];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 0; s < 14; s++) {
        in2048v[s] = x[s + 2] * 2048;
        in2048p[s] = in2048v[s] + 1024;
    }
    return pass_fast_core(x, in2048v, in2048p, c0, c1, scale_power);
}

// ---- speculative two-candidate resolution --------------------------------------------
// Lane A ran the pass at s1, lane B at s1+1 (the scale the reference tries next when A
// overflows without a bump).  From the two overflows alone every lane can tell which of
// the two passes (if any) is the one the reference's do-loop ends on.
struct Resolve {
    bool final_a;        // the reference stops after the pass at s1
    bool final_b;        // ... after the pass at s1+1
    int resume_sp;       // otherwise: value of scalePower when the do-loop is re-entered
};

VGA_HD Resolve resolve_candidates(int s1, int ov_a, int ov_b)
{
    Resolve z;
    z.final_a = z.final_b = false;
    z.resume_sp = 0;
    const int sp_a = apply_bumps(s1, ov_a);
    if (!(sp_a < 12 && ov_a > 1)) { z.final_a = true; return z; }
    if (sp_a != s1) { z.resume_sp = sp_a; return z; }        // bumped: next pass is not s1+1
    const int s2 = s1 + 1;
    const int sp_b = apply_bumps(s2, ov_b);
    if (!(sp_b < 12 && ov_b > 1)) { z.final_b = true; return z; }
    z.resume_sp = sp_b;
    return z;
}

// Same decision when neither overflow can trigger the bump loop (max_overflow + 8 <= 256):
// straight-line, no loops -- the kernel's common path.
VGA_HD Resolve resolve_candidates_nobump(int s1, int ov_a, int ov_b)
{
    Resolve z;
    z.final_a = !(s1 < 12 && ov_a > 1);
    z.final_b = !z.final_a && !(s1 + 1 < 12 && ov_b > 1);
    z.resume_sp = s1 + 1;
    return z;
}

// Continue the reference's do-loop from `scale_power` (value before the ++), precomputed in*2048 arrays.
VGA_HD PassOut resume_passes_core(const int (&x)[16], const int (&in2048v)[14], const int (&in2048p)[14], int c0,
                                  int c1, int scale_power, int &final_sp)
{
    PassOut r;
    bool at_max;
    do {
        scale_power++;
        at_max = scale_power >= 12;
        r = pass_fast_core(x, in2048v, in2048p, c0, c1, scale_power);
        if (!r.exact) r = pass_literal(x, c0, c1, scale_power);
        scale_power = apply_bumps(scale_power, r.max_overflow);
    } while (scale_power < 12 && r.max_overflow > 1 && !at_max);
    final_sp = scale_power;     // the reference's `out scalePower` (== the pass scale on a regular exit)
    return r;
}

// Continue the reference's do-loop from `scale_power` (value before the ++).
// Includes the termination guard documented in gcadpcm_kernels.hip / oracle.
VGA_HD PassOut resume_passes(const int (&x)[16], int c0, int c1, int scale_power, int &final_sp)
{
    PassOut r;
    bool at_max;
    do {
        scale_power++;
        at_max = scale_power >= 12;
        r = pass_fast(x, c0, c1, scale_power);
        if (!r.exact) r = pass_literal(x, c0, c1, scale_power);
        scale_power = apply_bumps(scale_power, r.max_overflow);
    } while (scale_power < 12 && r.max_overflow > 1 && !at_max);
    final_sp = scale_power;     // the reference's `out scalePower` (== the pass scale on a regular exit)
    return r;
}

}  // namespace gc
}  // namespace vga
