// hca_decode_core.hpp -- per-lane logic of the CRI HCA decoder, shared by the gfx950 kernels
// (hca_decode_kernels.hip) and the host-side lane emulator of the CPU test-suite
// (tests/host/hca_decode_emulator.cpp).  No codec result is produced on the CPU in the product path; the host
// build of this header exists for testing only.
//
// Reference: VGAudio/Codecs/CriHca/CriHcaPacking.cs:71-183 (UnpackFrameHeader, ReadScaleFactors, DeltaDecode,
// ReadSpectralCoefficients), Utilities/BitReader.cs:51-118 (PeekInt: MSB first, bits past the end of the frame read as
// 0), CriHcaDecoder.cs:83-192 (DequantizeFrame, RestoreMissingBands, RunImdct, PcmFloatToShort),
// Utilities/Mdct.cs:94-181 (RunImdct, Dct4).
//
// The decoder is two launches (LABNOTES.md 4.4):
//   scan   : lane = frame.  The frame is variable-length coded, so WHERE a symbol starts is a serial walk -- but only its
//            LENGTH is needed for that.  The scan reads the frame header (scale factors, intensity / HFR scales) and
//            then walks the 8 x nch x count spectral codes length-only, noting the bit offset of every 16th symbol.
//            Hand-over per frame: scale factors + those offsets (576 bytes for a stereo frame of 128 bands).
//   frames : workgroup = a run of consecutive frames of one stream, 128 threads.  The frame's bytes sit in LDS; every
//            16-symbol chunk is decoded by its own lane (values, dequantised straight into the transform's input
//            layout), the 128-point DCT-IV runs on 8 lanes per transform with the twiddles in registers, window +
//            overlap-add + PCM16 follow; the overlap (`_imdctPrevious`) is carried from frame to frame inside the run.
#pragma once
#include <cstdint>

#include "hca_info.hpp"

#ifndef VGA_HD
#if defined(__HIPCC__)
#define VGA_HD __host__ __device__ __forceinline__
#else
#define VGA_HD inline
#endif
#endif

namespace vga {
namespace hca {

constexpr int CHUNK_SYMBOLS = 16;      // spectral codes per decode chunk (one lane of the frames kernel)
constexpr int REC_CHANNEL_BYTES = 144; // scale factors[128], intensity[8], hfr scales[8]
constexpr int ROW_BYTES = 1152;        // one transform's LDS row: 64 complex values of 16 bytes + 16 bytes per 8 of them

// What both launches need to know about a stream's frames beyond DeviceInfo (derived on the host, decode_layout()).
struct DecodeLayout {
    int chunks_per_subframe;           // sum of chunk_count[]
    int chunk_base[8], chunk_count[8]; // per channel: first chunk inside a sub-frame, ceil(coded_count / 16)
    int record_bytes;                  // per frame, a multiple of 64
    int offsets_at;                    // byte offset of the chunk offsets inside a record (after the channels)
    int header_at;                     // byte offset of the 16-byte header piece (after the offsets)
    int wide_offsets;                  // 1: uint32 offsets (frames too long for 16 bits)
    int frame_dwords;                  // ceil(frame_size / 4)
};

inline DecodeLayout make_decode_layout(const DeviceInfo &info)
{
    DecodeLayout L{};
    int symbols = 0;
    for (int c = 0; c < info.nch; c++) {
        L.chunk_base[c] = L.chunks_per_subframe;
        L.chunk_count[c] = (info.coded_count[c] + CHUNK_SYMBOLS - 1) / CHUNK_SYMBOLS;
        L.chunks_per_subframe += L.chunk_count[c];
        symbols += info.coded_count[c];
    }
    // the furthest a (possibly corrupt) frame can move the position: every code at most 12 bits, the header at most
    // 35 + nch * (3 + 128 * 11 + 48) bits, on top of reading past the frame's end
    const int64_t max_pos = (int64_t)info.frame_size * 8 + (int64_t)SUBFRAMES * symbols * 12 + 35 + (int64_t)info.nch * 1500;
    L.wide_offsets = max_pos >= 65536 ? 1 : 0;
    L.offsets_at = info.nch * REC_CHANNEL_BYTES;
    const int per_piece = L.wide_offsets ? 4 : 8;
    const int offset_pieces = (SUBFRAMES * L.chunks_per_subframe + per_piece - 1) / per_piece;
    L.header_at = L.offsets_at + 16 * offset_pieces;
    L.record_bytes = (L.header_at + 16 + 63) / 64 * 64;
    L.frame_dwords = (info.frame_size + 3) / 4;
    return L;
}

// ---------------------------------------------------------------------------------------------------------------
// One entry per resolution 0..15 (CriHcaTables: QuantizedSpectrumMaxBits / Bits / Value; CriHcaPacking.cs:154-174):
//   len: bits [3c, 3c+3) = code length of code c (resolutions 1..7), bits [48, 52) = max bits, bit 52 = "large"
//        (resolution >= 8: sign-magnitude code of max bits, one bit shorter when the magnitude is 0);
//   val: bits [4c, 4c+4) = value of code c as a signed nibble.
struct Symbol {
    uint64_t len, val;
};

VGA_HD Symbol make_symbol(int resolution, const uint8_t *bits16, const int8_t *value16, int max_bits)
{
    Symbol s{0, 0};
    if (resolution < 8) {
        for (int c = 0; c < 16; c++) {
            s.len |= (uint64_t)(bits16[c] & 7) << (3 * c);
            s.val |= (uint64_t)(value16[c] & 15) << (4 * c);
        }
    } else {
        s.len |= (uint64_t)1 << 52;
    }
    s.len |= (uint64_t)(max_bits & 15) << 48;
    return s;
}

VGA_HD int symbol_max_bits(uint64_t len) { return (int)(len >> 48) & 15; }
VGA_HD bool symbol_large(uint64_t len) { return ((len >> 52) & 1) != 0; }

// the next `max_bits` bits of a left-aligned window (PeekInt); 0 bits -> 0
VGA_HD uint32_t peek_code(uint32_t w0, int max_bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(w0, (uint32_t)(32 - max_bits), (uint32_t)max_bits);   // width 0 -> 0
#else
    return max_bits ? w0 >> (32 - max_bits) : 0u;
#endif
}

VGA_HD int code_length(uint64_t len, uint32_t code)
{
    const int small = (int)(len >> (3 * (code & 15))) & 7;
    const int large = symbol_max_bits(len) - ((code >> 1) == 0 ? 1 : 0);
    return symbol_large(len) ? large : small;
}

VGA_HD int code_value(const Symbol &s, uint32_t code)
{
    const int nib = (int)(s.val >> (4 * (code & 15))) & 15;
    const int small = (nib ^ 8) - 8;
    const int mag = (int)(code >> 1);
    const int large = (code & 1) ? -mag : mag;         // code / 2 * (1 - (code % 2 * 2))
    return symbol_large(s.len) ? large : small;
}

// ---------------------------------------------------------------------------------------------------------------
// 128 bits of bitstream, left-aligned: the MSB of w0 is the next unread bit.  A symbol is at most 12 bits, so eight of
// them (96 bits) never run past the quad; it is reloaded every eight symbols ("service point").
struct Quad {
    uint32_t w0, w1, w2, w3;
};

VGA_HD void quad_shift(Quad &q, int n)                // 0 <= n <= 31
{
    const uint64_t a = (((uint64_t)q.w0 << 32) | q.w1) << n;
    const uint64_t b = (((uint64_t)q.w1 << 32) | q.w2) << n;
    const uint64_t c = (((uint64_t)q.w2 << 32) | q.w3) << n;
    q.w0 = (uint32_t)(a >> 32);
    q.w1 = (uint32_t)(b >> 32);
    q.w2 = (uint32_t)(c >> 32);
    q.w3 <<= n;
}

VGA_HD Quad quad_from(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4, int shift)   // 0 <= shift <= 31
{
    Quad q;
    q.w0 = (uint32_t)(((((uint64_t)x0 << 32) | x1) << shift) >> 32);
    q.w1 = (uint32_t)(((((uint64_t)x1 << 32) | x2) << shift) >> 32);
    q.w2 = (uint32_t)(((((uint64_t)x2 << 32) | x3) << shift) >> 32);
    q.w3 = (uint32_t)(((((uint64_t)x3 << 32) | x4) << shift) >> 32);
    return q;
}

VGA_HD uint32_t bswap32(uint32_t v)
{
    return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24);
}

// dword k of a frame whose bits occupy [first_bit, first_bit + frame_bits) of a dword-aligned bit numbering: the bits
// past the frame's end are cleared (BitReader: they read as 0)
VGA_HD uint32_t mask_past_end(uint32_t v, int k, int end_bit)
{
    const int n = end_bit - 32 * k;                    // bits of this dword that belong to the frame
    if (n >= 32) return v;
    if (n <= 0) return 0u;
    return v & ~(0xFFFFFFFFu >> n);
}

// CriHcaPacking.cs:60-69
VGA_HD int resolution_for(const uint8_t *res_curve, int scale_factor, int noise_level)
{
    if (scale_factor == 0) return 0;
    int p = noise_level - 5 * scale_factor / 2 + 2;
    p = p < 0 ? 0 : (p > 58 ? 58 : p);
    return res_curve[p];
}

// ---------------------------------------------------------------------------------------------------------------
// The scan of one frame by one lane.  Policies (device: LDS / global memory of the lane; host: plain arrays):
//   Src : uint32_t word(int64_t k)       dword k (frame-aligned numbering, raw little-endian load) -- any k >= 0
//         void     quad(int64_t k, uint32_t out[4])   dwords k..k+3
//   Ring: void put(int slot, uint32_t v); uint32_t get(int slot)        16 dwords of this lane
//   Res : void put(int c, int word, uint32_t v); uint32_t get(int c, int word)   eight 4-bit resolutions per word
//   Out : void piece(const uint32_t v[4])                                16 bytes of the record, in record order
//   Tab : const Symbol &symbol(int r) / uint64_t symbol_len(int r); const uint8_t *res_curve()
struct ScanParams {
    int nch, frame_bits, first_bit;    // first_bit: bit of the frame's first byte inside its first aligned dword (0, 8, 16, 24)
    int hfr_group_count;
    const int *coded_count;            // [nch]
    const int *channel_type;           // [nch]
    const uint8_t *ath_curve;          // [128]
    int wide_offsets;
};

template <class Src, class Ring>
struct LaneBits {
    Src &src;
    Ring &ring;
    Quad q;
    int pos;                           // bit position in the aligned numbering (first_bit + BitReader.Position)
    int end_bit;                       // first_bit + frame_bits
    int wr;                            // dwords landed in the ring so far (a multiple of 4)
    bool pending;
    uint32_t pend[4];

    VGA_HD LaneBits(Src &s, Ring &r) : src(s), ring(r), q{0, 0, 0, 0}, pos(0), end_bit(0), wr(0), pending(false), pend{0, 0, 0, 0} {}

    VGA_HD void land()
    {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e = 0; e < 4; e++) ring.put((wr + e) & 15, mask_past_end(bswap32(pend[e]), wr + e, end_bit));
        wr += 4;
    }
    VGA_HD void start(int first_bit, int frame_bits)
    {
        pos = first_bit;
        end_bit = first_bit + frame_bits;
        wr = 0;
        src.quad(0, pend);
        land();
        src.quad(4, pend);
        land();
        pending = false;
        service();
    }
    // Between two service points at most 96 bits are consumed.  Invariant: after landing, wr - (pos >> 5) >= 5.
    VGA_HD void service()
    {
        if (pending) land();
        const int d = pos >> 5;
        pending = wr - d <= 8;
        if (pending) src.quad(wr, pend);
        q = quad_from(ring.get(d & 15), ring.get((d + 1) & 15), ring.get((d + 2) & 15), ring.get((d + 3) & 15),
                      ring.get((d + 4) & 15), pos & 31);
    }
    VGA_HD int peek(int bits) const { return (int)peek_code(q.w0, bits); }   // bits <= 16
    VGA_HD void skip(int bits)
    {
        quad_shift(q, bits);
        pos += bits;
    }
    VGA_HD int read(int bits)
    {
        const int v = peek(bits);
        skip(bits);
        return v;
    }
};

// accumulates the chunk offsets into 16-byte pieces (eight uint16 or four uint32 each, in order)
template <class Out>
struct OffsetPacker {
    Out &out;
    uint32_t o[4];
    int count;
    bool wide;
    VGA_HD OffsetPacker(Out &o_, bool w) : out(o_), o{0, 0, 0, 0}, count(0), wide(w) {}
    VGA_HD void push(uint32_t v)
    {
        if (wide) {
            o[0] = o[1]; o[1] = o[2]; o[2] = o[3]; o[3] = v;
            if ((++count & 3) == 0) out.piece(o);
        } else {
            o[0] = (o[0] >> 16) | (o[1] << 16);
            o[1] = (o[1] >> 16) | (o[2] << 16);
            o[2] = (o[2] >> 16) | (o[3] << 16);
            o[3] = (o[3] >> 16) | (v << 16);
            if ((++count & 7) == 0) out.piece(o);
        }
    }
    VGA_HD void finish()
    {
        const int per = wide ? 4 : 8;
        if (count % per == 0) return;
        while (count % per != 0) push(0);
    }
};

// Returns the frame's flags (bit 0: bad sync word, bit 1: scale-factor delta decoding failed -- the reference then keeps
// the previous frame's state, which a frame-parallel decoder cannot reproduce: the caller reports it).
template <class Src, class Ring, class Res, class Out, class Tab>
VGA_HD int scan_frame(const ScanParams &P, Src &src, Ring &ring, Res &res, Out &out, const Tab &tab)
{
    LaneBits<Src, Ring> r(src, ring);
    r.start(P.first_bit, P.frame_bits);
    int flags = 0;
    if (r.read(16) != 0xffff) flags |= 1;
    const int noise_level = r.read(9);
    const int eval_boundary = r.read(7);
    const uint8_t *curve = tab.res_curve();

    for (int c = 0; c < P.nch; c++) {
        const int count = P.coded_count[c];
        // ReadScaleFactors / DeltaDecode (CriHcaPacking.cs:111-130, :185-211)
        const int delta_bits = r.read(3);
        const int max_delta = delta_bits > 0 ? 1 << (delta_bits - 1) : 0;
        int prev = 0;
        bool failed = false;
        for (int blk = 0; blk < 8; blk++) {
            uint32_t sfp[4] = {0, 0, 0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
            for (int half = 0; half < 2; half++) {
                r.service();
                uint32_t resw = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int e = 0; e < 8; e++) {
                    const int i = blk * 16 + half * 8 + e;
                    int sf = 0;
                    if (i < count && delta_bits != 0) {
                        if (delta_bits >= 6 || i == 0) {
                            sf = r.read(6);
                        } else if (!failed) {
                            const int delta = r.read(delta_bits) - (max_delta - 1);   // ReadOffsetBinary, positive bias
                            if (delta < max_delta) {
                                sf = prev + delta;
                                if (sf < 0 || sf > 63) { failed = true; sf = 0; }
                            } else {
                                sf = r.read(6);
                            }
                        }
                        prev = sf;
                    }
                    // delta_bits == 0: Array.Clear of ALL 128 scale factors (:114-118); bands >= count stay 0
                    int rs = 0;
                    if (i < count) rs = resolution_for(curve, sf, P.ath_curve[i] + noise_level - (i < eval_boundary ? 1 : 0));
                    resw |= (uint32_t)rs << (4 * e);
                    const int j = half * 8 + e;
                    sfp[j >> 2] |= (uint32_t)sf << (8 * (j & 3));
                }
                res.put(c, blk * 2 + half, resw);
            }
            out.piece(sfp);
        }
        if (failed) flags |= 2;
        uint32_t aux[4] = {0, 0, 0, 0};
        r.service();
        if (P.channel_type[c] == 2 /* StereoSecondary */) {
            for (int i = 0; i < 8; i++) aux[i >> 2] |= (uint32_t)r.read(4) << (8 * (i & 3));
        } else if (P.hfr_group_count > 0) {
            for (int i = 0; i < P.hfr_group_count; i++) aux[2 + (i >> 2)] |= (uint32_t)r.read(6) << (8 * (i & 3));
        }
        out.piece(aux);
    }

    // ReadSpectralCoefficients (:148-183), lengths only
    OffsetPacker<Out> offsets(out, P.wide_offsets != 0);
    for (int sf = 0; sf < 8; sf++) {
        for (int c = 0; c < P.nch; c++) {
            const int count = P.coded_count[c];
            for (int s = 0; s < count; s += 8) {
                r.service();
                if ((s & 15) == 0) offsets.push((uint32_t)(r.pos - P.first_bit));
                const uint32_t resw = res.get(c, s >> 3);
                if (s + 8 <= count) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                    for (int e = 0; e < 8; e++) {
                        const uint64_t len = tab.symbol_len((int)(resw >> (4 * e)) & 15);
                        const uint32_t code = peek_code(r.q.w0, symbol_max_bits(len));
                        r.skip(code_length(len, code));
                    }
                } else {
                    for (int e = 0; s + e < count; e++) {
                        const uint64_t len = tab.symbol_len((int)(resw >> (4 * e)) & 15);
                        const uint32_t code = peek_code(r.q.w0, symbol_max_bits(len));
                        r.skip(code_length(len, code));
                    }
                }
            }
        }
    }
    offsets.finish();
    uint32_t head[4] = {(uint32_t)noise_level | ((uint32_t)eval_boundary << 16) | ((uint32_t)flags << 24), 0, 0, 0};
    out.piece(head);
    return flags;
}

// ---------------------------------------------------------------------------------------------------------------
// Frames kernel, stage A: one 16-symbol chunk.  The values are dequantised (CriHcaDecoder.cs:83-100) and stored as the
// transform's input: the pre-rotation of Dct4 (Mdct.cs:137-147) pairs input[2i] with input[127 - 2i], so band s goes to
// complex slot i = s / 2 (even s, first double) or (127 - s) / 2 (odd s, second double); slot i sits at byte
// 16 i + 16 (i >> 3) of the row (16 bytes of padding per 8 slots: both the lane-per-chunk stores and the two access
// patterns of the transform are then free of LDS bank conflicts).
VGA_HD int spec_byte_offset(int s)
{
    const int i = (s & 1) ? (127 - s) >> 1 : s >> 1;
    return 16 * i + 16 * (i >> 3) + 8 * (s & 1);
}
VGA_HD int slot_byte_offset(int i) { return 16 * i + 16 * (i >> 3); }

// FB: uint32_t get(int k) -- big-endian dword k of the frame, 0 past its end (k is clamped by the callee)
// res16: the chunk's 16 resolutions (res16[e]); gain16: its 16 gains; nsym valid symbols (the rest of the chunk is 0.0).
template <class FB, class R16, class Tab>
VGA_HD void decode_chunk(const FB &fb, int bit_offset, int nsym, int s0, const R16 &res16, const double *gain16,
                         const Tab &tab, char *row)
{
    int pos = bit_offset;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int half = 0; half < 2; half++) {
        const int d = pos >> 5;
        Quad q = quad_from(fb.get(d), fb.get(d + 1), fb.get(d + 2), fb.get(d + 3), fb.get(d + 4), pos & 31);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int e8 = 0; e8 < 8; e8++) {
            const int e = half * 8 + e8;
            double v = 0.0;
            if (e < nsym) {
                const Symbol sym = tab.symbol(res16[e]);
                const uint32_t code = peek_code(q.w0, symbol_max_bits(sym.len));
                const int len = code_length(sym.len, code);
                v = (double)code_value(sym, code) * gain16[e];
                quad_shift(q, len);
                pos += len;
            }
            *reinterpret_cast<double *>(row + spec_byte_offset(s0 + e)) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Stage B: Dct4 (Mdct.cs:126-181) of one row on 8 lanes, L = 0..7.  The transform is six radix-2 stages over 64 complex
// values z[0..63] after a pre-rotation.  Lane L first owns z[L + 8k], k = 0..7: the pairs of stages 0-2 (distances 32,
// 16, 8) are then inside a lane; after ONE exchange through the row (write z[L + 8k], read z[8L + m]) the pairs of stages
// 3-5 (distances 4, 2, 1) are, and their twiddles no longer depend on the lane.  Every butterfly is the reference's,
// operand for operand (no FMA contraction), so the outputs are bit-identical; only the order in which independent
// butterflies are executed differs.
struct Twiddle {
    double s, c;
};
struct DctLane {                       // loop-invariant per lane: 15 twiddles + two store bases
    Twiddle pre[8], st0[4], st1[2], st2;
    int out_even, out_odd;             // byte offsets of this lane's outputs with even / odd code parity (see dct_store)
};
struct DctUniform {                    // the same for every lane
    Twiddle st3[4], st4[2], st5;
};

// sin_bits / cos_bits: the concatenated tables of sizes 1, 2, .. 128 (size 2^b starts at 2^b - 1)
VGA_HD double bits_to_double(uint64_t b)
{
    union { uint64_t u; double d; } x;
    x.u = b;
    return x.d;
}
VGA_HD Twiddle twiddle_at(const uint64_t *sin_bits, const uint64_t *cos_bits, int size, int i)
{
    return Twiddle{bits_to_double(sin_bits[size - 1 + i]), bits_to_double(cos_bits[size - 1 + i])};
}
VGA_HD DctLane make_dct_lane(const uint64_t *sin_bits, const uint64_t *cos_bits, int L)
{
    DctLane K;
    for (int k = 0; k < 8; k++) K.pre[k] = twiddle_at(sin_bits, cos_bits, 128, L + 8 * k);   // sinTable[i], i < 64 (Mdct.cs:143)
    for (int k = 0; k < 4; k++) K.st0[k] = twiddle_at(sin_bits, cos_bits, 32, L + 8 * k);
    for (int k = 0; k < 2; k++) K.st1[k] = twiddle_at(sin_bits, cos_bits, 16, L + 8 * k);
    K.st2 = twiddle_at(sin_bits, cos_bits, 8, L);
    // output index of dctTemp[16 L + j]: i = invgray(bitreverse7(16 L + j)) (the inverse of Mdct.cs:196-207), i.e.
    // 8 u(j) + (v(L) ^ (7 * parity(j))) with v(L) = invgray3(bitreverse3(L))
    const int rev = ((L & 1) << 2) | (L & 2) | ((L >> 2) & 1);
    const int v = rev ^ (rev >> 1) ^ (rev >> 2);
    K.out_even = 8 * v;
    K.out_odd = 8 * (v ^ 7);
    return K;
}
VGA_HD DctUniform make_dct_uniform(const uint64_t *sin_bits, const uint64_t *cos_bits)
{
    DctUniform U;
    for (int m = 0; m < 4; m++) U.st3[m] = twiddle_at(sin_bits, cos_bits, 4, m);
    for (int m = 0; m < 2; m++) U.st4[m] = twiddle_at(sin_bits, cos_bits, 2, m);
    U.st5 = twiddle_at(sin_bits, cos_bits, 1, 0);
    return U;
}

struct Cx {
    double re, im;
};
// Mdct.cs:163-172: front += back; back = (front - back) rotated
VGA_HD void butterfly(Cx &f, Cx &b, const Twiddle &t)
{
    const double a = f.re - b.re;
    const double d = f.im - b.im;
    f.re = f.re + b.re;
    f.im = f.im + b.im;
    b.re = a * t.c + d * t.s;
    b.im = a * t.s - d * t.c;
}

// pre-rotation + stages 0..2 in place: reads slot L + 8k, writes z[L + 8k] back to the same slot
VGA_HD void dct_first_half(char *row, int L, const DctLane &K)
{
    Cx z[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) {
        const double *p = reinterpret_cast<const double *>(row + slot_byte_offset(L + 8 * k));
        const double a = p[0], b = p[1];
        z[k].re = a * K.pre[k].c + b * K.pre[k].s;      // Mdct.cs:145-146
        z[k].im = a * K.pre[k].s - b * K.pre[k].c;
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) butterfly(z[k], z[k + 4], K.st0[k]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 2; k++) {
        butterfly(z[k], z[k + 2], K.st1[k]);
        butterfly(z[k + 4], z[k + 6], K.st1[k]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k += 2) butterfly(z[k], z[k + 1], K.st2);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) {
        double *p = reinterpret_cast<double *>(row + slot_byte_offset(L + 8 * k));
        p[0] = z[k].re;
        p[1] = z[k].im;
    }
}

// The same, with the lane's twiddles fetched stage by stage from the tables (L1 / constant-cache hits) instead of held
// in registers: a kernel whose other stages need the registers (the encoder) pays 15 small loads per transform for
// 60 VGPRs.  The fences keep the compiler from hoisting every load to the top, which would bring the 60 back.
VGA_HD void stage_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
}
VGA_HD void dct_first_half_streamed(char *row, int L, const uint64_t *sin_bits, const uint64_t *cos_bits)
{
    Cx z[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) {
        const double *p = reinterpret_cast<const double *>(row + slot_byte_offset(L + 8 * k));
        const double a = p[0], b = p[1];
        const Twiddle t = twiddle_at(sin_bits, cos_bits, 128, L + 8 * k);
        z[k].re = a * t.c + b * t.s;
        z[k].im = a * t.s - b * t.c;
    }
    stage_fence();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 4; k++) butterfly(z[k], z[k + 4], twiddle_at(sin_bits, cos_bits, 32, L + 8 * k));
    stage_fence();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 2; k++) {
        const Twiddle t = twiddle_at(sin_bits, cos_bits, 16, L + 8 * k);
        butterfly(z[k], z[k + 2], t);
        butterfly(z[k + 4], z[k + 6], t);
    }
    stage_fence();
    {
        const Twiddle t = twiddle_at(sin_bits, cos_bits, 8, L);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < 8; k += 2) butterfly(z[k], z[k + 1], t);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k++) {
        double *p = reinterpret_cast<double *>(row + slot_byte_offset(L + 8 * k));
        p[0] = z[k].re;
        p[1] = z[k].im;
    }
}

// stages 3..5 on z[8L + m]; y[2m], y[2m + 1] = dctTemp[16L + 2m], [16L + 2m + 1]
VGA_HD void dct_second_half(const char *row, int L, const DctUniform &U, double y[16])
{
    Cx z[8];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 8; m++) {
        const double *p = reinterpret_cast<const double *>(row + slot_byte_offset(8 * L + m));
        z[m].re = p[0];
        z[m].im = p[1];
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 4; m++) butterfly(z[m], z[m + 4], U.st3[m]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 2; m++) {
        butterfly(z[m], z[m + 2], U.st4[m]);
        butterfly(z[m + 4], z[m + 6], U.st4[m]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 8; m += 2) butterfly(z[m], z[m + 1], U.st5);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int m = 0; m < 8; m++) {
        y[2 * m] = z[m].re;
        y[2 * m + 1] = z[m].im;
    }
}

// output[i] = dctTemp[shuffle[i]] * Scale (Mdct.cs:177-180), as a scatter: y[j] = dctTemp[16 L + j] goes to output index
// 8 u(j) + (v(L) ^ 7 parity(j)); `out` is a plain array of 128 doubles (which may be the row itself: the eight lanes of
// a transform are in one wave and have all executed dct_second_half's reads before any of these stores).
VGA_HD constexpr int out_block_of(int j)               // u(j) = invgray4(bitreverse4(j))
{
    const int r = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
    return r ^ (r >> 1) ^ (r >> 2) ^ (r >> 3);
}
VGA_HD constexpr int parity4(int j) { return (j ^ (j >> 1) ^ (j >> 2) ^ (j >> 3)) & 1; }

VGA_HD void dct_store(char *out, const DctLane &K, const double y[16])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; j++) {
        const int base = parity4(j) ? K.out_odd : K.out_even;
        *reinterpret_cast<double *>(out + 64 * out_block_of(j) + base) = y[j] * 0.125;   // Scale = sqrt(2 / 128)
    }
}

// Stage C: one output sample of RunImdct (Mdct.cs:112-118) + PcmFloatToShort (CriHcaDecoder.cs:179-192).
// Sample j of a sub-frame needs dctOut[imdct_cur_index(j)] of this sub-frame, dctOut[imdct_prev_index(j)] of the one
// before (the previous frame's last one; zeros at the start of the stream: window * -0.0 + x == x and
// x - window * 0.0 == x, so zeros reproduce the cleared _imdctPrevious), window[j] and window[127 - j].
VGA_HD int imdct_cur_index(int j) { return j < 64 ? j + 64 : 191 - j; }
VGA_HD int imdct_prev_index(int j) { return j < 64 ? 63 - j : j - 64; }

VGA_HD int imdct_sample(bool lower_half, double w_cur, double w_prev, double cur, double prev)
{
    double out;
    if (lower_half) {
        const double p = w_prev * -prev;                           // _imdctPrevious[j] = window[127 - j] * -dctOut[63 - j]
        out = w_cur * cur + p;
    } else {
        const double p = w_prev * prev;                            // _imdctPrevious[j] = window[127 - j] * dctOut[j - 64]
        out = w_cur * -cur - p;
    }
    const double scaled = out * 32768.0;
    // (int)x in RyuJIT (cvttsd2si): out of range or NaN -> 0x80000000, then Clamp16
    int sample = (scaled > -2147483649.0 && scaled < 2147483648.0) ? (int)scaled : (int)0x80000000;
    sample = sample < -32768 ? -32768 : (sample > 32767 ? 32767 : sample);
    return sample;
}

}  // namespace hca
}  // namespace vga
