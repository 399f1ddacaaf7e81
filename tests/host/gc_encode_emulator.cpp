// gc_encode_emulator.cpp -- host-side lane emulator of gc_encode_kernel's per-frame control
// flow (16 lanes per channel: 8 predictors x 2 speculative scale candidates), built from
// the SAME header the kernel uses (vgaudio_amd/csrc/gc_encode_core.hpp).  TEST ONLY: lets the
// CPU suite check the exact-arithmetic shortcuts and the candidate-resolution state machine
// against the oracle without a GPU.  Compiled by tests/test_host_gc_encode_core.py.
#include "../../vgaudio_amd/csrc/gc_encode_core.hpp"
#include <cstring>

using namespace vga::gc;

extern "C" {

// exhaustive check of halvings() against the reference's while loop (GcAdpcmEncoder.cs:119-123)
int emu_check_halvings(void)
{
    for (int md0 = -32768; md0 <= 32767; md0++) {
        int md = md0, sp = 0;
        while (sp <= 12 && (md > 7 || md < -8)) { md /= 2; sp++; }
        if (sp != halvings(md0)) return md0 == 0 ? 1 : md0;
    }
    return 0;
}

// compare pass_fast against pass_literal on one frame; returns 0 ok, 1 mismatch, 2 fast not exact
int emu_compare_pass(const int16_t *x16, int c0, int c1, int scale_power)
{
    int x[16];
    for (int i = 0; i < 16; i++) x[i] = x16[i];
    const PassOut f = pass_fast(x, c0, c1, scale_power);
    if (!f.exact) return 2;
    const PassOut l = pass_literal(x, c0, c1, scale_power);
    bool same = f.total == l.total && f.max_overflow == l.max_overflow && f.o12 == l.o12 && f.o13 == l.o13;
    for (int s = 0; s < 14; s++) same = same && f.q[s] == l.q[s];
    uint32_t fa, fb, la, lb;
    frame_words(f, 3, scale_power, fa, fb);
    frame_words(l, 3, scale_power, la, lb);
    same = same && fa == la && fb == lb;
    return same ? 0 : 1;
}

// the pass without the f32 detour (NO_ROUND) against the fast pass: equal in every field whenever pass_no_round_is_exact says so.
// Returns 0 equal (trusted), 1 MISMATCH although trusted, 2 not trusted (the kernel runs the exact pass then)
int emu_compare_pass_no_round(const int16_t *x16, int c0, int c1, int scale_power)
{
    int x[16], m[14], mp[14];
    for (int i = 0; i < 16; i++) x[i] = x16[i];
    for (int s = 0; s < 14; s++) { m[s] = x[s + 2] * 2048; mp[s] = m[s] + 1024; }
    const PassOut n = pass_fast_core_no_round(x, m, mp, c0, c1, scale_power);
    if (!pass_no_round_is_exact(scale_power, n.max_overflow)) return 2;
    const PassOut f = pass_fast_core(x, m, mp, c0, c1, scale_power);
    bool same = (uint32_t)f.total == (uint32_t)n.total && f.max_overflow == n.max_overflow && f.hist_pair == n.hist_pair;
    for (int s = 0; s < 14; s++) same = same && f.q[s] == n.q[s];
    return same ? 0 : 1;
}

// the 64-bit-sum form of the fast pass (the kernel's cold block, round 5) against the literal pass: whatever the overflow,
// as long as the coefficients cannot wrap int32.  Returns 0 ok, 1 mismatch, 2 not exact (coefficients too large)
int emu_compare_pass_wide(const int16_t *x16, int c0, int c1, int scale_power)
{
    int x[16], m[14], mp[14];
    for (int i = 0; i < 16; i++) x[i] = x16[i];
    for (int s = 0; s < 14; s++) { m[s] = x[s + 2] * 2048; mp[s] = m[s] + 1024; }
    const PassOut f = pass_fast_core_wide(x, m, mp, c0, c1, scale_power);
    if (!f.exact) return 2;
    const PassOut l = pass_literal(x, c0, c1, scale_power);
    bool same = f.total == l.total && f.max_overflow == l.max_overflow && f.o12 == l.o12 && f.o13 == l.o13;
    for (int s = 0; s < 14; s++) same = same && f.q[s] == l.q[s];
    return same ? 0 : 1;
}

// stats[0] frames, [1] lanes whose fast pass was inexact, [2] pairs resolved by A, [3] by B,
// [4] pairs needing resume, [5] pre-scan ties, [6] frames taking the 64-bit argmin path
int emu_encode(const int16_t *pcm, int sample_count, const int16_t *coefs, int16_t hist1, int16_t hist2,
               uint8_t *out, uint64_t *stats)
{
    int x[16];
    x[0] = hist2;
    x[1] = hist1;
    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    const int frames = full_frames + (tail ? 1 : 0);
    for (int f = 0; f < frames; f++) {
        for (int s = 0; s < 14; s++) {
            const int idx = f * 14 + s;
            x[2 + s] = idx < sample_count ? pcm[idx] : 0;
        }
        stats[0]++;
        uint64_t best_key = ~0ull;
        int best_p = 0, best_sp = 0;
        PassOut best = {};
        bool wide = false;
        for (int p = 0; p < 8; p++) {
            const int c0 = coefs[2 * p], c1 = coefs[2 * p + 1];
            // pre-scan split over the two candidate lanes (samples 0..6 / 7..13), then combined
            int dmax_a = 0, dmin_a = 0, dmax_b = 0, dmin_b = 0;
            prescan_range(x, c0, c1, 0, 7, dmax_a, dmin_a);
            prescan_range(x, c0, c1, 7, 14, dmax_b, dmin_b);
            int s1 = first_scale_power_from_range(imax(dmax_a, dmax_b), imin(dmin_a, dmin_b));
            if (s1 == -100) {
                stats[5]++;
                s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
            }
            PassOut cand[2];
            for (int c = 0; c < 2; c++) {
                const int sp = imin(s1 + c, 12);
                cand[c] = pass_fast(x, c0, c1, sp);
                if (!cand[c].exact) { stats[1]++; cand[c] = pass_literal(x, c0, c1, sp); }
            }
            const bool bump = imax(cand[0].max_overflow, cand[1].max_overflow) > 248;
            const Resolve z = bump ? resolve_candidates(s1, cand[0].max_overflow, cand[1].max_overflow)
                                   : resolve_candidates_nobump(s1, cand[0].max_overflow, cand[1].max_overflow);
            PassOut fin;
            int fin_sp;
            if (z.final_a) { fin = cand[0]; fin_sp = s1; stats[2]++; }
            else if (z.final_b) { fin = cand[1]; fin_sp = s1 + 1; stats[3]++; }
            else { fin = resume_passes(x, c0, c1, z.resume_sp, fin_sp); stats[4]++; }
            if (fin.total >= (1ull << 28)) wide = true;
            const uint64_t key = (fin.total << 4) | (uint64_t)(p << 1);
            if (key < best_key) { best_key = key; best_p = p; best = fin; best_sp = fin_sp; }
        }
        if (wide) stats[6]++;
        uint8_t frame[8];
        uint32_t d0, d1;
        frame_words(best, best_p, best_sp, d0, d1);
        memcpy(frame, &d0, 4);
        memcpy(frame + 4, &d1, 4);
        const int nbytes = f < full_frames ? 8 : (tail + 2 + 1) / 2;
        memcpy(out + (size_t)f * 8, frame, (size_t)nbytes);
        x[0] = best.o12;
        x[1] = best.o13;
    }
    return 0;
}

// The (channel, predictor) layout's frame as the kernel resolves it since round 5 (encode_frame8 + encode_frame_cold in
// gc_encode_kernel.hip): both candidate passes, then per lane -- the bump loop entered (the reference's loop as written from
// the scale it moves to), hostile coefficients (the whole loop), a final pass at the cap that overflowed by more than 3 (same
// pass, 64-bit sum), third trips -- and the argmin over keys that saturate at 2^28 with the 64-bit keys as fall-back.
// stats: [0] frames [1] generic lanes [2] inexact-sum lanes [3] resume lanes [4] frames whose best key saturated
//        [5] frames encoded with the passes that skip the f32 detour
int emu_encode8(const int16_t *pcm, int sample_count, const int16_t *coefs, int16_t hist1, int16_t hist2, uint8_t *out, uint64_t *stats)
{
    int x[16];
    x[0] = hist2;
    x[1] = hist1;
    const int full_frames = sample_count / 14;
    const int tail = sample_count - full_frames * 14;
    const int frames = full_frames + (tail ? 1 : 0);
    for (int f = 0; f < frames; f++) {
        for (int s = 0; s < 14; s++) {
            const int idx = f * 14 + s;
            x[2 + s] = idx < sample_count ? pcm[idx] : 0;
        }
        stats[0]++;
        int m[14], mp[14];
        for (int s = 0; s < 14; s++) { m[s] = x[s + 2] * 2048; mp[s] = m[s] + 1024; }
        PassOut fin[8];
        int fin_sp[8];
        // the passes without the f32 detour when every lane (here: every predictor of this channel; on the GPU eight channels
        // share the decision) quantises at scale 9 or below and every lane can vouch for them from its overflow
        int s1_of[8];
        bool short_passes = true;
        for (int p = 0; p < 8; p++) {
            const int c0 = coefs[2 * p], c1 = coefs[2 * p + 1];
            int dmax = 0, dmin = 0;
            prescan_range(x, c0, c1, 0, 14, dmax, dmin);
            int s1 = first_scale_power_from_range(dmax, dmin);
            if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
            s1_of[p] = s1;
            if (imin(s1 + 1, 12) > 9) short_passes = false;
        }
        if (short_passes)
            for (int p = 0; p < 8; p++) {
                const int c0 = coefs[2 * p], c1 = coefs[2 * p + 1];
                const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;
                const int sp_a = imin(s1_of[p], 12), sp_b = imin(s1_of[p] + 1, 12);
                const PassOut nb = pass_fast_core_no_round(x, m, mp, c0, c1, sp_b), na = pass_fast_core_no_round(x, m, mp, c0, c1, sp_a);
                if (coef_ok && !(pass_no_round_is_exact(sp_a, na.max_overflow) && pass_no_round_is_exact(sp_b, nb.max_overflow))) short_passes = false;
            }
        if (short_passes) stats[5]++;
        for (int p = 0; p < 8; p++) {
            const int c0 = coefs[2 * p], c1 = coefs[2 * p + 1];
            const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;
            const int s1 = s1_of[p];
            const int sp_a = imin(s1, 12), sp_b = imin(s1 + 1, 12);
            const PassOut rb = short_passes ? pass_fast_core_no_round(x, m, mp, c0, c1, sp_b) : pass_fast_core(x, m, mp, c0, c1, sp_b);
            const PassOut ra = short_passes ? pass_fast_core_no_round(x, m, mp, c0, c1, sp_a) : pass_fast_core(x, m, mp, c0, c1, sp_a);
            const bool cap_a = sp_a >= 12, cap_b = sp_b >= 12;
            const int eff_a = cap_a ? 0 : ra.max_overflow, eff_b = cap_b ? 0 : rb.max_overflow;
            const bool fin_a = eff_a < 2;
            const bool bump_a = !cap_a && (unsigned)ra.max_overflow > 248u;
            const bool bump_b = !fin_a && !cap_b && (unsigned)rb.max_overflow > 248u;
            const bool generic = !coef_ok || bump_a || bump_b;
            const bool inexact = !generic && (fin_a ? (cap_a && (unsigned)ra.max_overflow > 3u) : (cap_b && (unsigned)rb.max_overflow > 3u));
            const bool resume = !generic && !fin_a && eff_b >= 2;
            PassOut r = fin_a ? ra : rb;
            int fsp = fin_a ? sp_a : sp_b;
            if (generic) {
                const int start = !coef_ok ? s1 - 1 : (bump_a ? apply_bumps(s1, ra.max_overflow) : apply_bumps(s1 + 1, rb.max_overflow));
                r = resume_passes(x, c0, c1, start, fsp);
                stats[1]++;
            }
            if (inexact) {
                const PassOut w = pass_fast_core_wide(x, m, mp, c0, c1, fsp);
                r.total = w.total;
                stats[2]++;
            }
            if (resume) {
                int sp = s1 + 1;
                for (;;) {
                    sp++;
                    bool short_pass = sp <= 9;             // (the kernel: when every lane still in the loop is at scale 9 or below)
                    if (short_pass) {
                        r = pass_fast_core_no_round(x, m, mp, c0, c1, sp);
                        short_pass = pass_no_round_is_exact(sp, r.max_overflow);
                    }
                    if (!short_pass) r = pass_fast_core(x, m, mp, c0, c1, sp);
                    const bool cap = sp >= 12;
                    if ((unsigned)r.max_overflow > (cap ? 3u : 248u)) { r = resume_passes(x, c0, c1, sp - 1, fsp); break; }
                    fsp = sp;
                    if (cap || r.max_overflow <= 1) break;
                }
                stats[3]++;
            }
            fin[p] = r;
            fin_sp[p] = fsp;
        }
        const unsigned SAT = (1u << 28) - 1;
        unsigned best32 = 0xFFFFFFFFu;
        for (int p = 0; p < 8; p++) {
            const unsigned tot = (fin[p].total >> 32) ? SAT : ((unsigned)fin[p].total < SAT ? (unsigned)fin[p].total : SAT);
            const unsigned key = (tot << 3) | (unsigned)p;
            if (key < best32) best32 = key;
        }
        int winner = (int)(best32 & 7u);
        if ((best32 >> 3) >= SAT) {
            stats[4]++;
            uint64_t best = ~0ull;
            for (int p = 0; p < 8; p++) {
                const uint64_t key = (fin[p].total << 3) | (uint64_t)p;
                if (key < best) best = key;
            }
            winner = (int)(best & 7u);
        }
        uint8_t frame[8];
        uint32_t d0, d1;
        frame_words(fin[winner], winner, fin_sp[winner], d0, d1);
        memcpy(frame, &d0, 4);
        memcpy(frame + 4, &d1, 4);
        const int nbytes = f < full_frames ? 8 : (tail + 2 + 1) / 2;
        memcpy(out + (size_t)f * 8, frame, (size_t)nbytes);
        x[0] = fin[winner].o12;
        x[1] = fin[winner].o13;
    }
    return 0;
}

}  // extern "C"
