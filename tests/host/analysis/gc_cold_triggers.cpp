// Analysis harness (not a test, not product): which condition sends a frame of gc_encode_kernel's (channel, predictor)
// layout to the cold block?  Mirrors encode_frame8's tests (vgaudio_amd/csrc/gc_encode_kernel.hip) per frame on one channel
// with the kernel's own arithmetic header.  Built and run by gc_cold_triggers.py next to this file.
#include "../../../vgaudio_amd/csrc/gc_encode_core.hpp"
#include <cstring>
using namespace vga::gc;

extern "C" void cold_triggers(const int16_t *pcm, int n, const int16_t *coefs, uint64_t *st /* [16] */)
{
    int x[16];
    x[0] = x[1] = 0;
    const int frames = n / 14;
    for (int f = 0; f < frames; f++) {
        for (int s = 0; s < 14; s++) x[2 + s] = pcm[f * 14 + s];
        st[0]++;
        bool any_rare = false, any_resume = false, any_wide = false, r_coef = false, r_bump_a = false, r_bump_b = false, r_cap_a = false,
             r_cap_b_used = false, r_cap_b_unused = false;
        uint64_t best_key = ~0ull;
        PassOut best = {};
        for (int p = 0; p < 8; p++) {
            const int c0 = coefs[2 * p], c1 = coefs[2 * p + 1];
            const bool coef_ok = (c0 < 0 ? -c0 : c0) + (c1 < 0 ? -c1 : c1) <= 32767;
            int dmax = 0, dmin = 0;
            prescan_range(x, c0, c1, 0, 14, dmax, dmin);
            int s1 = first_scale_power_from_range(dmax, dmin);
            if (s1 == -100) s1 = first_scale_power_from_md(prescan_sequential(x, c0, c1));
            const int sp_a = imin(s1, 12), sp_b = imin(s1 + 1, 12);
            const PassOut rb = pass_fast(x, c0, c1, sp_b), ra = pass_fast(x, c0, c1, sp_a);
            const bool cap_a = sp_a >= 12, cap_b = sp_b >= 12;
            const int eff_a = cap_a ? 0 : ra.max_overflow, eff_b = cap_b ? 0 : rb.max_overflow;
            const bool fin_a = eff_a < 2;
            if (!coef_ok) r_coef = true;
            if (!cap_a && ra.max_overflow > 248) r_bump_a = true;
            if (!cap_b && rb.max_overflow > 248) r_bump_b = true;
            if (cap_a && ra.max_overflow > 3) r_cap_a = true;
            if (cap_b && !cap_a && rb.max_overflow > 3) { if (fin_a) r_cap_b_unused = true; else r_cap_b_used = true; }
            const bool rare = !coef_ok || ra.max_overflow > (cap_a ? 3 : 248) || rb.max_overflow > (cap_b ? 3 : 248);
            const bool resume = !fin_a && eff_b >= 2;
            any_rare |= rare;
            any_resume |= resume;
            int fsp;
            PassOut r = fin_a ? ra : rb;
            if (rare || resume) r = resume_passes(x, c0, c1, rare ? s1 - 1 : s1 + 1, fsp);
            else if ((unsigned)r.total >= (1u << 28)) any_wide = true;
            const uint64_t key = (r.total << 3) | (uint64_t)p;
            if (key < best_key) { best_key = key; best = r; }
        }
        if (any_rare) st[1]++;
        if (any_resume) st[2]++;
        if (any_wide) st[3]++;
        if (any_rare || any_resume || any_wide) st[4]++;
        if (r_coef) st[5]++;
        if (r_bump_a) st[6]++;
        if (r_bump_b) st[7]++;
        if (r_cap_a) st[8]++;
        if (r_cap_b_used) st[9]++;
        if (r_cap_b_unused) st[10]++;
        if ((best_key >> 3) >= (1ull << 28)) st[11]++;           // the WINNER's total is wide
        x[0] = best.o12;
        x[1] = best.o13;
    }
}
