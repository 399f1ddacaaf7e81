/*
 * synth_oracle.c -- the synthetic PCM16 generator of the benchmark, in C (TEST INFRASTRUCTURE ONLY).
 *
 * Not a restatement of anything in the reference: SURVEY.md 8(d) defines the workload (an integer-only, counter-based
 * generator so that host and device produce the same bits), vgaudio_amd/synth.py is its definition and
 * csrc/gcadpcm_kernels.hip:synth_kernel the device form.  This file is the third form, fast enough on the host to
 * generate all 32 768 channels x 2 880 000 samples of BASELINE configs[4] for tests/golden/make_gc_shard_oracle_digests.py
 * (numpy needs 1.5 s per channel, this 0.05 s).  tests/test_oracle_gcadpcm.py holds it to synth.generate.
 */
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

#define SYNTH_SEED 0x5EEDull
#define SYNTH_NFREQ 96

static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int64_t tri(uint32_t phase)
{
    const int64_t q = (int64_t)(phase >> 15);
    return q < 65536 ? q - 32768 : 98303 - q;
}

/* synth.channel_params */
void vgo_synth_channel_params(int c, uint32_t out[4])
{
    uint32_t inc = 4921183u;                                   /* round(55 / 48000 * 2^32) */
    for (int k = 0; k < c % SYNTH_NFREQ; k++) inc = (uint32_t)(((uint64_t)inc * 69433u) >> 16);
    const uint64_t h = splitmix64((SYNTH_SEED << 32) ^ (uint64_t)(int64_t)c);
    out[0] = inc;
    out[1] = (uint32_t)(h >> 32);
    out[2] = 4000u + (uint32_t)((h & 0xFFFFFFFFull) % 20001u);
    out[3] = 2000u + (uint32_t)((h >> 20) & 0x3FFF);
}

/* synth.generate, one channel */
void vgo_synth_channel(int c, int64_t first_sample, int n, int16_t *out)
{
    uint32_t p[4];
    vgo_synth_channel_params(c, p);
    const uint32_t f_inc = p[0], phi = p[1], amp = p[2], lfo = p[3];
    const uint64_t base = ((SYNTH_SEED << 32) ^ (uint64_t)(int64_t)c) * 0x100000001B3ull;
    /* the four noise taps of sample i are those of samples i, i-1, i-2, i-3: keep a window */
    int64_t tap[4];
    for (int d = 1; d < 4; d++)
        tap[d] = (int64_t)(splitmix64(base ^ ((uint64_t)first_sample - (uint64_t)d)) & 4095) - 2048;
    for (int k = 0; k < n; k++) {
        const uint64_t i = (uint64_t)first_sample + (uint64_t)k;
        const uint32_t i32 = (uint32_t)i;
        int64_t s = ((tri(i32 * f_inc) * (int64_t)amp) >> 15) + ((tri(i32 * (3u * f_inc) + phi) * (int64_t)(amp / 3)) >> 15);
        tap[0] = (int64_t)(splitmix64(base ^ i) & 4095) - 2048;
        s += (tap[0] + tap[1] + tap[2] + tap[3]) >> 2;
        tap[3] = tap[2]; tap[2] = tap[1]; tap[1] = tap[0];
        const int64_t env = 20480 + ((tri(i32 * lfo) * 12287) >> 15);
        s = (s * env) >> 15;
        out[k] = (int16_t)(s < -32768 ? -32768 : (s > 32767 ? 32767 : s));
    }
}

typedef struct { int16_t *out; long pitch; int nch, n, first_channel; int next; pthread_mutex_t mu; } synth_job;

static void *synth_worker(void *arg)
{
    synth_job *j = (synth_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        const int k = j->next++;
        pthread_mutex_unlock(&j->mu);
        if (k >= j->nch) return 0;
        vgo_synth_channel(j->first_channel + k, 0, j->n, j->out + (long)k * j->pitch);
    }
}

void vgo_synth_generate(int16_t *out, long pitch, int nch, int n, int first_channel, int threads)
{
    synth_job j;
    memset(&j, 0, sizeof j);
    j.out = out; j.pitch = pitch; j.nch = nch; j.n = n; j.first_channel = first_channel;
    pthread_mutex_init(&j.mu, 0);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    for (int t = 1; t < threads; t++) pthread_create(&th[t], 0, synth_worker, &j);
    synth_worker(&j);
    for (int t = 1; t < threads; t++) pthread_join(th[t], 0);
    pthread_mutex_destroy(&j.mu);
}
