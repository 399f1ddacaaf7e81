// adx_seam_stats.c -- how many frames an ADX seam takes to close (analysis tool, not part of the product or the tests).
// A channel of the synthetic set is encoded serially (CriAdxCodec.EncodeFrame, version 4, linear scale, 18-byte frames);
// at every piece boundary a second run starts `warm` frames earlier from the guessed history (the two input samples
// before it) and the tool counts the frames past the boundary until both runs hold the same history at a frame end.
//   gcc -O2 -o /tmp/adx_seam_stats tests/host/analysis/adx_seam_stats.c -Ioracle -Loracle -loracle -lm -lpthread
//   LD_LIBRARY_PATH=oracle /tmp/adx_seam_stats [channels] [first] [pieces]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static int clamp16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

// one frame from the history (a = older, b = newer); returns through a, b the history after it
static void encode_frame(const int16_t *x, int *a, int *b, int c0, int c1)
{
    int pa = *a, pb = *b, max_distance = 0;
    for (int j = 0; j < 32; j++) {
        int d = clamp16(x[j] - (((pb * c0) >> 12) + ((pa * c1) >> 12)));
        if (d < 0) d = -d;
        if (d > max_distance) max_distance = d;
        pa = pb;
        pb = x[j];
    }
    int scale = (max_distance - 1) / 7 + 1;
    if (scale > 0x1000) scale = 0x1000;
    const double gain = max_distance == 0 ? 0.0 : 32767.0 / (double)max_distance;
    int h2 = *a, h1 = *b;
    for (int j = 0; j < 32; j++) {
        const int raw = x[j] - (((h1 * c0) >> 12) + ((h2 * c1) >> 12));
        int s = clamp16((int)(raw * gain));
        const int sign = (s > 0) - (s < 0);
        int q = (s + 2340 * sign) / 4681;
        q = q > 7 ? 7 : (q < -8 ? -8 : q);
        const int rec = clamp16(clamp16(scale * q) + ((h1 * c0 + h2 * c1) >> 12));
        h2 = h1;
        h1 = rec;
    }
    *a = h2;
    *b = h1;
}

static int cmp_int(const void *p, const void *q) { return *(const int *)p - *(const int *)q; }

int main(int argc, char **argv)
{
    const int nch = argc > 1 ? atoi(argv[1]) : 256, first = argc > 2 ? atoi(argv[2]) : 0, pieces = argc > 3 ? atoi(argv[3]) : 16;
    const int n = 2880000, frames = n / 32;
    int16_t coefs[2];
    vgo_adx_calculate_coefficients(500, 48000, coefs);
    const int c0 = coefs[0], c1 = coefs[1];
    int16_t *pcm = malloc((size_t)n * 2);
    int *ta = malloc(sizeof(int) * (frames + 1)), *tb = malloc(sizeof(int) * (frames + 1));
    const int warms[] = {0, 32, 128, 512};
    const int nw = 4, seams = pieces - 1;
    int *len = malloc(sizeof(int) * nw * nch * seams);
    for (int c = 0; c < nch; c++) {
        vgo_synth_generate(pcm, n, 1, n, first + c, 1);
        int a = pcm[0], b = pcm[0];                        // version 4: history = the first sample twice
        ta[0] = a; tb[0] = b;
        for (int f = 0; f < frames; f++) {
            encode_frame(pcm + (size_t)f * 32, &a, &b, c0, c1);
            ta[f + 1] = a; tb[f + 1] = b;
        }
        const int piece = (frames + pieces - 1) / pieces;
        for (int w = 0; w < nw; w++)
            for (int k = 1; k < pieces; k++) {
                const int f0 = k * piece, start = f0 - warms[w];
                int ga = pcm[(size_t)start * 32 - 2], gb = pcm[(size_t)start * 32 - 1], f = start, closed = -1;
                for (; f < frames; f++) {
                    if (f >= f0 && ga == ta[f] && gb == tb[f]) { closed = f - f0; break; }
                    encode_frame(pcm + (size_t)f * 32, &ga, &gb, c0, c1);
                }
                len[(w * nch + c) * seams + k - 1] = closed < 0 ? frames - f0 : closed;
            }
    }
    for (int w = 0; w < nw; w++) {
        int *v = len + (size_t)w * nch * seams;
        const int m = nch * seams;
        long sum = 0;
        // per wave of 64 channels a seam lasts as long as its slowest lane
        long wave_sum = 0; int wave_n = 0;
        for (int g = 0; g + 64 <= nch; g += 64)
            for (int k = 0; k < seams; k++) {
                int mx = 0;
                for (int c = g; c < g + 64; c++) if (v[c * seams + k] > mx) mx = v[c * seams + k];
                wave_sum += mx; wave_n++;
            }
        int worst_c = 0, worst = 0;
        for (int i = 0; i < m; i++) { sum += v[i]; if (v[i] > worst) { worst = v[i]; worst_c = first + i / seams; } }
        int *s = malloc(sizeof(int) * m);
        memcpy(s, v, sizeof(int) * m);
        qsort(s, m, sizeof(int), cmp_int);
        printf("warm-up %4d frames: mean %.1f  median %d  90%% %d  99%% %d  max %d (channel %d)  per wave of 64 mean %.1f\n", warms[w],
               (double)sum / m, s[m / 2], s[m * 9 / 10], s[m * 99 / 100], worst, worst_c, wave_n ? (double)wave_sum / wave_n : 0.0);
        free(s);
    }
    // the slowest channels without warm-up
    printf("channels whose slowest seam (no warm-up) takes over 1000 frames:");
    for (int c = 0; c < nch; c++) {
        int mx = 0;
        for (int k = 0; k < seams; k++) if (len[c * seams + k] > mx) mx = len[c * seams + k];
        if (mx > 1000) printf(" %d:%d", first + c, mx);
    }
    printf("\n");
    return 0;
}
