/* Analysis harness (not a test, not product): how often is a predictor that needs a third quantise pass
   (GcAdpcmEncoder.cs:127-170) the frame's winner (:66-76)?  Decides whether deferring third passes and
   batching them over several frames (speculating "a third-pass predictor does not win") can pay.
   Builds on the oracle's restatement by including its source.
   Built and run by third_trip_stats.py next to this file. */
#include "../../../oracle/gcadpcm_oracle.c"

typedef struct {
    uint64_t frames, pair_frames, third_pairs, frames_with_third, third_wins, third_wins_strict;
    uint64_t fourth_pairs, runs_hist[16];
} third_stats;

static int trips_of(const int16_t pcm_in[16], int n, const int16_t c[2], int pcm_out[16], int adpcm[14], int *scale, double *dist)
{
    uint64_t before[16]; memcpy(before, g_trip_hist, sizeof before);
    dsp_encode_coef(pcm_in, n, c, pcm_out, adpcm, scale, dist);
    for (int t = 0; t < 16; t++) if (g_trip_hist[t] != before[t]) return t;
    return 0;
}

void third_trip_stats(const int16_t *pcm, int n_samples, const int16_t coefs_in[16], third_stats *st)
{
    int16_t buf[16] = {0};
    int frames = n_samples / 14;
    for (int f = 0; f < frames; f++) {
        memcpy(buf + 2, pcm + (long)f * 14, 28);
        int pcm_out[8][16], adpcm[8][14], scale[8], trips[8];
        double dist[8];
        int any = 0;
        for (int i = 0; i < 8; i++) {
            int16_t c[2] = { coefs_in[2 * i], coefs_in[2 * i + 1] };
            trips[i] = trips_of(buf, 14, c, pcm_out[i], adpcm[i], &scale[i], &dist[i]);
            st->pair_frames++;
            if (trips[i] >= 3) { st->third_pairs++; any++; }
            if (trips[i] >= 4) st->fourth_pairs++;
        }
        st->frames++;
        st->runs_hist[any]++;
        if (any) st->frames_with_third++;
        int best = 0; double mn = 1.7976931348623157e308;
        for (int i = 0; i < 8; i++) if (dist[i] < mn) { mn = dist[i]; best = i; }
        if (trips[best] >= 3) st->third_wins++;
        buf[0] = (int16_t)pcm_out[best][14];
        buf[1] = (int16_t)pcm_out[best][15];
    }
}

/* How many quantise passes does a (frame, predictor) take?  hist[t] counts pairs that took t passes (t = 1 ..), frame_max[t]
   the frames whose slowest predictor took t: what a pass-level work queue (lanes that take passes instead of frames) would
   have to schedule -- the encoder's waves run passes for all 64 lanes as long as one lane needs another. */
void trip_histogram(const int16_t *pcm, int n_samples, const int16_t coefs_in[16], uint64_t hist[16], uint64_t frame_max[16])
{
    int16_t buf[16] = {0};
    int frames = n_samples / 14;
    for (int f = 0; f < frames; f++) {
        memcpy(buf + 2, pcm + (long)f * 14, 28);
        int pcm_out[8][16], adpcm[8][14], scale[8], mx = 0;
        double dist[8];
        for (int i = 0; i < 8; i++) {
            int16_t c[2] = { coefs_in[2 * i], coefs_in[2 * i + 1] };
            const int t = trips_of(buf, 14, c, pcm_out[i], adpcm[i], &scale[i], &dist[i]);
            hist[t < 15 ? t : 15]++;
            if (t > mx) mx = t;
        }
        frame_max[mx < 15 ? mx : 15]++;
        int best = 0; double mn = 1.7976931348623157e308;
        for (int i = 0; i < 8; i++) if (dist[i] < mn) { mn = dist[i]; best = i; }
        buf[0] = (int16_t)pcm_out[best][14];
        buf[1] = (int16_t)pcm_out[best][15];
    }
}
