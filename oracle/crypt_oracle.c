/* crypt_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): CPU restatement of the reference's ADX and HCA
 * encryption passes and key derivations.
 *
 *   VGAudio/Codecs/CriAdx/CriAdxKey.cs:10-66, CriAdxEncryption.cs:8-108, Utilities/Helpers.cs:115-139 (GetPrimes)
 *   VGAudio/Codecs/CriHca/CriHcaKey.cs:8-181, CriHcaEncryption.cs:12-33
 *
 * The reference has NO tests for these: parity unpinned by reference vectors.  Pinned instead by hand-derived
 * properties (tests/test_oracle_crypt.py): decrypt(encrypt(x)) == x, type-0 table is the identity, tables are
 * permutations with 0 and 0xFF fixed, KeyCode round trip, LCG stepping against a literal loop, CRC refresh. */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ ADX */
static int g_primes[0x400];
static int g_primes_ready;

static void build_primes(void)                                          /* CriAdxKey.BuildPrimesTable (:58-65) */
{
    if (g_primes_ready) return;
    const int max_prime = 0x8000, max = max_prime / 2;                  /* Helpers.GetPrimes (:115-139) */
    static unsigned char sieve[0x4000];
    memset(sieve, 0, sizeof sieve);
    for (int i = 3; i * i < max_prime; i += 2) {
        if (sieve[i >> 1]) continue;
        for (int j = i * i; j < max_prime; j += i * 2) sieve[j >> 1] = 1;
    }
    int n = 0;
    /* primes in ascending order: 2, then 2i+1 for unmarked i; keep the 0x400 that follow 0x4000 */
    for (int i = 1; i < max && n < 0x400; i++)
        if (!sieve[i] && i * 2 + 1 >= 0x4000) g_primes[n++] = i * 2 + 1;
    g_primes_ready = 1;
}

void vgo_adx_key_from_code(uint64_t key_code, vgo_adx_key *k)           /* CriAdxKey(ulong) (:17-23) */
{
    key_code--;
    k->seed = (int)(key_code >> 27 & 0x7fff);
    k->mult = (int)((key_code >> 12 & 0x7ffc) | 1);
    k->inc = (int)((key_code << 1 & 0x7fff) | 1);
}

void vgo_adx_key_from_string(const char *s, vgo_adx_key *k)             /* CriAdxKey(string) (:25-40) */
{
    k->seed = k->mult = k->inc = 0;
    if (!s || !*s) return;
    build_primes();
    k->seed = g_primes[0x100];
    k->mult = g_primes[0x200];
    k->inc = g_primes[0x300];
    for (const unsigned char *c = (const unsigned char *)s; *c; c++) {  /* ASCII key strings: char == byte */
        int p = g_primes[*c + 0x80];
        k->seed = g_primes[k->seed * p % 0x400];
        k->mult = g_primes[k->mult * p % 0x400];
        k->inc = g_primes[k->inc * p % 0x400];
    }
}

uint64_t vgo_adx_key_code(const vgo_adx_key *k)                         /* KeyCode (:48-56) */
{
    uint64_t seed = (uint64_t)k->seed << 27;
    uint64_t mult = (uint64_t)(k->mult & 0xfffc) << 12;
    uint64_t inc = (uint64_t)k->inc >> 1;
    return (seed | mult | inc) + 1;
}

/* EncryptDecryptChannel (CriAdxEncryption.cs:16-41), in place; adpcm_len a whole number of frames */
void vgo_adx_crypt_channel(uint8_t *adpcm, int adpcm_len, const vgo_adx_key *key, int encryption_type, int frame_size,
                           int channel_num, int channel_count)
{
    int xor = key->seed;
    int frame_count = (adpcm_len + frame_size - 1) / frame_size;
    for (int i = 0; i < channel_num; i++) xor = (xor * key->mult + key->inc) & 0x7fff;
    for (int i = 0; i < frame_count; i++) {
        int pos = i * frame_size, not_empty = 0;
        for (int b = pos; b < pos + frame_size; b++)                    /* FrameNotEmpty (:96-107) */
            if (adpcm[b]) { not_empty = 1; break; }
        if (not_empty) {
            adpcm[pos] ^= (uint8_t)(xor >> 8);
            if (encryption_type == 9) adpcm[pos] &= 0x1f;
            adpcm[pos + 1] ^= (uint8_t)xor;
        }
        for (int c = 0; c < channel_count; c++) xor = (xor * key->mult + key->inc) & 0x7fff;
    }
}

/* GetScales + TestKey (:59-94): 1 = the key explains every frame header of these channels */
int vgo_adx_test_key(const uint8_t *const *adpcm, int adpcm_len, int nch, const vgo_adx_key *key, int encryption_type, int frame_size)
{
    int frame_count = (adpcm_len + frame_size - 1) / frame_size;
    int mask = encryption_type == 8 ? 0xE000 : 0x1000;
    int xor = key->seed;
    for (int frame = 0; frame < frame_count; frame++)
        for (int ch = 0; ch < nch; ch++) {
            int pos = frame * frame_size;
            int scale = (adpcm[ch][pos] << 8) | adpcm[ch][pos + 1];
            if (((scale ^ xor) & mask) != 0 && scale != 0) return 0;
            xor = (xor * key->mult + key->inc) & 0x7fff;
        }
    return 1;
}

/* ------------------------------------------------------------------ VGAudio.Tools/CrackAdx/GuessAdx.cs
 * The key search of the reference's `crackadx` tool for ONE file's frame scales: Run/TryScale (:118-179),
 * FindStartingKey (:181-204), AddKey's validity filter KeyIsValid (:129-147, :206-218).  The confidence report
 * (re-encode and diff, :220-262) and the key-string lookup are reporting, not search, and are not restated.
 * mults / incs: candidate lists (NULL, 0 = the reference's sets for the encryption type, :47-69).  Writes the keys that
 * survive KeyIsValid, without duplicates, sorted by (seed, mult, inc); returns their count or -1 when more than
 * max_keys were found. */
static int adx_key_cmp(const void *a, const void *b)
{
    const vgo_adx_key *x = (const vgo_adx_key *)a, *y = (const vgo_adx_key *)b;
    if (x->seed != y->seed) return x->seed < y->seed ? -1 : 1;
    if (x->mult != y->mult) return x->mult < y->mult ? -1 : 1;
    if (x->inc != y->inc) return x->inc < y->inc ? -1 : 1;
    return 0;
}

int vgo_adx_default_candidates(int encryption_type, int *mults, int *nmult, int *incs, int *ninc)
{
    int nm = 0, ni = 0;
    if (encryption_type == 8) {
        build_primes();
        for (int i = 0; i < 0x400; i++) { if (mults) mults[nm] = g_primes[i]; nm++; if (incs) incs[ni] = g_primes[i]; ni++; }
    } else if (encryption_type == 9) {
        for (int x = 0; x < 0x2000; x++) {
            if ((x & 3) == 1) { if (mults) mults[nm] = x; nm++; }
            if ((x & 1) == 1) { if (incs) incs[ni] = x; ni++; }
        }
    } else {
        return -1;
    }
    *nmult = nm;
    *ninc = ni;
    return 0;
}

int vgo_adx_guess_keys(const uint16_t *scales, int nscales, int start_frame, int encryption_type, const int *mults_in,
                       int nmult, const int *incs_in, int ninc, vgo_adx_key *out, int max_keys)
{
    if (encryption_type != 8 && encryption_type != 9) return -2;
    if (nscales <= 0 || start_frame < 0 || start_frame >= nscales) return 0;
    const int validation_mask = encryption_type == 8 ? 0xE000 : 0x1000;
    const int max_seed = encryption_type == 8 ? 0x8000 : 0x2000;
    const int xor_mask = 0x7fff;
    int *mults = NULL, *incs = NULL;
    if (!mults_in || !incs_in) {
        mults = (int *)malloc(0x2000 * sizeof(int));
        incs = (int *)malloc(0x2000 * sizeof(int));
        vgo_adx_default_candidates(encryption_type, mults, &nmult, incs, &ninc);
        mults_in = mults;
        incs_in = incs;
    }
    /* PossibleSeeds: type 8 = the 0x400 primes after 0x4000, type 9 = 0 .. 0x1FFF (ascending enumeration) */
    unsigned char *seed_ok = (unsigned char *)calloc(0x8000, 1);
    int *seed_list = (int *)malloc(0x2000 * sizeof(int));
    int nseeds = 0;
    if (encryption_type == 8) {
        build_primes();
        for (int i = 0; i < 0x400; i++) { seed_ok[g_primes[i]] = 1; seed_list[nseeds++] = g_primes[i]; }
    } else {
        for (int x = 0; x < 0x2000; x++) { seed_ok[x] = 1; seed_list[nseeds++] = x; }
    }
    int count = 0, overflow = 0;
    for (int index = 0; index < 0x1000 && !overflow; index++) {                 /* Run (:118-128) */
        const int seed = (scales[start_frame] ^ index) & (max_seed - 1);        /* TryScale (:150-179) */
        if (start_frame == 0 && !seed_ok[seed]) continue;
        for (int m = 0; m < nmult && !overflow; m++)
            for (int n = 0; n < ninc && !overflow; n++) {
                const int mult = mults_in[m], inc = incs_in[n];
                int xor = seed, match = 1;
                for (int i = start_frame; i < nscales; i++) {
                    const int scale = scales[i];
                    if (((scale ^ xor) & validation_mask) != 0 && scale != 0) { match = 0; break; }
                    xor = (xor * mult + inc) & xor_mask;
                }
                if (!match) continue;
                vgo_adx_key key = {seed, mult, inc};                            /* FindStartingKey (:181-204) */
                int have = start_frame == 0;
                for (int k = 0; k < nseeds && !have; k++) {
                    int x = seed_list[k];
                    for (int i = 0; i < start_frame; i++) x = (x * mult + inc) & xor_mask;
                    if ((x & (max_seed - 1)) == seed) { key.seed = seed_list[k]; have = 1; }
                }
                if (!have) continue;
                int dup = 0;                                                    /* AddKey: TriedKeys (:131) */
                for (int k = 0; k < count && !dup; k++) dup = adx_key_cmp(&out[k], &key) == 0;
                if (dup) continue;
                int x = key.seed, valid = 1;                                    /* KeyIsValid (:206-218) */
                for (int i = 0; i < nscales && valid; i++) {
                    if (((scales[i] ^ x) & validation_mask) != 0 && scales[i] != 0) valid = 0;
                    x = (x * key.mult + key.inc) & xor_mask;
                }
                if (!valid) continue;
                if (count >= max_keys) { overflow = 1; break; }
                out[count++] = key;
            }
    }
    free(seed_ok);
    free(seed_list);
    free(mults);
    free(incs);
    if (overflow) return -1;
    qsort(out, (size_t)count, sizeof *out, adx_key_cmp);
    return count;
}

/* ------------------------------------------------------------------ VGAudio.Tools/CrackHca/Crack.cs:43-80
 * LoadFrequencies' counting step: how often each byte value occurs at each of the first `positions` bytes of the
 * frames (the statistics the HCA key analysis starts from).  counts: positions x 256. */
void vgo_hca_byte_position_counts(const uint8_t *frames, long frames_pitch, int nstreams, int frame_count, int frame_size,
                                  int positions, uint32_t *counts)
{
    memset(counts, 0, (size_t)positions * 256 * sizeof(uint32_t));
    for (int s = 0; s < nstreams; s++)
        for (int f = 0; f < frame_count; f++)
            for (int p = 0; p < positions && p < frame_size; p++)
                counts[(size_t)p * 256 + frames[(size_t)s * frames_pitch + (size_t)f * frame_size + p]]++;
}

/* ------------------------------------------------------------------ HCA */
static void random_row(uint8_t seed, uint8_t row[16])                   /* CreateRandomRow (:116-131) */
{
    int xor = seed >> 4;
    int mult = ((seed & 1) << 3) | 5;
    int inc = (seed & 0xe) | 1;
    for (int i = 0; i < 16; i++) {
        xor = (xor * mult + inc) % 16;
        row[i] = (uint8_t)xor;
    }
}

static void shuffle_table(const uint8_t in[256], uint8_t out[256])      /* ShuffleTable (:145-162) */
{
    memset(out, 0, 256);
    uint8_t x = 0;
    int out_pos = 1;
    for (int i = 0; i < 256; i++) {
        x = (uint8_t)(x + 17);
        if (in[x] != 0 && in[x] != 0xff) out[out_pos++] = in[x];
    }
    out[0xff] = 0xff;
}

static void invert_table(const uint8_t in[256], uint8_t out[256])       /* InvertTable (:164-175) */
{
    for (int i = 0; i < 256; i++) out[in[i]] = (uint8_t)i;
}

/* key_type 56: CriHcaKey(ulong) (:8-14, :41-66); 0 / 1: CriHcaKey(Type) (:16-33, :68-100).  Returns -2 otherwise. */
int vgo_hca_key_tables(int key_type, uint64_t key_code, uint8_t decryption[256], uint8_t encryption[256])
{
    if (key_type == 0) {
        for (int i = 0; i < 256; i++) decryption[i] = (uint8_t)i;
    } else if (key_type == 1) {
        memset(decryption, 0, 256);
        int xor = 0, out_pos = 1;
        for (int i = 0; i < 256; i++) {
            xor = (xor * 13 + 11) % 256;
            if (xor != 0 && xor != 0xff) decryption[out_pos++] = (uint8_t)xor;
        }
        decryption[0xff] = 0xff;
    } else if (key_type == 56) {
        uint64_t v = key_code - 1;
        uint8_t kc[8], seed[16], table[256], row[16], column[16];
        for (int i = 0; i < 8; i++) kc[i] = (uint8_t)(v >> (8 * i));      /* BitConverter.GetBytes: little-endian */
        seed[0] = kc[1];            seed[1] = kc[6] ^ kc[1];  seed[2] = kc[2] ^ kc[3];  seed[3] = kc[2];
        seed[4] = kc[1] ^ kc[2];    seed[5] = kc[3] ^ kc[4];  seed[6] = kc[3];          seed[7] = kc[2] ^ kc[3];
        seed[8] = kc[4] ^ kc[5];    seed[9] = kc[4];          seed[10] = kc[3] ^ kc[4]; seed[11] = kc[5] ^ kc[6];
        seed[12] = kc[5];           seed[13] = kc[4] ^ kc[5]; seed[14] = kc[6] ^ kc[1]; seed[15] = kc[6];
        random_row(kc[0], row);                                          /* CreateTable (:102-114) */
        for (int r = 0; r < 16; r++) {
            random_row(seed[r], column);
            for (int c = 0; c < 16; c++) table[16 * r + c] = (uint8_t)((row[r] << 4) | column[c]);   /* CombineNibbles */
        }
        shuffle_table(table, decryption);
    } else {
        return -2;
    }
    invert_table(decryption, encryption);
    return 0;
}

/* Crypt / CryptFrame (CriHcaEncryption.cs:12-33): frames = frame_count * frame_size bytes, in place */
void vgo_hca_crypt(uint8_t *frames, int frame_count, int frame_size, const uint8_t table[256])
{
    for (int f = 0; f < frame_count; f++) {
        uint8_t *a = frames + (size_t)f * frame_size;
        for (int b = 0; b < frame_size - 2; b++) a[b] = table[a[b]];
        uint16_t crc = vgo_crc16(a, frame_size - 2);
        a[frame_size - 2] = (uint8_t)(crc >> 8);
        a[frame_size - 1] = (uint8_t)crc;
    }
}
