// capi_crypt.hip -- C ABI of the ADX / HCA encryption passes and key derivations.
#include "common.hpp"
#include "crypt_kernels.hpp"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <tuple>
#include <vector>

using namespace vga;

namespace vga { namespace hca { int crc_pow_table(const uint16_t **out); int device_info_from(const vga_hca_info &h, DeviceInfo &d); } }

namespace {

// CriAdxKey.BuildPrimesTable (CriAdxKey.cs:58-65): the 0x400 primes that follow 0x4000, from a sieve below 0x8000
const int *adx_primes()
{
    static int primes[0x400];
    static std::once_flag once;
    std::call_once(once, [] {
        std::vector<bool> composite(0x8000, false);
        for (int i = 2; i * i < 0x8000; i++)
            if (!composite[i])
                for (int j = i * i; j < 0x8000; j += i) composite[j] = true;
        int n = 0;
        for (int v = 0x4000; v < 0x8000 && n < 0x400; v++)
            if (!composite[v]) primes[n++] = v;
    });
    return primes;
}

int check_adx_crypt_args(int audio_len, int nch, int frame_size, int encryption_type)
{
    if (audio_len < 0 || nch < 0) { set_error("negative size"); return VGA_ERR_ARGUMENT; }
    if (frame_size < 2) { set_error("ADX frame size %d too small", frame_size); return VGA_ERR_ARGUMENT; }
    if (audio_len % frame_size != 0) {      // FrameNotEmpty would read past the array (CriAdxEncryption.cs:96-107)
        set_error("ADX audio length %d is not a whole number of %d-byte frames", audio_len, frame_size);
        return VGA_ERR_ARGUMENT;
    }
    (void)encryption_type;                  // any value: 9 additionally masks the first byte (:33)
    return VGA_OK;
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- ADX (Codecs/CriAdx/CriAdxKey.cs, CriAdxEncryption.cs)
int vga_adx_key_from_code(uint64_t key_code, vga_adx_key *k)            // CriAdxKey(ulong) (:17-23)
{
    if (!k) { set_error("null key"); return VGA_ERR_ARGUMENT; }
    key_code--;
    k->seed = (int)(key_code >> 27 & 0x7fff);
    k->mult = (int)((key_code >> 12 & 0x7ffc) | 1);
    k->inc = (int)((key_code << 1 & 0x7fff) | 1);
    return VGA_OK;
}

int vga_adx_key_from_string(const char *key_string, vga_adx_key *k)     // CriAdxKey(string) (:25-40)
{
    if (!k) { set_error("null key"); return VGA_ERR_ARGUMENT; }
    k->seed = k->mult = k->inc = 0;
    if (!key_string || !*key_string) return VGA_OK;                     // IsNullOrEmpty: all zero
    const int *primes = adx_primes();
    k->seed = primes[0x100];
    k->mult = primes[0x200];
    k->inc = primes[0x300];
    for (const unsigned char *c = (const unsigned char *)key_string; *c; c++) {
        if (*c >= 0x80) { set_error("ADX key strings are ASCII"); return VGA_ERR_ARGUMENT; }
        const int p = primes[*c + 0x80];
        k->seed = primes[k->seed * p % 0x400];
        k->mult = primes[k->mult * p % 0x400];
        k->inc = primes[k->inc * p % 0x400];
    }
    return VGA_OK;
}

uint64_t vga_adx_key_code(const vga_adx_key *k)                         // CriAdxKey.KeyCode (:48-56)
{
    if (!k) return 0;
    const uint64_t seed = (uint64_t)k->seed << 27;
    const uint64_t mult = (uint64_t)(k->mult & 0xfffc) << 12;
    const uint64_t inc = (uint64_t)k->inc >> 1;
    return (seed | mult | inc) + 1;
}

// CriAdxEncryption.EncryptDecrypt (:8-14), in place on the device
int vga_adx_crypt_device(uint8_t *d_audio, int64_t audio_pitch, int audio_len, int nch, const vga_adx_key *key,
                         int encryption_type, int frame_size, void *stream)
{
    if (!key) { set_error("null key"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_adx_crypt_args(audio_len, nch, frame_size, encryption_type)) return rc;
    if (audio_len == 0 || nch == 0) return VGA_OK;
    if (!d_audio || audio_pitch < audio_len) { set_error("null pointer / pitch < length"); return VGA_ERR_ARGUMENT; }
    return crypt::launch_adx_crypt(d_audio, audio_pitch, audio_len / frame_size, nch, frame_size,
                                   crypt::AdxKey{key->seed, key->mult, key->inc}, encryption_type, (hipStream_t)stream);
}

int vga_adx_crypt(uint8_t *const *audio, int audio_len, int nch, const vga_adx_key *key, int encryption_type, int frame_size)
{
    if (!key || !audio) { set_error("null argument"); return VGA_ERR_ARGUMENT; }
    if (int rc = check_adx_crypt_args(audio_len, nch, frame_size, encryption_type)) return rc;
    if (audio_len == 0 || nch == 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    DevBuf d;
    const int64_t pitch = round_up(audio_len, 16);
    VGA_HIP_TRY(d.alloc((size_t)nch * pitch));
    for (int c = 0; c < nch; c++) {
        if (!audio[c]) { set_error("audio[%d] is null", c); return VGA_ERR_ARGUMENT; }
        VGA_HIP_TRY(hipMemcpyAsync(d.as<uint8_t>() + c * pitch, audio[c], (size_t)audio_len, hipMemcpyHostToDevice, st.s));
    }
    if (int rc = vga_adx_crypt_device(d.as<uint8_t>(), pitch, audio_len, nch, key, encryption_type, frame_size, st.s)) return rc;
    for (int c = 0; c < nch; c++)
        VGA_HIP_TRY(hipMemcpyAsync(audio[c], d.as<uint8_t>() + c * pitch, (size_t)audio_len, hipMemcpyDeviceToHost, st.s));
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    return VGA_OK;
}

// CriAdxEncryption.FindKey (:43-57) over a caller-supplied candidate list: *index_out = the first key that explains
// every frame header (GetScales + TestKey, :59-94), or -1.  All candidates are tested at once, one workgroup each.
int vga_adx_find_key_device(const uint8_t *d_audio, int64_t audio_pitch, int audio_len, int nch, int encryption_type,
                            int frame_size, const vga_adx_key *keys, int nkeys, int *index_out, void *stream)
{
    if (!index_out || nkeys < 0 || (nkeys > 0 && !keys)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    *index_out = -1;
    if (int rc = check_adx_crypt_args(audio_len, nch, frame_size, encryption_type)) return rc;
    if (nkeys == 0) return VGA_OK;
    if (nch < 1) { *index_out = 0; return VGA_OK; }                     // no scales: the first key "fits"
    if (audio_len > 0 && (!d_audio || audio_pitch < audio_len)) { set_error("null pointer / pitch < length"); return VGA_ERR_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_keys, d_valid;
    VGA_HIP_TRY(d_keys.alloc((size_t)nkeys * sizeof(crypt::AdxKey)));
    VGA_HIP_TRY(d_valid.alloc((size_t)nkeys * sizeof(int)));
    VGA_HIP_TRY(hipMemcpyAsync(d_keys.p, keys, (size_t)nkeys * sizeof(crypt::AdxKey), hipMemcpyHostToDevice, s));
    if (int rc = crypt::launch_adx_test_keys(d_audio, audio_pitch, audio_len / frame_size, nch, frame_size, encryption_type,
                                             d_keys.as<crypt::AdxKey>(), nkeys, d_valid.as<int>(), s))
        return rc;
    std::vector<int> valid((size_t)nkeys);
    VGA_HIP_TRY(hipMemcpyAsync(valid.data(), d_valid.p, (size_t)nkeys * sizeof(int), hipMemcpyDeviceToHost, s));
    VGA_HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < nkeys; i++)
        if (valid[i]) { *index_out = i; break; }
    return VGA_OK;
}


// ---------------------------------------------------------------- VGAudio.Tools/CrackAdx/GuessAdx.cs
// GuessAdx's candidate sets (:47-69): type 8 = the 0x400 primes after 0x4000 for seed, multiplier and increment;
// type 9 = seeds 0..0x1FFF, multipliers = 1 mod 4, increments odd, all below 0x2000.
int vga_adx_guess_default_candidates(int encryption_type, int *mults, int *nmult, int *incs, int *ninc)
{
    if (!nmult || !ninc) { set_error("null count pointer"); return VGA_ERR_ARGUMENT; }
    int nm = 0, ni = 0;
    if (encryption_type == 8) {
        const int *pr = adx_primes();
        for (int i = 0; i < 0x400; i++) { if (mults) mults[nm] = pr[i]; nm++; if (incs) incs[ni] = pr[i]; ni++; }
    } else if (encryption_type == 9) {
        for (int x = 0; x < 0x2000; x++) {
            if ((x & 3) == 1) { if (mults) mults[nm] = x; nm++; }
            if ((x & 1) == 1) { if (incs) incs[ni] = x; ni++; }
        }
    } else {
        set_error("encryption type %d: the key search knows types 8 and 9", encryption_type);
        return VGA_ERR_ARGUMENT;
    }
    *nmult = nm;
    *ninc = ni;
    return VGA_OK;
}

// GuessAdx.Run / TryScale / FindStartingKey / AddKey's KeyIsValid filter (:118-218) for one file's frame scales
// (AdxFile.Scales :283-291: big-endian 16-bit frame headers; start_frame = AdxFile.StartFrame, the frame holding the
// first non-zero byte).  The (index, multiplier, increment) sweep runs on the device; FindStartingKey, the duplicate
// filter and KeyIsValid run here over the survivors.  Keys come back sorted by (seed, mult, inc).
int vga_adx_guess_keys(const uint16_t *scales, int nscales, int start_frame, int encryption_type, const int *mults, int nmult,
                       const int *incs, int ninc, vga_adx_key *keys_out, int max_keys, int *nkeys_out)
{
    if (!nkeys_out || max_keys < 0 || (max_keys > 0 && !keys_out)) { set_error("null / negative output arguments"); return VGA_ERR_ARGUMENT; }
    *nkeys_out = 0;
    if (encryption_type != 8 && encryption_type != 9) {
        set_error("encryption type %d: the key search knows types 8 and 9", encryption_type);
        return VGA_ERR_ARGUMENT;
    }
    if (nscales < 0 || start_frame < 0 || (nscales > 0 && !scales)) { set_error("bad scales"); return VGA_ERR_ARGUMENT; }
    if (nscales == 0 || start_frame >= nscales) return VGA_OK;
    std::vector<int> dm, di;
    if (!mults || !incs) {
        int nm = 0, ni = 0;
        dm.resize(0x2000);
        di.resize(0x2000);
        if (int rc = vga_adx_guess_default_candidates(encryption_type, dm.data(), &nm, di.data(), &ni)) return rc;
        mults = dm.data(); nmult = nm; incs = di.data(); ninc = ni;
    }
    if (nmult <= 0 || ninc <= 0) return VGA_OK;
    if (int rc = require_device()) return rc;
    const int max_seed = encryption_type == 8 ? 0x8000 : 0x2000, mask = encryption_type == 8 ? 0xE000 : 0x1000;
    // PossibleSeeds (:57, :62), in the order the reference enumerates them
    std::vector<int> seeds;
    std::vector<uint32_t> bitmap(0x8000 / 32, 0u);
    if (encryption_type == 8) { const int *pr = adx_primes(); seeds.assign(pr, pr + 0x400); }
    else for (int x = 0; x < 0x2000; x++) seeds.push_back(x);
    for (int v : seeds) bitmap[v >> 5] |= 1u << (v & 31);
    Stream st;
    VGA_HIP_TRY(st.create());
    const int cap = 1 << 20;                                         // raw survivors (short files have many)
    DevBuf d_scales, d_bitmap, d_mults, d_incs, d_out, d_count;
    VGA_HIP_TRY(d_scales.alloc((size_t)nscales * 2));
    VGA_HIP_TRY(d_bitmap.alloc(bitmap.size() * 4));
    VGA_HIP_TRY(d_mults.alloc((size_t)nmult * 4));
    VGA_HIP_TRY(d_incs.alloc((size_t)ninc * 4));
    VGA_HIP_TRY(d_out.alloc((size_t)cap * 12));
    VGA_HIP_TRY(d_count.alloc(4));
    VGA_HIP_TRY(hipMemcpyAsync(d_scales.p, scales, (size_t)nscales * 2, hipMemcpyHostToDevice, st.s));
    VGA_HIP_TRY(hipMemcpyAsync(d_bitmap.p, bitmap.data(), bitmap.size() * 4, hipMemcpyHostToDevice, st.s));
    VGA_HIP_TRY(hipMemcpyAsync(d_mults.p, mults, (size_t)nmult * 4, hipMemcpyHostToDevice, st.s));
    VGA_HIP_TRY(hipMemcpyAsync(d_incs.p, incs, (size_t)ninc * 4, hipMemcpyHostToDevice, st.s));
    VGA_HIP_TRY(hipMemsetAsync(d_count.p, 0, 4, st.s));
    if (int rc = crypt::launch_adx_guess_keys(d_scales.as<uint16_t>(), nscales, start_frame, encryption_type,
                                              start_frame == 0 ? d_bitmap.as<uint32_t>() : nullptr, d_mults.as<int>(), nmult,
                                              d_incs.as<int>(), ninc, d_out.as<int>(), cap, d_count.as<int>(), st.s))
        return rc;
    int count = 0;
    VGA_HIP_TRY(hipMemcpyAsync(&count, d_count.p, 4, hipMemcpyDeviceToHost, st.s));
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    if (count > cap) { set_error("%d candidate keys survive the scales: too few frames to search", count); return VGA_ERR_INVALID_OP; }
    std::vector<int> raw((size_t)count * 3);
    if (count > 0) VGA_HIP_TRY(hipMemcpy(raw.data(), d_out.p, (size_t)count * 12, hipMemcpyDeviceToHost));
    std::vector<std::tuple<int, int, int>> found;
    for (int i = 0; i < count; i++) {
        const int seed = raw[3 * i], mult = raw[3 * i + 1], inc = raw[3 * i + 2];
        int real = seed;
        bool have = start_frame == 0;
        for (size_t k = 0; k < seeds.size() && !have; k++) {        // FindStartingKey (:181-204)
            int x = seeds[k];
            for (int j = 0; j < start_frame; j++) x = (x * mult + inc) & 0x7fff;
            if ((x & (max_seed - 1)) == seed) { real = seeds[k]; have = true; }
        }
        if (!have) continue;
        int x = real;                                                // KeyIsValid (:206-218)
        bool valid = true;
        for (int j = 0; j < nscales && valid; j++) {
            if (((scales[j] ^ x) & mask) != 0 && scales[j] != 0) valid = false;
            x = (x * mult + inc) & 0x7fff;
        }
        if (valid) found.emplace_back(real, mult, inc);
    }
    std::sort(found.begin(), found.end());
    found.erase(std::unique(found.begin(), found.end()), found.end());
    if ((int)found.size() > max_keys) {
        set_error("%d keys found, room for %d", (int)found.size(), max_keys);
        *nkeys_out = (int)found.size();
        return VGA_ERR_ARGUMENT;
    }
    for (size_t i = 0; i < found.size(); i++) keys_out[i] = vga_adx_key{std::get<0>(found[i]), std::get<1>(found[i]), std::get<2>(found[i])};
    *nkeys_out = (int)found.size();
    return VGA_OK;
}

// ---------------------------------------------------------------- HCA (Codecs/CriHca/CriHcaKey.cs, CriHcaEncryption.cs)
// key_type 56: CriHcaKey(ulong keyCode); 0 / 1: CriHcaKey(Type).  Tables of 256 bytes each.
int vga_hca_key_tables(int key_type, uint64_t key_code, uint8_t *decryption_table, uint8_t *encryption_table)
{
    if (!decryption_table || !encryption_table) { set_error("null table"); return VGA_ERR_ARGUMENT; }
    uint8_t *dec = decryption_table;
    auto finish = [&] {                                                 // InvertTable (:164-175)
        for (int i = 0; i < 256; i++) encryption_table[dec[i]] = (uint8_t)i;
        return VGA_OK;
    };
    if (key_type == 0) {                                                // CreateDecryptionTableType0 (:68-78)
        for (int i = 0; i < 256; i++) dec[i] = (uint8_t)i;
        return finish();
    }
    if (key_type == 1) {                                                // CreateDecryptionTableType1 (:80-100)
        std::memset(dec, 0, 256);
        int x = 0, out = 1;
        for (int i = 0; i < 256; i++) {
            x = (x * 13 + 11) % 256;
            if (x != 0 && x != 0xff) dec[out++] = (uint8_t)x;
        }
        dec[0xff] = 0xff;
        return finish();
    }
    if (key_type != 56) { set_error("HCA key type %d unknown (0, 1 or 56)", key_type); return VGA_ERR_OUT_OF_RANGE; }
    auto random_row = [](uint8_t seed, uint8_t row[16]) {               // CreateRandomRow (:116-131)
        int x = seed >> 4;
        const int mult = ((seed & 1) << 3) | 5, inc = (seed & 0xe) | 1;
        for (int i = 0; i < 16; i++) { x = (x * mult + inc) % 16; row[i] = (uint8_t)x; }
    };
    const uint64_t v = key_code - 1;                                    // CreateDecryptionTable (:41-66)
    uint8_t kc[8];
    for (int i = 0; i < 8; i++) kc[i] = (uint8_t)(v >> (8 * i));
    const uint8_t seed[16] = {kc[1], (uint8_t)(kc[6] ^ kc[1]), (uint8_t)(kc[2] ^ kc[3]), kc[2],
                              (uint8_t)(kc[1] ^ kc[2]), (uint8_t)(kc[3] ^ kc[4]), kc[3], (uint8_t)(kc[2] ^ kc[3]),
                              (uint8_t)(kc[4] ^ kc[5]), kc[4], (uint8_t)(kc[3] ^ kc[4]), (uint8_t)(kc[5] ^ kc[6]),
                              kc[5], (uint8_t)(kc[4] ^ kc[5]), (uint8_t)(kc[6] ^ kc[1]), kc[6]};
    uint8_t table[256], row[16], column[16];
    random_row(kc[0], row);                                             // CreateTable (:102-114)
    for (int r = 0; r < 16; r++) {
        random_row(seed[r], column);
        for (int c = 0; c < 16; c++) table[16 * r + c] = (uint8_t)((row[r] << 4) | column[c]);
    }
    std::memset(dec, 0, 256);                                           // ShuffleTable (:145-162)
    uint8_t x = 0;
    int out = 1;
    for (int i = 0; i < 256; i++) {
        x = (uint8_t)(x + 17);
        if (table[x] != 0 && table[x] != 0xff) dec[out++] = table[x];
    }
    dec[0xff] = 0xff;
    return finish();
}

// CriHcaEncryption.Crypt (:12-18) for nstreams streams of frame_count frames, in place; table = the key's
// EncryptionTable or DecryptionTable (host memory, 256 bytes)
int vga_hca_crypt_device(uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                         const uint8_t *table, void *stream)
{
    if (!table || nstreams < 0 || frame_count < 0) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    if (frame_size < 3 || frame_size > 4096) { set_error("HCA frame size %d out of range (3..4096)", frame_size); return VGA_ERR_ARGUMENT; }
    if (nstreams == 0 || frame_count == 0) return VGA_OK;
    if (!d_frames || frames_pitch < (int64_t)frame_count * frame_size) { set_error("null pointer / pitch too small"); return VGA_ERR_ARGUMENT; }
    const uint16_t *pow = nullptr;
    if (int rc = hca::crc_pow_table(&pow)) return rc;
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_table;
    VGA_HIP_TRY(d_table.alloc(256));
    VGA_HIP_TRY(hipMemcpyAsync(d_table.p, table, 256, hipMemcpyHostToDevice, s));
    if (int rc = crypt::launch_hca_crypt(d_frames, frames_pitch, nstreams, frame_count, frame_size, d_table.as<uint8_t>(), pow, s)) return rc;
    VGA_HIP_TRY(hipStreamSynchronize(s));                               // d_table is freed on return
    return VGA_OK;
}

int vga_hca_crypt(uint8_t *frames, int frame_count, int frame_size, const uint8_t *table)
{
    if (!table || frame_count < 0) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    if (frame_size < 3 || frame_size > 4096) { set_error("HCA frame size %d out of range (3..4096)", frame_size); return VGA_ERR_ARGUMENT; }
    if (frame_count == 0) return VGA_OK;
    if (!frames) { set_error("null frames"); return VGA_ERR_ARGUMENT; }
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    DevBuf d;
    const size_t bytes = (size_t)frame_count * frame_size;
    VGA_HIP_TRY(d.alloc(bytes));
    VGA_HIP_TRY(hipMemcpyAsync(d.p, frames, bytes, hipMemcpyHostToDevice, st.s));
    if (int rc = vga_hca_crypt_device(d.as<uint8_t>(), (int64_t)bytes, 1, frame_count, frame_size, table, st.s)) return rc;
    VGA_HIP_TRY(hipMemcpyAsync(frames, d.p, bytes, hipMemcpyDeviceToHost, st.s));
    VGA_HIP_TRY(hipStreamSynchronize(st.s));
    return VGA_OK;
}

// CriHcaEncryption.FindKey (CriHcaEncryption.cs:34-46) over caller-supplied candidates: decryption_tables = nkeys x 256
// bytes (CriHcaKey.DecryptionTable, vga_hca_key_tables) in host memory; *index_out = the first key under which the first
// ten non-empty frames of the stream unpack (TestKey :48-63), or -1.  A frame whose sync word is wrong is the
// reference's InvalidDataException.
int vga_hca_find_key_device(const vga_hca_info *h, const uint8_t *d_frames, int frame_count, const uint8_t *decryption_tables,
                            int nkeys, int *index_out, void *stream)
{
    if (!h || !index_out || nkeys < 0 || frame_count < 0 || (nkeys > 0 && !decryption_tables)) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    *index_out = -1;
    hca::DeviceInfo d;
    if (int rc = hca::device_info_from(*h, d)) return rc;
    if (nkeys == 0) return VGA_OK;
    if (frame_count == 0) { *index_out = 0; return VGA_OK; }            // no frame to refute the first key
    if (!d_frames) { set_error("null frames"); return VGA_ERR_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_tables, d_valid, d_small;
    VGA_HIP_TRY(d_tables.alloc((size_t)nkeys * 256));
    VGA_HIP_TRY(d_valid.alloc((size_t)nkeys * sizeof(int)));
    VGA_HIP_TRY(d_small.alloc(2 * sizeof(int)));
    VGA_HIP_TRY(hipMemcpyAsync(d_tables.p, decryption_tables, (size_t)nkeys * 256, hipMemcpyHostToDevice, s));
    if (int rc = crypt::launch_hca_find_key(d_frames, frame_count, d, d_tables.as<uint8_t>(), nkeys, d_small.as<int>(), d_valid.as<int>(),
                                            d_small.as<int>() + 1, s))
        return rc;
    std::vector<int> valid((size_t)nkeys);
    int flags = 0;
    VGA_HIP_TRY(hipMemcpyAsync(valid.data(), d_valid.p, (size_t)nkeys * sizeof(int), hipMemcpyDeviceToHost, s));
    VGA_HIP_TRY(hipMemcpyAsync(&flags, d_small.as<int>() + 1, sizeof(int), hipMemcpyDeviceToHost, s));
    VGA_HIP_TRY(hipStreamSynchronize(s));
    (void)flags;
    // keys in the caller's order, as the reference tries them: the first that unpacks wins; one that meets a wrong sync
    // word before its first unpack failure is where the reference throws -- unless an earlier key had already won
    for (int i = 0; i < nkeys; i++) {
        if (valid[i] == 2) { set_error("Invalid frame header"); return VGA_ERR_INVALID_DATA; }
        if (valid[i] == 1) { *index_out = i; break; }
    }
    return VGA_OK;
}

int vga_hca_find_key(const vga_hca_info *h, const uint8_t *frames, int frame_count, const uint8_t *decryption_tables, int nkeys,
                     int *index_out)
{
    if (!h || !index_out || frame_count < 0) { set_error("null / negative argument"); return VGA_ERR_ARGUMENT; }
    *index_out = -1;
    if (frame_count > 0 && !frames) { set_error("null frames"); return VGA_ERR_ARGUMENT; }
    if (nkeys <= 0 || frame_count == 0)
        return vga_hca_find_key_device(h, nullptr, frame_count, decryption_tables, nkeys, index_out, nullptr);
    if (int rc = require_device()) return rc;
    Stream st;
    VGA_HIP_TRY(st.create());
    // FindFirstNonEmptyFrame on the host copy: only the ten frames TestKey reads travel
    const int fs = h->frame_size;
    if (fs < 8) { set_error("frame size %d", fs); return VGA_ERR_ARGUMENT; }
    int first = 0;
    for (int i = 0; i < frame_count; i++) {
        bool empty = true;
        for (int b = 2; b < fs - 2 && empty; b++) empty = frames[(size_t)i * fs + b] == 0;
        if (!empty) { first = i; break; }
    }
    const int n = std::min(10, frame_count - first);
    DevBuf d;
    VGA_HIP_TRY(d.alloc((size_t)n * fs));
    VGA_HIP_TRY(hipMemcpyAsync(d.p, frames + (size_t)first * fs, (size_t)n * fs, hipMemcpyHostToDevice, st.s));
    // the copied window starts at the first non-empty frame (or at frame 0 when all are empty): the device search
    // finds it at index 0 again
    return vga_hca_find_key_device(h, d.as<uint8_t>(), n, decryption_tables, nkeys, index_out, st.s);
}

// VGAudio.Tools/CrackHca/Crack.cs:43-80 (LoadFrequencies): counts[p * 256 + v] = frames whose byte p equals v, for the
// first `positions` bytes (the reference uses 30); counts_out in host memory, frames on the device.
int vga_hca_byte_position_counts_device(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count,
                                        int frame_size, int positions, uint32_t *counts_out, void *stream)
{
    if (!counts_out || positions < 1 || positions > 64 || nstreams < 0 || frame_count < 0 || frame_size < 1) {
        set_error("bad arguments (positions 1..64)");
        return VGA_ERR_ARGUMENT;
    }
    if (nstreams > 0 && frame_count > 0 && (!d_frames || frames_pitch < (int64_t)frame_count * frame_size)) { set_error("null pointer / pitch too small"); return VGA_ERR_ARGUMENT; }
    hipStream_t s = (hipStream_t)stream;
    DevBuf d_counts;
    VGA_HIP_TRY(d_counts.alloc((size_t)positions * 256 * sizeof(unsigned)));
    if (int rc = crypt::launch_hca_byte_position_counts(d_frames, frames_pitch, nstreams, frame_count, frame_size, positions,
                                                        d_counts.as<unsigned>(), s))
        return rc;
    VGA_HIP_TRY(hipMemcpyAsync(counts_out, d_counts.p, (size_t)positions * 256 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
    VGA_HIP_TRY(hipStreamSynchronize(s));
    return VGA_OK;
}

}  // extern "C"
