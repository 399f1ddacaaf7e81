// crypt_kernels.hpp -- launchers of the ADX / HCA encryption passes
#pragma once
#include "common.hpp"
#include "hca_device.hpp"

namespace vga {
namespace crypt {

struct AdxKey { int seed, mult, inc; };    // CriAdxKey (Codecs/CriAdx/CriAdxKey.cs:42-44)

int launch_adx_crypt(uint8_t *d_audio, int64_t pitch, int frame_count, int nch, int frame_size, const AdxKey &key,
                     int encryption_type, hipStream_t stream);
int launch_adx_test_keys(const uint8_t *d_audio, int64_t pitch, int frame_count, int nch, int frame_size, int encryption_type,
                         const AdxKey *d_keys, int nkeys, int *d_valid, hipStream_t stream);
// d_crc_pow: uint16[4096], x^(8k) mod 0x18005 (the HCA encoder's table)
int launch_hca_crypt(uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                     const uint8_t *d_table, const uint16_t *d_crc_pow, hipStream_t stream);

// GuessAdx.Run / TryScale (VGAudio.Tools/CrackAdx/GuessAdx.cs:118-179): survivors appended to d_out[cap][3], *d_count = how many
int launch_adx_guess_keys(const uint16_t *d_scales, int nscales, int start_frame, int encryption_type, const uint32_t *d_seed_bitmap,
                          const int *d_mults, int nmult, const int *d_incs, int ninc, int *d_out, int cap, int *d_count,
                          hipStream_t stream);
// CriHcaEncryption.FindKey / TestKey (CriHcaEncryption.cs:34-88): d_valid[k] = 1 when key k unpacks the first ten non-empty frames
int launch_hca_find_key(const uint8_t *d_frames, int frame_count, const hca::DeviceInfo &info, const uint8_t *d_tables, int nkeys,
                        int *d_first, int *d_valid, int *d_flags, hipStream_t stream);
// Crack.LoadFrequencies' counting step (VGAudio.Tools/CrackHca/Crack.cs:43-80)
int launch_hca_byte_position_counts(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                                    int positions, unsigned *d_counts, hipStream_t stream);

}  // namespace crypt
}  // namespace vga
