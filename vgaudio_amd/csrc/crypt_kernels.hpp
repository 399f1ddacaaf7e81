// crypt_kernels.hpp -- launchers of the ADX / HCA encryption passes
#pragma once
#include "common.hpp"

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

}  // namespace crypt
}  // namespace vga
