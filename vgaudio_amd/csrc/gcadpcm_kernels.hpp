// gcadpcm_kernels.hpp -- launchers for the GC-ADPCM kernels (all pointers are device pointers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vga {
namespace gc {

// A ragged batch (the `*_v` entry points: channels of different lengths, the reference's file-level batch,
// VGAudio.Cli/Batch.cs:24-25): per-channel shapes in device tables.  Channel c's PCM starts at d_pcm + pcm_off[c]
// (samples; a multiple of 8), its ADPCM at d_adpcm + adpcm_off[c] (bytes; a multiple of 16), its records at
// workspace + rec_off[c].  Work slots (workgroup / wave / lane indices) map to channels through `order`, longest channel
// first: the slots that share a wave or a workgroup then hold channels of similar length and the long ones start first.
// The kernels take the struct by value; order == nullptr means "uniform batch" (rows `pitch` apart, one length).
// Where the time pieces of a channel begin (gc_encode_kernel.hip): `nb` pieces of `big` frames, then pieces of `small`
// frames.  The persistent encoder workgroups take their items in an order that ends with the short ones, so that little
// is left to wait for when the queue runs dry (a plain grid uses one size: nb = every piece).
struct Pieces {
    int big = 0, nb = 0, small = 0;
    __host__ __device__ int64_t first(int k) const { return (int64_t)(k < nb ? k : nb) * big + (int64_t)(k > nb ? k - nb : 0) * small; }
    __host__ __device__ int frames(int k) const { return k < nb ? big : small; }
};
// The encoder's piece schedule for `groups` channel groups (16 channels each) whose longest channel has `frames` frames and
// which hold `group_frames` frames of work in all (sum over groups of the group's longest channel); persistent = workgroups
// that take (group, piece) items from a queue (the (channel, predictor) layout only: layout == 8).  Returns the number of
// pieces the longest channel has.
int plan_encode_pieces(int groups, int frames, int64_t group_frames, bool ragged, bool *persistent, Pieces *seg, int layout = 8);
// ... on a device with `cus` compute units (host arithmetic only)
int plan_encode_pieces_on(int cus, int groups, int frames, int64_t group_frames, bool ragged, bool *persistent, Pieces *seg, int layout = 8);

struct Ragged {
    // the encoder's items (channel group | piece << 20), biggest first, when the host has planned them (ragged batches)
    const uint32_t *items = nullptr;
    int n_items = 0, segments = 0, persistent = 0;
    Pieces seg;
    const int *order = nullptr;         // [nch] work slot -> channel
    const int *length = nullptr;        // [nch] samples
    const int64_t *pcm_off = nullptr;   // [nch]
    const int64_t *adpcm_off = nullptr; // [nch]
    const int64_t *rec_off = nullptr;   // [nch]
    int solo_channels = 0;              // coefficient search: the first n work slots get five waves each (ragged_solo_count)
    int solo_usable = 0;                // ... and how many could (test hook: coefficient kernel variant 3)
    int max_length = 0;                 // host-side copy: the longest channel
    int64_t total_frames = 0;           // host-side copy: sum of ceil(length / 14)
};

// Records of the coefficient search (workspace): a channel of `frames` frames owns coef_record_pitch(frames) slots of 16
// bytes -- whole blocks of 256, inside which gc_coefs_kernel permutes the records (record_position, gcadpcm_kernels.hip).
constexpr int COEF_RECORD_BLOCK = 256;
__host__ __device__ inline int64_t coef_record_pitch(int64_t frames)
{
    return (frames > 0 ? frames + COEF_RECORD_BLOCK - 1 : COEF_RECORD_BLOCK) / COEF_RECORD_BLOCK * COEF_RECORD_BLOCK;
}

int ragged_solo_count(const int *lengths_longest_first, int nch, int64_t total_frames, int cus, int *usable_out = nullptr);
// rg != nullptr: pcm_pitch / length (sample_count) are ignored, d_pcm / d_adpcm are the bases the offsets count from
int launch_coefs(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length, int16_t *d_coefs,
                 void *d_workspace, hipStream_t stream, const Ragged *rg = nullptr);
// d_scratch (optional, encode_scratch_bytes(nch) bytes): where the time pieces' states go; allocated stream-ordered when null
int launch_encode(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count, const int16_t *d_coefs,
                  const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, int64_t adpcm_pitch,
                  hipStream_t stream, void *d_scratch = nullptr, size_t scratch_bytes = 0, const Ragged *rg = nullptr);
size_t encode_scratch_bytes(int nch);
// gc_decode_kernel.hip
int launch_decode(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch, int sample_count,
                  const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int64_t pcm_pitch, int *d_status,
                  hipStream_t stream, const Ragged *rg = nullptr);
// gc_channel_kernels.hip (channel metadata)
int launch_align_gather(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int loop_start, int loop_end,
                        int samples_to_keep, int samples_to_encode, int16_t *d_new_pcm, int64_t new_pitch,
                        int16_t *d_hist1, int16_t *d_hist2, hipStream_t stream);
int launch_channel_meta(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_pcm, int64_t pcm_pitch, int nch,
                        int loop_start, int samples_per_entry, int entries, int16_t *d_seek, int64_t seek_pitch,
                        int16_t *d_loop_context, hipStream_t stream);
int launch_dsp_image(const uint8_t *d_adpcm, int64_t adpcm_pitch, int adpcm_len, const int16_t *d_coefs,
                     const int16_t *d_gain, const int16_t *d_start_context, const int16_t *d_loop_context, int nch,
                     int sample_count, int nibble_count, int sample_rate, int looping, int start_addr, int end_addr,
                     int cur_addr, int bytes_per_interleave, int frames_per_interleave, int audio_data_size,
                     int mono_bytes, uint8_t *d_file, size_t file_size, hipStream_t stream);
int launch_synth(int16_t *d_pcm, int64_t pitch, int nch, int length, int first_channel, const uint32_t *d_params,
                 hipStream_t stream);

}  // namespace gc
}  // namespace vga
