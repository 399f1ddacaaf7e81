/*
 * vgaudio_hip.h -- C ABI of libvgaudio_hip.so, the MI355X (gfx950) batch
 * audio-codec engine that drops in behind VGAudio's IAudioFormat seam.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to /root/reference/src/).  Signatures use plain pointers and sizes only; the
 * caller owns (and, from C#, pins) every buffer, the callee keeps no pointer
 * past return.  All exports are thread-safe and re-entrant: host-buffer
 * entry points create and destroy their own HIP stream; *_device entry points
 * run on the caller's stream and never synchronise it.
 *
 * Errors: the reference throws .NET exceptions; here every fallible function
 * returns an int status that the managed shim maps back:
 *   VGA_OK                 0
 *   VGA_ERR_ARGUMENT      -1   ArgumentException
 *   VGA_ERR_OUT_OF_RANGE  -2   ArgumentOutOfRangeException
 *   VGA_ERR_INVALID_DATA  -3   InvalidDataException
 *   VGA_ERR_INVALID_OP    -4   InvalidOperationException
 *   VGA_ERR_DEVICE        -5   HIP runtime failure / no gfx950 device (no CPU fallback exists)
 * vga_last_error() returns a thread-local message for the last failure.
 */
#ifndef VGAUDIO_HIP_H
#define VGAUDIO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGA_OK 0
#define VGA_ERR_ARGUMENT (-1)
#define VGA_ERR_OUT_OF_RANGE (-2)
#define VGA_ERR_INVALID_DATA (-3)
#define VGA_ERR_INVALID_OP (-4)
#define VGA_ERR_DEVICE (-5)

const char *vga_last_error(void);
/* number of visible HIP devices (0 when none; never fails) */
int vga_device_count(void);
/* selects the device for the calling thread (one process per GPU: LOCAL_RANK) */
int vga_set_device(int device);
/* ONE process, several GPUs -- the shape of the reference's own parallelism (Parallel.For over channels inside one
 * process, Formats/GcAdpcm/GcAdpcmFormat.cs:65-68; a worker per file, VGAudio.Cli/Batch.cs:24-25) and of a P/Invoke host.
 * Lists the devices every host-buffer (`*_batch`) entry point spreads a call's units (channels / streams) over: contiguous
 * shares, one per listed device, each share a whole pipelined call on its own host thread with its own device buffers and
 * PCIe link; results land directly in the caller's rows, there is no collective and no peer copy.  Shares are at least
 * 128 channels / 32 streams, so small calls stay on the first listed device.  count = 0 (the default) restores "the calling
 * thread's current device, nothing is spread".  A device may be listed more than once.  Process-wide; takes effect for
 * calls that start afterwards.  Output bytes do not depend on the list. */
int vga_set_devices(const int *devices, int count);
/* the current list: returns its length and fills devices[0 .. min(length, capacity)) */
int vga_get_devices(int *devices, int capacity);
/* Progress of the host-buffer (`*_batch`) entry points: IProgressReport (VGAudio/IProgressReport.cs:3-28), which the
 * reference drives with ReportAdd(1) per frame (GcAdpcmEncoder.cs:42, CriAdxCodec.cs:101, CriHcaFormat.cs:71,79).  Here a
 * call works through its channels / streams in chunks (512 or 1024 channels, 256 streams; one chunk for small calls), and
 * fn(user, done, total) is called once per chunk when that chunk's ROWS (adpcm_out / pcm_out / frames_out / out) are
 * complete in the caller's memory: `done` counts channels (GC-ADPCM, ADX) or streams (HCA) finished so far, `total` is the
 * call's count; the host multiplies by its frames per channel for ReportAdd.  The small per-call arrays (coefs_out,
 * history_out, info_out) are complete when the call RETURNS, not when done == total is reported (a coefficients-only
 * call has no rows: its chunks are reported when their kernels have run).  With vga_set_devices() the shares of all devices report into the same count.
 * fn runs on a worker thread of the library while the call is in progress, never concurrently with itself; it must not
 * call back into the library and should return quickly (a drainer thread is waiting for it).  Per calling thread: the
 * callback applies to `*_batch` calls made afterwards from the thread that set it; fn = NULL removes it. */
typedef void (*vga_progress_fn)(void *user, int64_t done, int64_t total);
int vga_set_progress_callback(vga_progress_fn fn, void *user);
/* The host-buffer entry points keep the device buffers and the page-locked staging rings of their last calls for the
 * next one (allocation costs more than a call's transfers and kernels: ~1.4 s for the 44 GB of BASELINE configs[1]); at
 * most 64 GiB of device memory PER DEVICE (environment variable VGA_HIP_POOL_GIB changes the figure, 0 turns the cache off)
 * and 1 GiB of pinned host memory stay parked -- memory other allocators in the process (torch, ...) cannot see or reclaim.
 * This returns all of it to the system. */
void vga_release_cached_memory(void);
/* Process-wide side effects of the host-buffer (`*_batch`) entry points, stated here because a drop-in must not surprise
 * its host (details: INTEGRATION.md, "What the host pipeline does to the process"):
 *  - loading the library sets the environment variable GPU_MAX_HW_QUEUES=16 unless the host has set it or has set
 *    VGA_HIP_NO_ENV=1 (the HIP runtime reads the variable when it initialises; with the default of 4 hardware queues a
 *    copy stream shares a queue with a kernel stream and every download waits for the last kernel).  setenv is not
 *    thread-safe against a host thread calling getenv at that moment: a host that cares sets the variable itself before
 *    loading the library, or opts out.  If the runtime was already up when the library was loaded (it then never saw the
 *    value), or fewer than 6 queues are configured, the calls run one kernel lane instead of two -- slower, not wrong;
 *  - input rows of 256 KB or more (and the output rows of a decode) are page-locked with hipHostRegister for the duration
 *    of the call; a row that cannot be registered is copied through the runtime's pageable path;
 *  - one feeder and two drainer threads per call, joined before it returns. */
/* library version string */
const char *vga_version(void);

/* ======================================================================
 * GC-ADPCM (Nintendo DSP-ADPCM)
 * ====================================================================== */

/* VGAudio/Codecs/GcAdpcm/GcAdpcmMath.cs:11-47 (host-side, no device needed) */
int vga_gcadpcm_nibble_count_to_sample_count(int nibble_count);
int vga_gcadpcm_sample_count_to_nibble_count(int sample_count);
int vga_gcadpcm_nibble_to_sample(int nibble);
int vga_gcadpcm_sample_to_nibble(int sample);
int vga_gcadpcm_sample_count_to_byte_count(int sample_count);
int vga_gcadpcm_byte_count_to_sample_count(int byte_count);

/* Replaces the body of GcAdpcmFormat.EncodeFromPcm16
 * (VGAudio/Formats/GcAdpcm/GcAdpcmFormat.cs:58-74 + EncodeChannel :129-135):
 * for every channel, CalculateCoefficients (GcAdpcmCoefficients.cs:9) then
 * Encode (GcAdpcmEncoder.cs:14).  pcm[c] -> sample_count shorts (planar, the
 * layout of Pcm16Format.Channels); coefs_out -> nch*16 shorts; adpcm_out[c] ->
 * vga_gcadpcm_sample_count_to_byte_count(sample_count) bytes.  hist1/hist2 are
 * GcAdpcmParameters.History1/2 (0 when config is null). */
int vga_gcadpcm_encode_batch(const int16_t *const *pcm, int nch, int sample_count,
                             int16_t hist1, int16_t hist2,
                             int16_t *coefs_out, uint8_t *const *adpcm_out);

/* Static-codec level, batched: GcAdpcmCoefficients.CalculateCoefficients
 * (GcAdpcmCoefficients.cs:9) for nch channels of `length` samples. */
int vga_gcadpcm_calculate_coefficients_batch(const int16_t *const *pcm, int nch, int length,
                                             int16_t *coefs_out);

/* GcAdpcmEncoder.Encode (GcAdpcmEncoder.cs:14) with caller-supplied coefs.
 * coefs: nch*16.  sample_count == -1 means pcm_length (CodecParameters.SampleCount,
 * VGAudio/Codecs/CodecParameters.cs:6); sample_count > pcm_length -> VGA_ERR_ARGUMENT
 * (the reference's Array.Copy throws).  hist1/hist2: per-channel arrays (nch) or NULL for 0. */
int vga_gcadpcm_encode_with_coefs_batch(const int16_t *const *pcm, int nch, int pcm_length,
                                        int sample_count, const int16_t *coefs,
                                        const int16_t *hist1, const int16_t *hist2,
                                        uint8_t *const *adpcm_out);

/* Replaces the body of GcAdpcmFormat.ToPcm16 (GcAdpcmFormat.cs:42-54 ->
 * GcAdpcmChannel.GetPcmAudio -> GcAdpcmDecoder.Decode, GcAdpcmDecoder.cs:10-54).
 * adpcm[c] holds sample_count_to_byte_count(sample_count) bytes, coefs nch*16,
 * hist1/hist2 per-channel arrays or NULL; pcm_out[c] -> sample_count shorts.
 * A predictor index > 7 in a frame header -> VGA_ERR_ARGUMENT (IndexOutOfRange in C#). */
int vga_gcadpcm_decode_batch(const uint8_t *const *adpcm, const int16_t *coefs, int nch,
                             int sample_count, const int16_t *hist1, const int16_t *hist2,
                             int16_t *const *pcm_out);

/* ---- dsptool-compatible single-channel exports -------------------------
 * Same names/signatures as the cdecl entry points VGAudio.Tools binds from
 * Nintendo's dsptool DLLs (VGAudio.Tools/GcAdpcm/DspToolDll.cs:16-29,94-108;
 * struct layout VGAudio.Tools/GcAdpcm/Native.cs:18-32), so `VGAudio.Tools
 * gcadpcm` can A/B this library unchanged as an "OpenSource"-type DLL. */
#pragma pack(push, 1)
typedef struct {
    int16_t coef[16];
    uint16_t gain;
    uint16_t pred_scale;
    int16_t yn1;
    int16_t yn2;
    uint16_t loop_pred_scale;
    int16_t loop_yn1;
    int16_t loop_yn2;
} ADPCMINFO;
#pragma pack(pop)

void encode(int16_t *src, uint8_t *dst, ADPCMINFO *cxt, uint32_t samples);
void decode(uint8_t *src, int16_t *dst, ADPCMINFO *cxt, uint32_t samples);
void correlateCoefs(int16_t *src, uint32_t samples, int16_t *coefsOut);
/* one 14-sample frame: src = 16 shorts (2 history + 14), reconstructed in place */
void encodeFrame(int16_t *src, uint8_t *dst, int16_t *coefs, uint8_t one);

/* ---- device-resident GC-ADPCM (inputs/outputs already in HBM) ----------
 * d_pcm: planar, channel c at d_pcm + c*pcm_pitch (pitch in samples, even,
 * base 4-byte aligned).  d_adpcm: channel c at d_adpcm + c*adpcm_pitch
 * (bytes, multiple of 8, base 8-byte aligned).  stream: hipStream_t. */
size_t vga_gcadpcm_coefs_workspace_bytes(int nch, int length);
int vga_gcadpcm_coefs_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int length,
                             int16_t *d_coefs, void *d_workspace, size_t workspace_bytes,
                             void *stream);
int vga_gcadpcm_encode_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int sample_count,
                              const int16_t *d_coefs, const int16_t *d_hist1, const int16_t *d_hist2,
                              uint8_t *d_adpcm, int64_t adpcm_pitch, void *stream);
int vga_gcadpcm_decode_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs,
                              int nch, int sample_count, const int16_t *d_hist1, const int16_t *d_hist2,
                              int16_t *d_pcm, int64_t pcm_pitch, int *d_status, void *stream);

/* ---- ragged batches: channels of DIFFERENT lengths in one call ------------
 * The reference's batch conversion is a Parallel.ForEach over files (VGAudio.Cli/Batch.cs:24-25 -> Convert.cs:19): every
 * file has its own length and channel count and ends in GcAdpcmFormat.EncodeFromPcm16 (GcAdpcmFormat.cs:58-74), one
 * EncodeChannel (:129-135) per channel.  The `_v` entry points take the channels of many files at once, each with its own
 * sample count; results are byte for byte those of one call per channel.  Equal counts take the equal-length kernels.
 * hist1 / hist2: per-channel arrays (GcAdpcmParameters.History1/2 of that channel's file) or NULL for 0.
 * pcm[c] / adpcm_out[c] may be NULL for a channel of 0 samples.  coefs_out: nch*16 (a channel without samples gets the
 * coefficients the reference computes for an empty array: all 0). */
int vga_gcadpcm_encode_batch_v(const int16_t *const *pcm, const int *sample_counts, int nch,
                               const int16_t *hist1, const int16_t *hist2,
                               int16_t *coefs_out, uint8_t *const *adpcm_out);
int vga_gcadpcm_calculate_coefficients_batch_v(const int16_t *const *pcm, const int *lengths, int nch,
                                               int16_t *coefs_out);
int vga_gcadpcm_encode_with_coefs_batch_v(const int16_t *const *pcm, const int *sample_counts, int nch,
                                          const int16_t *coefs, const int16_t *hist1, const int16_t *hist2,
                                          uint8_t *const *adpcm_out);
int vga_gcadpcm_decode_batch_v(const uint8_t *const *adpcm, const int16_t *coefs, const int *sample_counts, int nch,
                               const int16_t *hist1, const int16_t *hist2, int16_t *const *pcm_out);
/* Device-resident ragged batches.  vga_gcadpcm_ragged_create() fixes the batch's shape (the channels' sample counts) and
 * the PACKED layout of its device buffers: channel c's PCM starts pcm_offsets[c] samples into d_pcm, its ADPCM
 * adpcm_offsets[c] bytes into d_adpcm (rows follow each other, rounded up to 8 samples / 16 bytes); d_pcm must hold
 * vga_gcadpcm_ragged_pcm_samples() samples and d_adpcm vga_gcadpcm_ragged_adpcm_bytes() bytes (both include a guard the
 * kernels' clamped loads may touch), both 16-byte aligned.  The object keeps its tables in the memory of the device that
 * was current when it was created and may be used by any number of calls (concurrently too); d_coefs: nch*16 shorts,
 * d_hist1 / d_hist2: nch shorts or NULL. */
typedef struct vga_gcadpcm_ragged vga_gcadpcm_ragged;
int vga_gcadpcm_ragged_create(const int *sample_counts, int nch, vga_gcadpcm_ragged **out);
void vga_gcadpcm_ragged_destroy(vga_gcadpcm_ragged *r);
int vga_gcadpcm_ragged_channels(const vga_gcadpcm_ragged *r);
int64_t vga_gcadpcm_ragged_pcm_samples(const vga_gcadpcm_ragged *r);
int64_t vga_gcadpcm_ragged_adpcm_bytes(const vga_gcadpcm_ragged *r);
size_t vga_gcadpcm_ragged_coefs_workspace_bytes(const vga_gcadpcm_ragged *r);
int vga_gcadpcm_ragged_offsets(const vga_gcadpcm_ragged *r, int64_t *pcm_offsets_out, int64_t *adpcm_offsets_out);
int vga_gcadpcm_coefs_device_v(const vga_gcadpcm_ragged *r, const int16_t *d_pcm, int16_t *d_coefs,
                               void *d_workspace, size_t workspace_bytes, void *stream);
int vga_gcadpcm_encode_device_v(const vga_gcadpcm_ragged *r, const int16_t *d_pcm, const int16_t *d_coefs,
                                const int16_t *d_hist1, const int16_t *d_hist2, uint8_t *d_adpcm, void *stream);
int vga_gcadpcm_decode_device_v(const vga_gcadpcm_ragged *r, const uint8_t *d_adpcm, const int16_t *d_coefs,
                                const int16_t *d_hist1, const int16_t *d_hist2, int16_t *d_pcm, int *d_status,
                                void *stream);

/* ----------------------------------------------------------------------
 * GC-ADPCM channel metadata: what the reference derives when a channel is
 * built after encoding -- GcAdpcmChannel(GcAdpcmChannelBuilder), VGAudio/Formats/GcAdpcm/
 * GcAdpcmChannel.cs:31-55 -> GcAdpcmChannelBuilder.GetAlignment / GetLoopContext / GetSeekTable
 * (GcAdpcmChannelBuilder.cs:148-202).  One loop per batch: GcAdpcmFormat applies its loop to every
 * channel (GcAdpcmFormat.cs:27-40).
 * -------------------------------------------------------------------- */
typedef struct {
    int sample_count;                  /* GcAdpcmChannelBuilder.SampleCount (unaligned) */
    int looping, loop_start, loop_end; /* WithLoop (:103-120); not looping: pass 0, 0 */
    int loop_alignment_multiple;       /* WithLoopAlignment (:64-68); 0 = none */
    int samples_per_seek_table_entry;  /* WithSamplesPerSeekTableEntry (:78-87); 0 = no seek table */
} vga_gcadpcm_channel_params;
typedef struct {
    int alignment_needed;              /* GcAdpcmAlignment.AlignmentNeeded (GcAdpcmAlignment.cs:25) */
    int loop_start_aligned;            /* LoopStartAligned (:30), or loop_start */
    int sample_count_aligned;          /* SampleCountAligned (:31), or sample_count: GcAdpcmChannel.SampleCount */
    int seek_table_entries;            /* GcAdpcmSeekTable.cs:27; the table holds 2 shorts per entry */
} vga_gcadpcm_channel_layout;
/* size math only (no device needed); VGA_ERR_OUT_OF_RANGE for negative / inverted loop points */
int vga_gcadpcm_channel_layout_for(const vga_gcadpcm_channel_params *p, vga_gcadpcm_channel_layout *out);
/* adpcm[c]: SampleCountToByteCount(sample_count) bytes; coefs: nch*16.  Outputs (each may be NULL):
 *   adpcm_out[c]       SampleCountToByteCount(sample_count_aligned) bytes = GetAdpcmAudio()
 *                      (required when the loop needs alignment: GcAdpcmAlignment.cs:20-63 re-encodes the tail
 *                      from the wrapped loop with the history of the last kept frame);
 *   pcm_out[c]         sample_count_aligned shorts = the decoded PCM the builder ends up holding;
 *   seek_table_out[c]  2*seek_table_entries shorts (GcAdpcmSeekTable.cs:25-38);
 *   loop_context_out   nch*3 shorts: pred/scale byte, hist1, hist2 (GcAdpcmLoopContext.cs:17-26; a loop
 *                      start of 0 yields the default (0,0,0) like the reference, whose builder considers
 *                      the unset context valid for LoopContextStart == 0, GcAdpcmChannelBuilder.cs:135-137).
 * VGA_ERR_INVALID_OP: zero-length loop that needs alignment (the reference never returns);
 * VGA_ERR_OUT_OF_RANGE: aligned loop start past the ORIGINAL data (the reference indexes b.Adpcm, :179). */
int vga_gcadpcm_build_channels_batch(const uint8_t *const *adpcm, const int16_t *coefs, int nch,
                                     const vga_gcadpcm_channel_params *p, uint8_t *const *adpcm_out,
                                     int16_t *const *pcm_out, int16_t *const *seek_table_out,
                                     int16_t *loop_context_out);
/* device-resident variant: layouts as vga_gcadpcm_encode_device / decode_device; seek_pitch in shorts;
 * d_workspace: 16-byte aligned, vga_gcadpcm_build_channels_workspace_bytes(nch, p) bytes. */
size_t vga_gcadpcm_build_channels_workspace_bytes(int nch, const vga_gcadpcm_channel_params *p);
int vga_gcadpcm_build_channels_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, const int16_t *d_coefs, int nch,
                                      const vga_gcadpcm_channel_params *p, uint8_t *d_adpcm_out, int64_t out_pitch,
                                      int16_t *d_pcm_out, int64_t pcm_pitch, int16_t *d_seek_out, int64_t seek_pitch,
                                      int16_t *d_loop_context_out, void *d_workspace, size_t workspace_bytes,
                                      void *stream);

/* ----------------------------------------------------------------------
 * DSP container for GC-ADPCM: VGAudio/Containers/Dsp/DspWriter.cs:14-103.
 * One 0x60-byte big-endian header per channel, then the audio -- one channel verbatim, several channels
 * interleaved in blocks of SampleCountToByteCount(SamplesPerInterleave) bytes (Utilities/Interleave.cs:43-78).
 * -------------------------------------------------------------------- */
typedef struct {
    int sample_rate;
    int sample_count;                  /* GcAdpcmFormat.SampleCount */
    int looping, loop_start, loop_end; /* GcAdpcmFormat.Looping / LoopStart / LoopEnd */
    int samples_per_interleave;        /* DspConfiguration.SamplesPerInterleave (default 0x3800, divisible by 14) */
    int loop_point_alignment;          /* DspConfiguration.LoopPointAlignment (default 1) */
    int trim_file;                     /* Configuration.TrimFile (default true) */
} vga_dsp_params;
typedef struct {
    int sample_count, loop_start, loop_end;            /* as the header carries them (DspWriter.cs:22,29-31) */
    int start_addr, end_addr, cur_addr;                /* :33-35, nibble addresses */
    int bytes_per_interleave, frames_per_interleave;   /* :25-27 */
    int audio_data_size, file_size;                    /* :99-100, :18 */
} vga_dsp_layout;
/* size math only; VGA_ERR_OUT_OF_RANGE like DspConfiguration's setters (DspConfiguration.cs:31-46) */
int vga_dsp_layout_for(const vga_dsp_params *p, int nch, vga_dsp_layout *out);
/* adpcm[c]: GcAdpcmChannel.GetAdpcmAudio(), adpcm_len bytes each (equal lengths, Interleave.cs:49-50);
 * coefs nch*16; gain nch or NULL (0); start_context / loop_context nch*3 (pred/scale, hist1, hist2) or NULL
 * (start: (Adpcm[0], 0, 0) as GcAdpcmChannel.cs:45 builds it; loop: zeros).  file_out: layout.file_size bytes. */
int vga_dsp_write(const uint8_t *const *adpcm, int adpcm_len, const int16_t *coefs, const int16_t *gain,
                  const int16_t *start_context, const int16_t *loop_context, int nch, const vga_dsp_params *p,
                  uint8_t *file_out);
/* device-resident variant: the image is assembled in HBM (d_file 8-byte aligned, layout.file_size bytes) */
int vga_dsp_write_device(const uint8_t *d_adpcm, int64_t adpcm_pitch, int adpcm_len, const int16_t *d_coefs,
                         const int16_t *d_gain, const int16_t *d_start_context, const int16_t *d_loop_context,
                         int nch, const vga_dsp_params *p, uint8_t *d_file, void *stream);

/* ======================================================================
 * CRI ADX
 * ====================================================================== */

/* VGAudio/Codecs/CriAdx/CriAdxParameters.cs:5-12 (same fields, same defaults via
 * vga_adx_default_params).  type: 2 Fixed, 3 Linear, 4 Exponential (CriAdxType.cs). */
typedef struct {
    int sample_rate;        /* 48000 */
    int highpass_frequency; /* 500 */
    int frame_size;         /* 18 */
    int version;            /* 4 */
    int16_t history;        /* decoder start history (CriAdxCodec.cs:16-17) */
    int padding;            /* alignment samples (CriAdxFormat.cs:62) */
    int type;               /* 3 */
    int filter;             /* Fixed type only: 0..3 */
} vga_adx_params;

void vga_adx_default_params(vga_adx_params *p);
/* CriAdxCodec.CalculateCoefficients (CriAdxCodec.cs:173-184) */
int vga_adx_calculate_coefficients(int highpass_freq, int sample_rate, int16_t *coefs_out /*[2]*/);
/* VGAudio/Formats/CriAdx/CriAdxHelpers.cs:7-31 */
int vga_adx_nibble_count_to_sample_count(int nibble_count, int frame_size);
int vga_adx_sample_count_to_nibble_count(int sample_count, int frame_size);
int vga_adx_sample_count_to_byte_count(int sample_count, int frame_size);
/* bytes CriAdxCodec.Encode allocates: ceil((pcm_length + padding) / samplesPerFrame) * FrameSize
 * (CriAdxCodec.cs:59-66); negative on invalid parameters */
int vga_adx_encoded_byte_count(int pcm_length, const vga_adx_params *p);

/* Replaces the Parallel.For body of CriAdxFormat.EncodeFromPcm16 (Formats/CriAdx/CriAdxFormat.cs:67-81
 * -> CriAdxCodec.Encode, CriAdxCodec.cs:56-105).  pcm[c]: pcm_length shorts; out[c]:
 * vga_adx_encoded_byte_count bytes; history_out[c] (may be NULL) receives channelConfig.History
 * (= pcm[c][0] for version 4 without padding, CriAdxCodec.cs:73).  Empty PCM with version 4 and
 * no padding -> VGA_ERR_ARGUMENT (the reference reads pcm[0]). */
int vga_adx_encode_batch(const int16_t *const *pcm, int nch, int pcm_length, const vga_adx_params *p,
                         uint8_t *const *out, int16_t *history_out);
/* Replaces the Parallel.For body of CriAdxFormat.ToPcm16 (CriAdxFormat.cs:37-49 -> CriAdxCodec.Decode,
 * CriAdxCodec.cs:9-54).  adpcm[c]: adpcm_length bytes each.  A stream shorter than the decoder
 * reads, or a frame naming a filter outside the coefficient table, -> VGA_ERR_ARGUMENT. */
int vga_adx_decode_batch(const uint8_t *const *adpcm, int adpcm_length, int nch, int sample_count,
                         const vga_adx_params *p, int16_t *const *pcm_out);
/* Ragged batches: the channels of many files in one call (VGAudio.Cli/Batch.cs:24-25 runs a worker per file, each through
 * CriAdxFormat.EncodeFromPcm16 / ToPcm16), channel c with its own length and its own parameters (params: nch entries; a
 * file's sample rate sets its high-pass coefficients, CriAdxCodec.cs:64).  out[c]: vga_adx_encoded_byte_count(
 * pcm_lengths[c], &params[c]) bytes; results are byte for byte those of one call per channel.  Channels are grouped by
 * parameter set and length (buckets a quarter wide, zero-padded on the device: the encoder is causal and pads its last
 * frame with zeros itself, CriAdxCodec.cs:78-91). */
int vga_adx_encode_batch_v(const int16_t *const *pcm, const int *pcm_lengths, int nch, const vga_adx_params *params,
                           uint8_t *const *out, int16_t *history_out);
int vga_adx_decode_batch_v(const uint8_t *const *adpcm, const int *adpcm_lengths, int nch, const int *sample_counts,
                           const vga_adx_params *params, int16_t *const *pcm_out);
/* device-resident variants (pitches: pcm in samples, bytes for ADX data and even) */
int vga_adx_encode_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, int pcm_length,
                          const vga_adx_params *p, uint8_t *d_out, int64_t out_pitch,
                          int16_t *d_history_out, void *stream);
int vga_adx_decode_device(const uint8_t *d_adpcm, int64_t in_pitch, int adpcm_length, int nch,
                          int sample_count, const vga_adx_params *p, int16_t *d_pcm,
                          int64_t pcm_pitch, int *d_status, void *stream);

/* ----------------------------------------------------------------------
 * ADX container: VGAudio/Containers/Adx/AdxWriter.cs:14-139.
 * Header (big-endian; version 4 adds the channel histories), "(c)CRI", the channels' frames interleaved one
 * frame at a time, footer 0x8001 + padding.  The reference writes the header fields one after the other
 * whatever HeaderSize is and lets "(c)CRI" and the audio overwrite what ran past it; it puts the footer where
 * the interleaver left the stream -- both are reproduced.  Encryption (EncryptionKey) is rank 4, not here.
 * -------------------------------------------------------------------- */
typedef struct {
    int sample_rate;
    int sample_count;                  /* CriAdxFormat.SampleCount (unaligned + AlignmentSamples) */
    int looping, loop_start, loop_end; /* CriAdxFormat.Looping / LoopStart / LoopEnd (aligned) */
    int alignment_samples;             /* CriAdxFormat.AlignmentSamples */
    int frame_size, version, type;     /* CriAdxFormat.FrameSize / Version / Type (2 Fixed, 3 Linear, 4 Exponential) */
    int highpass_frequency;            /* CriAdxFormat.HighpassFrequency */
    int encryption_type;               /* AdxConfiguration.EncryptionType: header byte only */
    int trim_file;                     /* Configuration.TrimFile (default true) */
} vga_adx_file_params;
typedef struct {
    int sample_count, frame_count;                     /* AdxWriter.cs:21,28 */
    int base_header_size, alignment_bytes, header_size;/* :30-31,58-69 */
    int audio_offset, audio_size;                      /* :32-33 */
    int footer_offset, footer_size;                    /* :34-35 */
    int loop_start_offset, loop_end_offset;            /* :36-37 */
    int file_size;                                     /* :18 */
} vga_adx_file_layout;
int vga_adx_file_layout_for(const vga_adx_file_params *p, int nch, vga_adx_file_layout *out);
/* audio[c]: CriAdxChannel.Audio (audio_len bytes each); history[c]: CriAdxChannel.History (needed for version 4);
 * file_out: layout.file_size bytes */
int vga_adx_write(const uint8_t *const *audio, int audio_len, const int16_t *history, int nch,
                  const vga_adx_file_params *p, uint8_t *file_out);
/* device-resident: d_audio rows audio_pitch bytes apart and d_history as vga_adx_encode_device leaves them */
int vga_adx_write_device(const uint8_t *d_audio, int64_t audio_pitch, int audio_len, const int16_t *d_history, int nch,
                         const vga_adx_file_params *p, uint8_t *d_file, void *stream);

/* ======================================================================
 * CRI HCA
 * ====================================================================== */

/* VGAudio/Codecs/CriHca/HcaInfo.cs:5-50 (the fields the codec path reads or derives) */
typedef struct {
    int channel_count, sample_rate, sample_count, frame_count;
    int inserted_samples, appended_samples, header_size, frame_size;
    int min_resolution, max_resolution, track_count, channel_config;
    int total_band_count, base_band_count, stereo_band_count, hfr_band_count;
    int bands_per_hfr_group, hfr_group_count;
    int looping, loop_start_frame, loop_end_frame, pre_loop_samples, post_loop_samples;
    int use_ath_curve, comment_length;
} vga_hca_info;

/* VGAudio/Codecs/CriHca/CriHcaParameters.cs:3-15.  quality: CriHcaQuality (0 NotSet, 1 Highest,
 * 2 High, 3 Middle, 4 Low, 5 Lowest). */
typedef struct {
    int quality, bitrate, limit_bitrate, channel_count, sample_rate, sample_count;
    int looping, loop_start, loop_end;
} vga_hca_params;

/* CriHcaEncoder.Initialize (VGAudio/Codecs/CriHca/CriHcaEncoder.cs:61-114): derives bitrate, frame
 * size, band counts, channel configuration, loop/header layout, frame count.  More than 8 channels or
 * an invalid channel mapping -> VGA_ERR_OUT_OF_RANGE (ArgumentOutOfRangeException). */
int vga_hca_encoder_initialize(const vga_hca_params *config, vga_hca_info *info_out);

/* Replaces CriHcaFormat.EncodeFromPcm16 (VGAudio/Formats/CriHca/CriHcaFormat.cs:34-84: Initialize +
 * the serial frame loop over CriHcaEncoder.Encode, CriHcaEncoder.cs:126-286) for a batch of
 * `nstreams` equally shaped streams.  pcm: nstreams*channel_count planar pointers (stream-major,
 * sample_count shorts each); frames_out[s]: frame_count*frame_size bytes (from
 * vga_hca_encoder_initialize).  "Bitrate is set too low." -> VGA_ERR_INVALID_DATA.  Looping
 * streams: the encoder's pre-audio / replayed loop audio / trailing zeros (CriHcaEncoder.cs:170-254)
 * are reproduced from the loop fields of the HcaInfo. */
int vga_hca_encode_batch(const int16_t *const *pcm, int nstreams, const vga_hca_params *config,
                         vga_hca_info *info_out, uint8_t *const *frames_out);
/* Replaces CriHcaFormat.ToPcm16 (CriHcaFormat.cs:26-32 -> CriHcaDecoder.Decode,
 * VGAudio/Codecs/CriHca/CriHcaDecoder.cs:11-192).  frames[s]: frame_count*frame_size bytes;
 * pcm_out: nstreams*channel_count pointers, info->sample_count shorts each.  Invalid sync word ->
 * VGA_ERR_INVALID_DATA ("Invalid frame header"). */
int vga_hca_decode_batch(const vga_hca_info *info, const uint8_t *const *frames, int nstreams,
                         int16_t *const *pcm_out);
/* The reference's streaming encoder object (CriHcaEncoder.InitializeNew / Encode / GetPendingFrame / PendingFrameCount /
 * FramesProcessed / FrameSize, VGAudio/Codecs/CriHca/CriHcaEncoder.cs:20-31, :46-51, :126-163): a host that produces its PCM
 * 1024 samples at a time.  vga_hca_stream_create = InitializeNew (same errors as vga_hca_encoder_initialize;
 * info_out may be NULL).  vga_hca_stream_encode = Encode(short[][] pcm, byte[] hcaOut): pcm = channel_count pointers to
 * 1024 samples each (a block that reaches past the stream's last sample is read whole, as the reference's SaveLoopAudio does,
 * :244-254); *frames_output = the number of frames the block completed -- 0 while the encoder's buffer fills, several when
 * pre-audio or post-audio flush whole frames -- the first written to hca_out (frame_size bytes), the others queued for
 * vga_hca_stream_get_pending_frame (= GetPendingFrame; VGA_ERR_INVALID_OP "There are no pending frames" when none).
 * Encode after the last frame -> VGA_ERR_INVALID_OP ("All audio frames have already been output by the encoder").  The
 * frames are byte for byte those of vga_hca_encode_batch on the same PCM.  One stream per object, calls on one object from
 * one thread at a time; the object holds the stream's PCM and frames in HBM (a 60 s stereo stream: 14 MB). */
typedef struct vga_hca_stream vga_hca_stream;
int vga_hca_stream_create(const vga_hca_params *config, vga_hca_info *info_out, vga_hca_stream **stream_out);
int vga_hca_stream_encode(vga_hca_stream *stream, const int16_t *const *pcm, uint8_t *hca_out, int *frames_output);
int vga_hca_stream_pending_frame_count(const vga_hca_stream *stream);
int vga_hca_stream_get_pending_frame(vga_hca_stream *stream, uint8_t *frame_out);
int vga_hca_stream_frames_processed(const vga_hca_stream *stream);
int vga_hca_stream_frame_size(const vga_hca_stream *stream);
void vga_hca_stream_destroy(vga_hca_stream *stream);

/* Ragged batches: streams of different shapes in one call (a worker per file, VGAudio.Cli/Batch.cs:24-25).  configs /
 * infos_out: nstreams entries; pcm: the streams' channels one after the other (sum of the channel counts pointers);
 * frames_out[s]: infos_out[s].frame_count * frame_size bytes.  Streams that differ in length only (not looping) share
 * their launches (zero-padded buckets a quarter wide: frame size and band counts follow from the bitrate,
 * CriHcaEncoder.cs:288-368, every frame is encoded on its own and the encoder's input past the PCM is silence, :234-240);
 * results are byte for byte those of one call per stream.  Decoding groups the streams by HcaInfo: streams of one shape
 * decode together, a batch of all-different lengths costs a call per stream. */
int vga_hca_encode_batch_v(const int16_t *const *pcm, int nstreams, const vga_hca_params *configs,
                           vga_hca_info *infos_out, uint8_t *const *frames_out);
int vga_hca_decode_batch_v(const vga_hca_info *infos, const uint8_t *const *frames, int nstreams,
                           int16_t *const *pcm_out);
/* device-resident variants: pcm stream s / channel c at d_pcm + s*stream_pitch + c*ch_pitch (samples);
 * frames of stream s at d_frames + s*frames_pitch (even; decode: 4-byte aligned with >= 8 bytes of
 * slack after frame_count*frame_size).  *d_status receives flag bits (16 internal: cost table; 1 bad sync, 2 bad scale-factor
 * delta, 4 bitrate too low, 8 boundary search failed). */
size_t vga_hca_decode_workspace_bytes(const vga_hca_info *info, int nstreams);
int vga_hca_encode_device(const int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch, int nstreams,
                          int pcm_length, const vga_hca_info *info, uint8_t *d_frames,
                          int64_t frames_pitch, int *d_status, void *stream);
int vga_hca_decode_device(const vga_hca_info *info, const uint8_t *d_frames, int64_t frames_pitch,
                          int nstreams, int16_t *d_pcm, int64_t stream_pitch, int64_t ch_pitch,
                          void *d_workspace, size_t workspace_bytes, int *d_status, void *stream);

/* ======================================================================
 * Synthetic PCM16 source for benchmarks/tests: integer-only,
 * counter-based; bit-identical to vgaudio_amd/synth.py.  d_params: nch x 4
 * uint32 {f_inc, phi, amp, lfo_inc}; channel ids first_channel..+nch.
 * ====================================================================== */
int vga_synth_pcm16_device(int16_t *d_pcm, int64_t pcm_pitch, int nch, int length,
                           int first_channel, const uint32_t *d_params, void *stream);

/* ----------------------------------------------------------------------
 * HCA container: VGAudio/Containers/Hca/HcaWriter.cs:12-185.
 * Chunks HCA/fmt/comp/[loop]/ciph/[rva]/pad|comm, zero padding to HeaderSize - 2, CRC-16 of the header, then the
 * frames back to back.  comment: NUL-terminated or NULL (HcaInfo.Comment; info.comment_length must have been its
 * length when the encoder sized the header); volume: HcaInfo.Volume (1 = no rva chunk); encryption_type:
 * HcaInfo.EncryptionType; encrypted_ids: non-zero when Configuration.EncryptionKey is set -- every
 * chunk id byte gets its top bit (WriteChunkId :158-171).  The frames are encrypted beforehand with vga_hca_crypt*.
 * -------------------------------------------------------------------- */
int vga_hca_file_size(const vga_hca_info *info);                              /* HcaWriter.FileSize (:22), < 0 = error */
int vga_hca_file_header(const vga_hca_info *info, const char *comment, float volume, int encryption_type,
                        int encrypted_ids, uint8_t *header_out /* info->header_size bytes */);   /* WriteHeader (:57-82); host only */
int vga_hca_write(const vga_hca_info *info, const uint8_t *frames, const char *comment, float volume,
                  int encryption_type, int encrypted_ids, uint8_t *file_out);   /* host memory; header + copy */
/* nstreams equally shaped streams: image s (file_pitch apart) = header + the frames of stream s (frames_pitch apart) */
int vga_hca_write_device(const vga_hca_info *info, const uint8_t *d_frames, int64_t frames_pitch, int nstreams,
                         const char *comment, float volume, int encryption_type, int encrypted_ids, uint8_t *d_files,
                         int64_t file_pitch, void *stream);

/* ----------------------------------------------------------------------
 * WAVE, 16-bit PCM: the step before the codec path.
 * VGAudio/Containers/Wave/WaveReader.cs:13-100 + Utilities/Riff/RiffParser.cs:36-78 (parse: host),
 * Utilities/Interleave.cs:188-207 InterleavedByteToShort / :168-186 ShortToInterleavedByte (device transposes),
 * Containers/Wave/WaveWriter.cs:12-165.  8-bit files are parsed but not converted (Pcm8 is outside this path).
 * -------------------------------------------------------------------- */
typedef struct {
    int channel_count, sample_rate, bits_per_sample;
    int sample_count;            /* per channel, from the data bytes present: the length of Pcm16Format's channels */
    int sample_count_declared;   /* from the data chunk's declared size (WaveStructure.SampleCount, :27) */
    int looping, loop_start, loop_end;   /* first smpl loop; Looping = End > Start (:32-37) */
    int64_t data_offset;         /* of the data chunk's bytes inside the file */
    int data_size, data_size_declared;
} vga_wave_info;
typedef struct { int sample_rate, sample_count, looping, loop_start, loop_end; } vga_wave_params;
/* VGA_ERR_INVALID_DATA with the reference's message for what ValidateWaveFile rejects (:70-98) and for files that
 * end inside a chunk; VGA_ERR_OUT_OF_RANGE for loop points Pcm16FormatBuilder.WithLoop rejects */
int vga_wave_parse(const uint8_t *file, int64_t file_len, vga_wave_info *out);
int vga_wave_read_pcm16(const uint8_t *file, int64_t file_len, const vga_wave_info *info, int16_t *const *pcm_out);
int vga_wave_deinterleave_pcm16_device(const uint8_t *d_data, int sample_count, int nch, int16_t *d_pcm,
                                       int64_t pcm_pitch /* samples */, void *stream);
int64_t vga_wave_file_size(const vga_wave_params *p, int nch);                 /* WaveWriter.FileSize (:25), < 0 = error */
int vga_wave_write_pcm16(const int16_t *const *pcm, int nch, const vga_wave_params *p, uint8_t *file_out);
int vga_wave_write_pcm16_device(const int16_t *d_pcm, int64_t pcm_pitch, int nch, const vga_wave_params *p,
                                uint8_t *d_file, void *stream);

/* ----------------------------------------------------------------------
 * ADX / HCA encryption passes and key derivations.
 * VGAudio/Codecs/CriAdx/CriAdxKey.cs:10-66, CriAdxEncryption.cs:8-94;
 * VGAudio/Codecs/CriHca/CriHcaKey.cs:8-181, CriHcaEncryption.cs:12-33.
 * The known-key lists (CriAdxEncryptionKeys.cs, CriHcaEncryptionKeys.cs) stay with the caller: find_key takes
 * the candidates.  The reference has no tests for any of this (parity unpinned by reference vectors).
 * -------------------------------------------------------------------- */
typedef struct { int seed, mult, inc; } vga_adx_key;                     /* CriAdxKey.Seed / Mult / Inc */
int vga_adx_key_from_code(uint64_t key_code, vga_adx_key *out);          /* CriAdxKey(ulong) */
int vga_adx_key_from_string(const char *key_string, vga_adx_key *out);   /* CriAdxKey(string), ASCII */
uint64_t vga_adx_key_code(const vga_adx_key *key);                       /* CriAdxKey.KeyCode */
/* EncryptDecrypt (its own inverse for type 8; type 9 also masks the first header byte), in place.
 * audio_len must be whole frames. */
int vga_adx_crypt(uint8_t *const *audio, int audio_len, int nch, const vga_adx_key *key, int encryption_type,
                  int frame_size);
int vga_adx_crypt_device(uint8_t *d_audio, int64_t audio_pitch, int audio_len, int nch, const vga_adx_key *key,
                         int encryption_type, int frame_size, void *stream);
/* FindKey over `keys` (host array): *index_out = first candidate every frame header agrees with, or -1 */
int vga_adx_find_key_device(const uint8_t *d_audio, int64_t audio_pitch, int audio_len, int nch, int encryption_type,
                            int frame_size, const vga_adx_key *keys, int nkeys, int *index_out, void *stream);
/* The brute-force key search of the reference's `crackadx` tool for one file (VGAudio.Tools/CrackAdx/GuessAdx.cs:118-218:
 * Run / TryScale / FindStartingKey / KeyIsValid).  scales = the file's big-endian 16-bit frame headers in stream order
 * (AdxFile.Scales), start_frame = the frame holding the file's first non-zero byte (AdxFile.StartFrame).  mults / incs:
 * candidate lists, NULL = the reference's sets for the type (vga_adx_guess_default_candidates: 0x400 primes each for
 * type 8; 2048 / 4096 values for type 9).  keys_out receives the keys every scale agrees with, sorted by
 * (seed, mult, inc), without duplicates; the confidence report (re-encode and diff) is the caller's
 * (vga_adx_decode_batch + vga_adx_encode_batch). */
int vga_adx_guess_default_candidates(int encryption_type, int *mults, int *nmult, int *incs, int *ninc);
int vga_adx_guess_keys(const uint16_t *scales, int nscales, int start_frame, int encryption_type, const int *mults, int nmult,
                       const int *incs, int ninc, vga_adx_key *keys_out, int max_keys, int *nkeys_out);
/* CriHcaKey: key_type 56 = CriHcaKey(ulong keyCode), 0 / 1 = CriHcaKey(Type); both tables are 256 bytes */
int vga_hca_key_tables(int key_type, uint64_t key_code, uint8_t *decryption_table, uint8_t *encryption_table);
/* Crypt: substitute the first FrameSize - 2 bytes of every frame, refresh its CRC-16; table in host memory */
int vga_hca_crypt(uint8_t *frames, int frame_count, int frame_size, const uint8_t *table);
int vga_hca_crypt_device(uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count, int frame_size,
                         const uint8_t *table, void *stream);
/* CriHcaEncryption.FindKey / TestKey (CriHcaEncryption.cs:34-88) over caller-supplied candidates: decryption_tables =
 * nkeys x 256 bytes (host memory).  *index_out = the first key under which the first ten non-empty frames unpack
 * (CriHcaPacking.UnpackFrame), or -1.  VGA_ERR_INVALID_DATA: a frame's sync word is wrong (InvalidDataException). */
int vga_hca_find_key(const vga_hca_info *info, const uint8_t *frames, int frame_count, const uint8_t *decryption_tables,
                     int nkeys, int *index_out);
int vga_hca_find_key_device(const vga_hca_info *info, const uint8_t *d_frames, int frame_count,
                            const uint8_t *decryption_tables, int nkeys, int *index_out, void *stream);
/* The statistics the reference's `crackhca` analysis starts from (VGAudio.Tools/CrackHca/Crack.cs:43-80): how often each
 * byte value occurs at each of the first `positions` (<= 64; the tool uses 30) bytes of the frames.  counts_out:
 * positions x 256 uint32 in host memory.  The table solver that follows (Solver.cs, Table.cs) is interactive analysis
 * on those 30 x 256 numbers and stays with the tool. */
int vga_hca_byte_position_counts_device(const uint8_t *d_frames, int64_t frames_pitch, int nstreams, int frame_count,
                                        int frame_size, int positions, uint32_t *counts_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif
