/*
 * vgaudio_hip_testing.h -- test-only exports of libvgaudio_hip.so.  NOT part of the drop-in surface
 * (include/vgaudio_hip.h); nothing a managed shim binds lives here.
 */
#ifndef VGAUDIO_HIP_TESTING_H
#define VGAUDIO_HIP_TESTING_H

#ifdef __cplusplus
extern "C" {
#endif

/* GC-ADPCM encode/decode and ADX encode/decode cut long channels into time segments that run side by side and
 * close the seams afterwards (LABNOTES.md 4.3).  mode = 1: calls made FROM THE CALLING THREAD never accept a seam
 * as closed; 2: only the even seams of every third channel are kept open (a mix of open and closed seams inside
 * one workgroup); 3: every seam, and the decoders count them as seams that would not close, so that their REPAIR launch
 * (one piece per wave from the first open seam on; round 5) produces the output instead of the chained tail kernel;
 * 0: normal operation.  The serial fall-backs then produce the output, which must not change.
 * The setting is thread-local (no other thread's calls see it) and is passed to the kernels as a launch argument.
 * Returns the calling thread's previous mode. */
int vga_testing_force_open_seams_this_thread(int mode);

/* GC-ADPCM encoder wave layout for calls made FROM THE CALLING THREAD: 0 = the launcher's choice (the product: 4 for
 * batches below ~512 channels, 8 from there on and for ragged batches); 8 = lane per (channel, predictor); 4 = lane per
 * (channel, predictor, scale candidate).  Both produce the same bytes.  Returns the previous value; other arguments leave
 * it unchanged. */
int vga_testing_gc_encoder_layout_this_thread(int channels_per_wave);
/* GC-ADPCM coefficient-search kernel for calls made from the calling thread: 0 = the launcher's choice by channel count
 * (the product), 1 = one wave per channel, 2 = workgroups of four channels and a summing wave, 3 = the same five waves on
 * one channel (the choice for small batches).  Same coefficients.
 * Returns the previous value; other arguments leave it unchanged. */
int vga_testing_gc_coefs_variant_this_thread(int variant);
/* Forces the number of time pieces a channel is cut into by the kernels that speculate over time -- the GC-ADPCM and ADX
 * encoders (pieces of 64 frames or more) and decoders (8 frames or more) -- for calls made from the calling thread (0 = the
 * launcher's own choice).  Results must not depend on it.  Returns the previous value. */
int vga_testing_gc_encoder_segments_this_thread(int segments);
/* How the GC-ADPCM encoder hands its (channel group, time piece) items to workgroups, for calls made from the calling
 * thread: 0 = the launcher's choice, 1 = one workgroup per item (a plain grid), 2 = persistent workgroups taking items from
 * a queue.  Results must not depend on it.  Returns the previous value. */
int vga_testing_gc_encoder_persistent_this_thread(int mode);

/* The GC-ADPCM encoder's piece schedule (gc::plan_encode_pieces, vgaudio_amd/csrc/gc_encode_kernel.hip) for a device of
 * `cus` compute units: `groups` channel groups of sixteen whose longest channel has `frames` frames and whose longest
 * channels hold `group_frames` frames in all.  Host arithmetic, needs no GPU.  out5 = {pieces, big, nb, small, persistent}:
 * piece k begins at frame min(k, nb) * big + max(k - nb, 0) * small.  The pieces must cover `frames`.  Returns 0. */
int vga_testing_gc_plan_pieces(int cus, int groups, int frames, long long group_frames, int ragged, int *out5);

/* Diagnostics of the GC-ADPCM encoder's data-dependent parts, summed over every launch of the process on the current
 * device since the last reset (the call synchronises the device first): out8 = {seams closed inside their piece, seams left
 * open for the chain kernel, frames re-encoded by seam runs, wave-frames encoded, wave-frames that took the cold block
 * (third trips of the retry loop, GcAdpcmEncoder.cs:127-170; counted by -DVGA_GC_STATS builds only), channels the chain kernel
 * walked, pieces encoded, 1 if this build counts the cold blocks}.
 * reset != 0 clears the counters afterwards.  Returns 0, or -1 when the device cannot be read. */
int vga_testing_gc_encode_stats(unsigned long long *out8, int reset);

/* The host-pointer entry points (vga_*_batch) move data through a pipeline of feeder threads, pinned rings, per-chunk
 * kernel launches and drainer threads (vgaudio_amd/csrc/host_pipeline.hpp); its shape normally follows the volume of
 * the call.  Non-zero arguments override it for calls made FROM THE CALLING THREAD (0 = automatic): feeder / drainer
 * thread counts (a NEGATIVE feeder count: that many feeders with an upload stream each instead of one shared stream, and a
 * download stream per drainer), units (channels, streams) per chunk, bytes per ring slot.  Results must not depend on any of
 * them. */
void vga_testing_host_pipeline_this_thread(int feeders, int drainers, int chunk_units, int slot_bytes);

/* ... and the size of the call's LAST chunk (the last regular chunk is split once into (rest, tail); 0 = automatic: 3/8 of a
 * chunk). */
void vga_testing_host_pipeline_tail_this_thread(int tail_units);

/* The ragged ADX / HCA entry points (vga_adx_*_batch_v, vga_hca_encode_batch_v) sort their units into chunks of one parameter
 * group and similar length (vgaudio_amd/csrc/host_batch.hpp, plan_buckets) and run the chunks shortest first (ADX) or longest
 * first (HCA).  vga_testing_plan_buckets() returns that plan for n units -- order_out[n]: position -> unit;
 * chunk_begin_out[chunks + 1]: positions; chunk_length_out / chunk_group_out[chunks]: a chunk's largest length and its group --
 * and the number of chunks (-1: bad arguments or more than max_chunks).  Host code, no GPU.
 * vga_testing_buckets_order_this_thread(1 | 2) forces shortest / longest first for calls made from the calling thread
 * (0: the entry point's own order; timing comparisons -- results must not depend on it). */
int vga_testing_plan_buckets(const int *group, const int *length, int n, int max_units, long long max_volume, int longest_first,
                             int *order_out, int *chunk_begin_out, int *chunk_length_out, int *chunk_group_out, int max_chunks);
void vga_testing_buckets_order_this_thread(int order);
/* Page-locked rows of the host-pointer entry points: 0 (default) = moved by transfer kernels on compute units reserved for
 * them (calls of 256 MB and more), 1 = one hipMemcpyAsync per row as until round 5 -- for calls made from the calling thread. */
void vga_testing_host_transfer_this_thread(int mode);
/* Compute streams of the host-pointer entry points' pipeline (chunk k's kernels on stream k % lanes, at most 4): 0 = the
 * entry point's own choice -- for calls made from the calling thread (timing comparisons; results must not depend on it). */
void vga_testing_host_compute_lanes_this_thread(int lanes);

/* Where the wall time of the calling thread's last pipelined call went, in seconds (diagnostics for bench.py's e2e
 * block): [0] total [1] set-up [2] feeders' memcpy (sum over threads) [3] feeders waiting for a ring slot [4] feeders
 * inside hipMemcpyAsync/hipEventRecord [5] slowest feeder [6] caller waiting for uploads [7] caller launching kernels
 * [8] caller's final stream sync [9] drainers waiting for compute [10] drainers waiting for downloads [11] drainers'
 * memcpy [12] slowest drainer [13] feeders [14] drainers [15] chunks [16] units per chunk [17] device allocation
 * before the pipeline [18] the whole entry point up to its return value [19] feeders at chunk boundaries [20] feeders' final
 * stream synchronisation [21] drainers page-locking the output rows.  Returns the number of fields. */
int vga_testing_last_pipeline_stats(double *out, int n);

/* The HCA decoder's second kernel gives a workgroup a run of consecutive frames of one stream and carries the IMDCT overlap
 * inside the run (1..16 frames, by the size of the batch).  > 0 forces the run length for calls made from the calling
 * thread (0 = the launcher's choice).  PCM must not depend on it.  Returns the previous value. */
int vga_testing_hca_frames_per_group_this_thread(int frames);

/* The HCA kernels' view of a stream (vgaudio_amd/csrc/hca_info.hpp: DeviceInfo -- channel types, coded band counts, the
 * scaled ATH curve), as the library derives it from an HcaInfo; `out` receives sizeof(DeviceInfo) = 240 bytes.  Host code,
 * needs no GPU: the CPU suite feeds it to the lane emulator of the HCA decoder (tests/host/hca_decode_emulator.cpp). */
int vga_testing_hca_device_info(const void *hca_info, void *out, int out_bytes);

#ifdef __cplusplus
}
#endif
#endif
