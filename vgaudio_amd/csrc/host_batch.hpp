// host_batch.hpp -- glue between the host-pointer entry points of the C ABI and host_pipeline.hpp.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include "common.hpp"
#include "host_pipeline.hpp"

namespace vga {

// Per-thread overrides of the pipeline's shape (vga_testing_host_pipeline_this_thread): the tests force many feeders,
// one-row slots and small chunks on small inputs so that every hand-off of the pipeline is exercised on the GPU box.
struct PipeOverride {
    int feeders = 0, drainers = 0, chunk_units = 0, slot_bytes = 0, tail_units = 0;   // slot_bytes < 0: direct copies (no staging ring); > 0: staged
    int buckets_order = 0;                   // plan_buckets: 0 = the entry point's own order, 1 = shortest chunks first, 2 = longest first
    int transfer = 0;                        // direct rows: 0 = by transfer kernels on reserved compute units (round 6), 1 = one copy per row
    int compute_lanes = 0;                   // > 0: that many compute streams (chunk k on stream k % lanes) whatever the entry point asked for
};
// transfer_kernels.hip
int launch_transfer(const pipe::Job::TransferPiece *pieces, int n, hipStream_t stream);
int device_cu_count();                               // common.hpp
PipeOverride &pipe_override();                       // capi_gcadpcm.hip
// the calling thread's last pipeline run, plus what the entry point spent around it (device allocation, small copies)
struct PipeReport { pipe::Stats stats; double t_alloc = 0, t_entry = 0; };
PipeReport &pipe_report();                           // capi_gcadpcm.hip
int hardware_queues_requested();                     // capi_gcadpcm.hip: GPU_MAX_HW_QUEUES as this process will see it (4 when unset)

// vga_set_devices(): the GPUs the host-pointer entry points spread one call's units over (empty: the calling thread's
// current device, nothing is spread) -- capi_gcadpcm.hip
std::vector<int> batch_devices();
// the calling thread's test hooks (include/vgaudio_hip_testing.h), handed on to the threads that run a call's other shares
struct ThreadHooks { int force_open_seams, encoder_layout, coefs_variant, encoder_segments, hca_frames_per_group; PipeOverride pipe; int encoder_persistent = 0; };
ThreadHooks capture_thread_hooks();                  // capi_gcadpcm.hip
void apply_thread_hooks(const ThreadHooks &h);       // capi_gcadpcm.hip

// vga_set_progress_callback(): the calling thread's callback, and the state of one call's reports -- shared by the call's
// device shares, whose pipelines report chunks from their own worker threads
struct ProgressCallback { void (*fn)(void *user, int64_t done, int64_t total) = nullptr; void *user = nullptr; };
ProgressCallback progress_callback();                // capi_gcadpcm.hip (thread-local)
struct ProgressSink {
    ProgressCallback cb;
    int64_t total = 0, done = 0;
    std::mutex m;
    void add(int64_t units)
    {
        std::lock_guard<std::mutex> g(m);            // one report at a time, whatever thread it comes from
        done += units;
        cb.fn(cb.user, done, total);
    }
};
ProgressSink *&current_progress_sink();              // capi_gcadpcm.hip (thread-local): the call this thread works for

// Runs body(first_unit, unit_count) -- the single-device form of an entry point, which reports failures through
// set_error() + its status code -- once per share of `units` over vga_set_devices()'s list (pipe::run_on_devices: a host
// thread and a whole pipeline per device; results land in the caller's rows, no collective).  Shares are at least
// `min_units` units, so small calls stay on one GPU.
template <class Body>
inline int for_each_device_share(int units, int min_units, Body &&body)
{
    ProgressSink sink;
    sink.cb = progress_callback();
    sink.total = units;
    struct SinkScope {                                 // the pipelines this thread starts report to `sink`
        ProgressSink *before;
        explicit SinkScope(ProgressSink *s) : before(current_progress_sink()) { current_progress_sink() = s; }
        ~SinkScope() { current_progress_sink() = before; }
    } scope(sink.cb.fn && units > 0 ? &sink : nullptr);
    const std::vector<int> devices = batch_devices();
    if (devices.empty() || units <= 0) return body(0, units);
    const std::vector<pipe::Share> shares = pipe::plan_shares(devices, units, min_units);
    const ThreadHooks hooks = capture_thread_hooks();
    if (shares.size() == 1) {                          // one GPU, but the listed one
        int before = 0;
        VGA_HIP_TRY(hipGetDevice(&before));
        VGA_HIP_TRY(hipSetDevice(shares[0].device));
        const int rc = body(0, units);
        (void)hipSetDevice(before);
        return rc;
    }
    ProgressSink *const shared_sink = current_progress_sink();
    const pipe::Result r = pipe::run_on_devices(shares, [&](const pipe::Share &sh, std::string &why) -> int {
        if (sh.index != 0) {
            apply_thread_hooks(hooks);
            current_progress_sink() = shared_sink;
        }
        const int rc = body(sh.first, sh.count);
        if (rc) why = vga_last_error();
        return rc;
    });
    if (r.code) set_error("%s", r.why.c_str());
    return r.code;
}

// bytes a job uploads / downloads (ragged jobs: the sum of their rows)
inline size_t job_in_total(const pipe::Job &job)
{
    if (!job.in_rows) return 0;
    const size_t rows = (size_t)job.units * job.in_rows_per_unit;
    if (!job.in_row_sizes) return rows * job.in_row_bytes;
    size_t t = 0;
    for (size_t r = 0; r < rows; r++) t += job.in_row_sizes[r];
    return t;
}
inline size_t job_out_total(const pipe::Job &job)
{
    if (!job.out_rows) return 0;
    const size_t rows = (size_t)job.units * job.out_rows_per_unit;
    if (!job.out_row_sizes) return rows * job.out_row_bytes;
    size_t t = 0;
    for (size_t r = 0; r < rows; r++) t += job.out_row_sizes[r];
    return t;
}

// Units per chunk run_batch_pipeline() will use for this job (callers size per-chunk scratch with it).
inline int planned_chunk_units(const pipe::Job &job, int default_chunk_units)
{
    const PipeOverride &o = pipe_override();
    const size_t in_total = job_in_total(job);
    const size_t out_total = job_out_total(job);
    int chunk = o.chunk_units > 0 ? o.chunk_units : default_chunk_units;
    // a call that downloads more than it uploads (a decode) is bound by the download, which cannot start before the first
    // chunk's kernels have run: quarter chunks there (GC decode, 4096 x 60 s: the first download started at 87 ms)
    if (o.chunk_units <= 0 && out_total > in_total) chunk = std::max(1, chunk / 4);
    // small batches: one chunk (a chunk boundary only pays when the upload of the next chunk is worth hiding)
    if (o.chunk_units <= 0 && in_total + out_total < ((size_t)256 << 20)) chunk = job.units;
    return std::max(1, std::min(chunk, job.units));
}

// Shapes the job (workers by volume, chunk size) and runs it; a failure becomes set_error() + its status code.
inline int run_batch_pipeline(pipe::Job &job, int default_chunk_units)
{
    const PipeOverride &o = pipe_override();
    // set-up done on the legacy stream (hipMemset of status words, slack bytes) must have run: the pipeline's streams
    // are non-blocking and do not order themselves behind it
    VGA_HIP_TRY(hipStreamSynchronize(nullptr));
    int device = 0;
    VGA_HIP_TRY(hipGetDevice(&device));
    job.device = device;
    const size_t in_total = job_in_total(job);
    const size_t out_total = job_out_total(job);
    // a ragged job's typical row decides between page-locking rows one by one and the staging ring
    const size_t in_row_typical = job.in_row_sizes ? in_total / std::max<size_t>(1, (size_t)job.units * job.in_rows_per_unit) : job.in_row_bytes;
    const size_t out_row_typical = job.out_row_sizes ? out_total / std::max<size_t>(1, (size_t)job.units * job.out_rows_per_unit) : job.out_row_bytes;
    // Measured on the MI355X box (tools/bench_h2d_modes.hip, tools/bench_overlap.hip, tools/sweep_host_pipeline.py;
    // profiles/r02_*): page-locking the caller's rows for the call (hipHostRegister) lets ONE stream of direct copies
    // run at the link's rate (~56 GB/s) on the DMA engines next to the kernels; copies issued on FOUR streams at once
    // were held back until the kernels ended (two: measured again in round 5, below), and a ring filled by memcpy threads was no faster than its four threads.
    // Downloads are rows of a megabyte or two; issued one by one next to two lanes of kernels they were left waiting
    // until the kernels ended (118 ms after the last kernel), so they go through the page-locked ring in 32 MB copies and
    // two drainer threads hand the rows out.  So: one feeder with direct uploads, two drainers behind a ring, all
    // uploads on one stream and all downloads on another (a slot = the rows a worker takes at a time); slot_bytes < 0
    // (testing hook) makes both directions direct, > 0 both staged.
    if (o.compute_lanes > 0) job.compute_lanes = o.compute_lanes;
    job.feeders = o.feeders > 0 ? o.feeders : (o.feeders < 0 ? -o.feeders : 1);   // (test hook, negative: that many feeders, a stream each)
    job.drainers = o.drainers > 0 ? o.drainers : (out_total >= ((size_t)256 << 20) ? 2 : 1);
    // rows worth page-locking one by one: from 256 KB on (smaller rows are cheap to copy into the ring)
    job.direct = o.slot_bytes < 0 || (o.slot_bytes == 0 && in_row_typical >= ((size_t)256 << 10));
    // downloads: through the ring when they are the smaller direction (an encode), direct into page-locked caller rows
    // when they are the larger one (a decode: 23.6 GB of PCM through two memcpy threads would be the bottleneck;
    // tools/time_decode_batches.py: ADX 582 -> 527 ms, HCA 323 -> 271 ms)
    job.direct_out = o.slot_bytes < 0 || (o.slot_bytes == 0 && out_row_typical >= ((size_t)256 << 10) && out_total > in_total);
    job.shared_streams = o.feeders >= 0;
    // Ragged jobs pay ~11 us of idle copy engine per row (10 008 files, 23.6 GB: 528 ms on the one stream against 412 ms at
    // the link's rate).  Two feeders with a stream each (the test hook's negative feeder count) were measured in round 5
    // (tools/time_ragged_host.py --feeders 1 0 -2 -3, profiles/r05_t_upload_streams.log): the upload then takes EITHER 447-485 ms
    // OR 535-545 ms, call by call in one process (ADX: 4 calls of 6 slow, HCA 2 of 6), and one call inside bench.py took 809 ms
    // -- against 528 +- 3 ms on one stream.  Not used: a steady 600 ms beats 545-620 with outliers.
    job.slot_bytes = o.slot_bytes > 0 ? (size_t)o.slot_bytes : (o.slot_bytes < -1 ? (size_t)(-o.slot_bytes) : (size_t)32 << 20);
    job.chunk_units = planned_chunk_units(job, default_chunk_units);
    // the last chunk is split once, into (5/8, 3/8) of a chunk: the first part's kernels end about when the second part's
    // upload does, so the tail of the call is one short chunk's kernels running alone (measured at 4096 x 60 s, tail =
    // 192 / 256 / 320 / 384 / 448 / 512 of 1024: 555 / 545 / 539 / 519 / 528 / 529 ms; halving down to 1/8: 533 ms --
    // short chunks that overlap slow each other down)
    job.tail_units = in_total + out_total >= ((size_t)256 << 20) ? std::max(1, job.chunk_units * 3 / 8) : 0;   // small calls: one chunk
    if (o.tail_units > 0) job.tail_units = o.tail_units;
    // A short FIRST chunk (job.head_units: a quarter chunk starts the first kernel 29 ms instead of 115 ms into a
    // 4096 x 60 s call) was measured and is not used: the call is bound by its upload (461 of ~530 ms) and ends one short
    // chunk's kernels + download after it, whenever the first kernel started (profiles/r03_b_pipeline_timeline_head_chunk.log:
    // 539 and 550 ms against 518-538 ms without).
    job.head_units = 0;
    // Round 6: page-locked rows travel by transfer kernels (transfer_kernels.hip) on sixteen compute units of their own -- a
    // ragged call's 10 008 uploads no longer leave the copy engine idle between rows.  Calls of at least 256 MB only: below,
    // the rows are few and the masked streams' two rounds of persistent workgroups cost more than the copies' gaps.
    // And only where the rows are many for their bytes: the copy engine loses ~11 us per row, the kernels ~30 ms per call
    // (sixteen compute units less for the call's own kernels, the link shared with the scatter) -- the 4096 rows of 5.8 MB of
    // configs[1]'s equal-length call are better off with a copy each (519-535 ms against 590), the 10 008 files of a ragged
    // batch (2.4 MB on average) with the kernels (597 -> 556 ms): rows under 4 MB on average.
    if (o.transfer == 0 && in_total + out_total >= ((size_t)256 << 20) && in_row_typical < ((size_t)4 << 20)) {
        job.transfer = [](const pipe::Job::TransferPiece *pieces, int n, hipStream_t s, std::string &why) -> int {
            const int rc = launch_transfer(pieces, n, s);
            if (rc) why = vga_last_error();
            return rc;
        };
        job.gather_in = job.direct && job.feeders == 1;
        // downloads too, whichever way they go: a staged slot's copy (the ring is page-locked already) or, for a decode, the
        // caller's page-locked rows.  (Left to hipMemcpyAsync next to the gather, chunk 0 of a ragged encode was back after
        // 410 ms instead of 305: profiles/r06_j_pipeline_timeline_transfer.log.  Page-locking the 10 008 OUTPUT rows of an
        // encode as well so that they can be written directly costs ~100 ms of hipHostRegister / hipHostUnregister.)
        job.scatter_out = true;
        // with the upload a fifth shorter, two threads copying the ring's slots out to the caller's rows (3.4 GB each for
        // the ragged GC call) end 100 ms after the last kernel: four of them
        if (o.drainers == 0 && !job.direct_out && out_total >= ((size_t)1 << 30)) job.drainers = 4;
        job.total_cus = device_cu_count();
        job.transfer_cus = 16;
    }
    if (ProgressSink *sink = current_progress_sink())  // vga_set_progress_callback(): one report per chunk
        job.chunk_done = [sink](int, int count) { sink->add(count); };
    const pipe::Result r = pipe::run(job);
    pipe_report().stats = r.stats;
    if (r.code) {
        set_error("%s", r.why.c_str());
        return r.code;
    }
    return VGA_OK;
}

// ---------------------------------------------------------------- ragged calls by length buckets
// The `_v` entry points of codecs whose kernels take one length per launch (ADX, HCA): units are sorted by (parameter
// group, length) and cut into chunks whose lengths differ by at most a quarter; a chunk's rows are zero-padded on the
// device to the chunk's longest and the kernels run once per chunk.  Every unit's output is the prefix it would get alone:
// the encoders are causal (a frame depends on the samples up to its end and on the frames before it) and the
// reference pads a last partial frame with zeros itself (CriAdxCodec.cs:78-91; CriHcaEncoder.cs:234-240).
struct BucketPlan {
    std::vector<int> order;          // position -> the caller's unit
    std::vector<int> chunk_begin;    // positions: chunk k = [chunk_begin[k], chunk_begin[k + 1])
    std::vector<int> chunk_length;   // the chunk's largest length
    std::vector<int> chunk_group;    // the chunk's parameter group
    int chunk_of(int first) const { return (int)(std::upper_bound(chunk_begin.begin(), chunk_begin.end(), first) - chunk_begin.begin()) - 1; }
};
// group[i]: units of different groups never share a chunk; max_units / max_volume (sum of padded lengths) bound a chunk.
// longest_first: the order the chunks run in (inside a chunk the units stay in ascending order, its longest last).  A call is
// as long as its upload plus what the last chunk still has to compute and hand back, so the smallest chunk should come last
// -- as long as the downloads keep up: measured on bench.py's 10 008 files (tools/time_ragged_host.py,
// profiles/r05_q_ragged_host_orders.log), an HCA call (output a tenth of its input) gains 14 ms of 559 that way, but an ADX
// call (output 0.28 of its input) loses 135 ms of 592: while gigabyte chunks of uploads are queued, the downloads of the first
// big chunks crawl (1.35 GB in 410 ms) and everything is handed back after the upload has ended.
inline BucketPlan plan_buckets(const std::vector<int> &group, const std::vector<int> &length, int max_units, int64_t max_volume,
                               bool longest_first)
{
    BucketPlan b;
    const int n = (int)group.size();
    b.order.resize(n);
    for (int i = 0; i < n; i++) b.order[i] = i;
    std::stable_sort(b.order.begin(), b.order.end(), [&](int x, int y) {
        return group[x] != group[y] ? group[x] < group[y] : length[x] < length[y];
    });
    const PipeOverride &o = pipe_override();
    if (o.chunk_units > 0) max_units = o.chunk_units;
    int first = 0;
    for (int i = 0; i <= n; i++) {
        bool cut = i == n;
        if (!cut && i > first) {
            const int u = b.order[i], f = b.order[first];
            cut = group[u] != group[f] || (int64_t)length[u] > (int64_t)length[f] + length[f] / 4 + 1024 || i - first >= max_units ||
                  (int64_t)(i - first + 1) * std::max(length[u], 1) > max_volume;
        }
        if (cut && i > first) {
            b.chunk_begin.push_back(first);
            b.chunk_length.push_back(length[b.order[i - 1]]);
            b.chunk_group.push_back(group[b.order[first]]);
            first = i;
        }
    }
    b.chunk_begin.push_back(n);
    if (o.buckets_order ? o.buckets_order == 1 : !longest_first) return b;
    BucketPlan r;
    const int chunks = (int)b.chunk_length.size();
    r.order.reserve(n);
    for (int k = chunks - 1; k >= 0; k--) {
        r.chunk_begin.push_back((int)r.order.size());
        r.chunk_length.push_back(b.chunk_length[k]);
        r.chunk_group.push_back(b.chunk_group[k]);
        r.order.insert(r.order.end(), b.order.begin() + b.chunk_begin[k], b.order.begin() + b.chunk_begin[k + 1]);
    }
    r.chunk_begin.push_back(n);
    return r;
}

}  // namespace vga
