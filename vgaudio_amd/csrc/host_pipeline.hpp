// host_pipeline.hpp -- what a P/Invoke caller actually gets: the host-pointer entry points of the C ABI as a
// three-stage pipeline over CHUNKS of units (caller memory -> HBM -> kernels -> HBM -> caller memory).
//
// The reference fans channels out with Parallel.For over managed arrays (Formats/GcAdpcm/GcAdpcmFormat.cs:65-68,
// Formats/CriAdx/CriAdxFormat.cs:67-81) -- arrays that are pinned for the garbage collector but PAGEABLE for a DMA engine.
// One hipMemcpyAsync per channel from such memory goes through the runtime's single staging thread (round 1:
// 8-11 Gsamples/s through the ABI against 52 Gsamples/s in HBM).  What run_batch_pipeline() (host_batch.hpp) sets up since
// round 3, and what run() below does by default (Job::direct, Job::register_rows):
//   * ONE feeder thread page-locks the caller's rows where they lie (hipHostRegister, for the duration of the call) and
//     moves them to HBM chunk by chunk -- round 6: with ONE gather kernel per chunk that pulls the rows over PCIe from a
//     table of (page-locked row piece, device address) on a stream whose CU mask keeps a few compute units free of the
//     call's other kernels (Job::gather_in / transfer_cus: 57 GB/s against 45 GB/s for one hipMemcpyAsync per row, whose
//     copy engine idles ~11 us per row; tools/bench_h2d_gather.hip); rows that cannot be mapped take a plain copy;
//   * the calling thread launches the kernels of a chunk on one of two compute streams as soon as the chunk's upload
//     event is recorded -- chunk k computes while chunk k+1 uploads;
//   * D drainer threads wait for the chunk's compute event and bring its output rows back: through their own rings of
//     page-locked slots (32 MB copies, then memcpy into the caller's rows: encodes, whose output is the smaller
//     direction), or straight into the caller's page-locked rows (decodes; Job::direct_out).
// The staged mode of rounds 1-2 (F feeder threads filling page-locked ring slots with memcpy, one contiguous copy per
// slot) is still here for rows too small to be worth page-locking (Job::direct = false: rows under 256 KB).
// Page-locked memory comes from a process-wide pool (hipHostMalloc costs ~0.3 s per GB; the rings are a few hundred MB
// and are reused by later calls).  No caller pointer is retained past return.
//
// Host-only code (no kernels): tests/host/test_host_pipeline.cpp compiles this header against a mock of the few HIP
// entry points it uses and runs it under ThreadSanitizer on the CPU.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace vga {
namespace pipe {

// ---------------------------------------------------------------- page-locked staging pool (process-wide)
class PinnedPool {
public:
    static PinnedPool &get()
    {
        static PinnedPool pool;
        return pool;
    }
    // a page-locked block of at least `bytes` (nullptr on failure)
    void *acquire(size_t bytes)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            int best = -1;
            for (int i = 0; i < (int)blocks_.size(); i++)
                if (!blocks_[i].busy && blocks_[i].bytes >= bytes && (best < 0 || blocks_[i].bytes < blocks_[best].bytes)) best = i;
            if (best >= 0) {
                blocks_[best].busy = true;
                return blocks_[best].p;
            }
        }
        void *p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> g(m_);
        blocks_.push_back({p, bytes, true});
        return p;
    }
    void release(void *p)
    {
        std::lock_guard<std::mutex> g(m_);
        size_t idle = 0;
        for (auto &b : blocks_) {
            if (b.p == p) b.busy = false;
            if (!b.busy) idle += b.bytes;
        }
        // keep at most ~1 GiB parked: drop the largest idle blocks beyond that
        while (idle > kKeepBytes) {
            int big = -1;
            for (int i = 0; i < (int)blocks_.size(); i++)
                if (!blocks_[i].busy && (big < 0 || blocks_[i].bytes > blocks_[big].bytes)) big = i;
            if (big < 0) break;
            idle -= blocks_[big].bytes;
            (void)hipHostFree(blocks_[big].p);
            blocks_.erase(blocks_.begin() + big);
        }
    }
    void trim()                                            // frees every idle block (tests; library unload)
    {
        std::lock_guard<std::mutex> g(m_);
        for (int i = (int)blocks_.size() - 1; i >= 0; i--)
            if (!blocks_[i].busy) {
                (void)hipHostFree(blocks_[i].p);
                blocks_.erase(blocks_.begin() + i);
            }
    }

private:
    static constexpr size_t kKeepBytes = (size_t)1 << 30;
    struct Block { void *p; size_t bytes; bool busy; };
    std::mutex m_;
    std::vector<Block> blocks_;
};

// ---------------------------------------------------------------- streams with a CU mask (process-wide)
// A stream with a CU mask is a hardware queue of its own: creating and destroying four of them per call cost the ragged GC
// call ~150 ms (round 6).  They are kept, idle, between calls -- keyed by device and mask -- and handed to the next call.
class MaskedStreamPool {
public:
    static MaskedStreamPool &get()
    {
        static MaskedStreamPool pool;
        return pool;
    }
    hipError_t acquire(hipStream_t *out, int device, const std::vector<uint32_t> &mask)
    {
        {
            std::lock_guard<std::mutex> g(m_);
            for (auto &e : entries_)
                if (!e.busy && e.device == device && e.mask == mask) {
                    e.busy = true;
                    *out = e.s;
                    return hipSuccess;
                }
        }
        const hipError_t err = hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
        if (err != hipSuccess) return err;
        std::lock_guard<std::mutex> g(m_);
        entries_.push_back({*out, device, mask, true});
        return hipSuccess;
    }
    // true: the stream is the pool's (it stays alive); false: the caller destroys it
    bool release(hipStream_t s)
    {
        std::lock_guard<std::mutex> g(m_);
        for (auto &e : entries_)
            if (e.s == s) {
                e.busy = false;
                return true;
            }
        return false;
    }
    void trim()                                            // destroys every idle stream (tests; vga_release_cached_memory)
    {
        std::lock_guard<std::mutex> g(m_);
        for (int i = (int)entries_.size() - 1; i >= 0; i--)
            if (!entries_[i].busy) {
                (void)hipStreamDestroy(entries_[i].s);
                entries_.erase(entries_.begin() + i);
            }
    }

private:
    struct Entry { hipStream_t s; int device; std::vector<uint32_t> mask; bool busy; };
    std::mutex m_;
    std::vector<Entry> entries_;
};

struct PinnedBlock {
    void *p = nullptr;
    ~PinnedBlock() { if (p) PinnedPool::get().release(p); }
    bool alloc(size_t bytes) { p = PinnedPool::get().acquire(bytes ? bytes : 1); return p != nullptr; }
};

// ---------------------------------------------------------------- the job
// A "unit" is what the reference hands to one task: a channel (GC-ADPCM, ADX) or a stream (HCA).  Unit u owns the
// input rows [u * in_rows_per_unit, (u + 1) * in_rows_per_unit) and the output rows [u * out_rows_per_unit, ...).
// the compute lane of the chunk whose callback is running on this thread (0 when compute_lanes == 1)
inline int &compute_lane() { static thread_local int lane = 0; return lane; }

struct Job {
    int units = 0;
    int chunk_units = 0;                 // kernels are launched per chunk of this many units
    int taper_min_units = 0;             // > 0: the LAST chunk is split into halves down to this size (chunk, ..., chunk/2,
                                         // chunk/4, chunk/4): what runs after the last upload is a small chunk's kernels
    int tail_units = 0;                  // > 0 (takes precedence): the last chunk is split once, into (rest, tail_units)
    int head_units = 0;                  // > 0: the FIRST chunk is this short (when more than one chunk follows it), so that
                                         // the first kernels start after a fraction of a chunk's upload
    // input side (rows of in_row_bytes bytes at host pointers in_rows[r]; device row r at d_in + r * d_in_pitch)
    int in_rows_per_unit = 1;
    const void *const *in_rows = nullptr;
    size_t in_row_bytes = 0;
    char *d_in = nullptr;
    size_t d_in_pitch = 0;
    // output side
    int out_rows_per_unit = 1;
    void *const *out_rows = nullptr;
    size_t out_row_bytes = 0;
    const char *d_out = nullptr;
    size_t d_out_pitch = 0;
    // enqueues the kernels for units [first, first + count) on `stream`; returns 0 or an error code (message via `why`)
    std::function<int(int first, int count, hipStream_t stream, std::string &why)> compute;
    // compute_lanes > 1: chunk k's kernels go to compute stream k % compute_lanes, so that a chunk need not wait for the one
    // before it (the short chunks at the end are bound by their kernels' latency, not by the chip).  The callback learns
    // the lane through compute_lane() and must keep per-lane scratch; chunks of one lane still run in order.
    int compute_lanes = 1;
    // called once per chunk, chunks in order, when the chunk's output rows are complete in the caller's memory (jobs
    // without output rows: when its kernels have run) -- on one of the call's worker threads, never concurrently
    std::function<void(int first, int count)> chunk_done;
    int feeders = 8, drainers = 4;
    // Ragged jobs (the `*_v` entry points: rows of different sizes, packed on the device).  in_row_sizes[r] / d_in_offsets[r]
    // replace in_row_bytes / r * d_in_pitch (same for the output side); in_row_bytes / d_in_pitch then hold the largest row
    // (what a ring slot must be able to take).  A staged slot takes consecutive rows and mirrors their device layout, gaps
    // included, so that it still travels with one copy.
    const size_t *in_row_sizes = nullptr, *d_in_offsets = nullptr;
    const size_t *out_row_sizes = nullptr, *d_out_offsets = nullptr;
    // explicit chunk boundaries in units: chunk k = [chunk_begin[k], chunk_begin[k + 1]) (ragged jobs cut by bytes, not by
    // count); empty = by chunk_units / head_units / tail_units
    std::vector<int> chunk_begin;
    size_t slot_bytes = (size_t)8 << 20;   // target size of one ring slot (whole rows; at least one row)
    int ring = 3;                          // slots per feeder / drainer
    // true: no staging -- the workers hand the caller's rows straight to hipMemcpyAsync (the runtime moves pageable
    // memory at ~51 GB/s on an MI355X host, tools/bench_h2d_modes.hip; CPU-side copies into a pinned ring only compete
    // with the DMA engines for the host's memory bandwidth).  false: stage through the page-locked rings.
    bool direct = true;                    // uploads
    bool direct_out = true;                // downloads (false: through the page-locked ring, whatever `direct` says)
    bool register_rows = true;             // direct mode: hipHostRegister each row for the duration of the call
    bool shared_streams = false;           // every feeder issues on one stream, every drainer on another (3 streams in all)
    int device = 0;
    // Round 6: direct mode with a kernel instead of one copy per row.  transfer(pieces, n, stream, why) enqueues ONE launch
    // that copies n pieces (at most piece_bytes each; device-visible addresses on both sides: a page-locked row's mapped
    // address, an address in d_in / d_out); `pieces` lies in page-locked memory that stays valid until run() returns.
    // gather_in: the feeder (one feeder only) maps each row it page-locks and launches one transfer per chunk instead of a
    // hipMemcpyAsync per row; scatter_out: the drainers likewise (direct_out jobs).  transfer_cus > 0: that many of the
    // device's total_cus compute units are kept free of the job's kernels (the compute streams' CU mask) and the transfer
    // streams are confined to them -- a transfer kernel launched while other kernels hold every CU would start only when
    // their workgroups end (tools/bench_h2d_gather.hip: 11 GB/s instead of 57).
    struct TransferPiece { const void *src; void *dst; unsigned bytes, pad; };
    std::function<int(const TransferPiece *pieces, int n, hipStream_t stream, std::string &why)> transfer;
    bool gather_in = false, scatter_out = false;
    int transfer_cus = 0, total_cus = 0;
    size_t piece_bytes = (size_t)256 << 10;
};

// Where the wall time of one run() went (seconds): per-thread sums, and the slowest thread of each kind.
struct Stats {
    double setup = 0, total = 0;
    double feed_copy = 0, feed_wait_slot = 0, feed_issue = 0, feed_max = 0;       // feeders: memcpy into the ring, waiting for a slot, hipMemcpyAsync calls
    double drain_register = 0;                                                      // drainers: page-locking the caller's output rows (direct mode)
    double feed_boundary = 0, feed_final = 0;                                       // feeders: chunk events + notifications, the final stream sync
    double main_wait_upload = 0, main_launch = 0, main_tail_sync = 0;               // calling thread
    double drain_wait_compute = 0, drain_wait_copy = 0, drain_copy = 0, drain_max = 0;
    int feeders = 0, drainers = 0, chunks = 0, chunk_units = 0;
};

struct Result {
    int code = 0;                          // 0 = ok; otherwise the first failure
    std::string why;
    Stats stats;
};

namespace detail {

inline double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Shared {
    std::mutex m;
    Stats st;                              // guarded by m
    std::condition_variable cv;
    std::vector<int> uploaded;             // per chunk: feeders that have recorded its event
    std::vector<char> launched;            // per chunk: compute event recorded
    std::atomic<int> err{0};
    std::string why;

    void fail(int code, const std::string &msg)
    {
        std::lock_guard<std::mutex> g(m);
        if (!err.load()) {
            why = msg;
            err.store(code ? code : -5);
        }
        cv.notify_all();
    }
};

inline std::string hip_msg(const char *what, hipError_t e)
{
    return std::string(what) + " failed: " + hipGetErrorString(e);
}

#define VGA_PIPE_TRY(expr)                                                   \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) {                                              \
            sh.fail(-5, detail::hip_msg(#expr, _e));                         \
            return;                                                          \
        }                                                                    \
    } while (0)

}  // namespace detail

inline Result run(const Job &job)
{
    using namespace detail;
    Result res;
    if (job.units <= 0) return res;
    const double t_begin = now();
    const int chunk_units = std::max(1, std::min(job.chunk_units > 0 ? job.chunk_units : job.units, job.units));
    // chunk k covers units [cbegin[k], cbegin[k + 1])
    std::vector<int> cbegin;
    if (job.chunk_begin.size() >= 2 && job.chunk_begin.front() == 0 && job.chunk_begin.back() == job.units) {
        cbegin = job.chunk_begin;
    } else {
        int u = 0;
        if (job.head_units > 0 && job.head_units < chunk_units && job.units > job.head_units + chunk_units) {
            cbegin.push_back(0);
            u = job.head_units;
        }
        for (; u < job.units; u += chunk_units) cbegin.push_back(u);
        cbegin.push_back(job.units);
    }
    if (!job.chunk_begin.empty()) {
    } else if (job.tail_units > 0 && cbegin.size() >= 2 && job.units - cbegin[cbegin.size() - 2] > job.tail_units) {
        // the last chunk (the only one of a small call) as (rest, tail): the tail's upload runs under the rest's kernels,
        // and with two kernel lanes both parts' kernels run side by side, the short one last
        cbegin.back() = job.units - job.tail_units;
        cbegin.push_back(job.units);
    } else if (job.taper_min_units > 0 && cbegin.size() >= 3) {   // at least two chunks: taper the last one
        int lo = cbegin[cbegin.size() - 2];
        const int hi = job.units;
        cbegin.pop_back();
        int len = hi - lo;
        while (len >= 2 * job.taper_min_units) {
            lo += len / 2;
            cbegin.push_back(lo);
            len = hi - lo;
        }
        cbegin.push_back(hi);
    }
    const int chunks = (int)cbegin.size() - 1;
    const bool has_in = job.in_rows && job.in_row_bytes > 0 && job.in_rows_per_unit > 0;
    const bool has_out = job.out_rows && job.out_row_bytes > 0 && job.out_rows_per_unit > 0;
    const int F = has_in ? std::max(1, job.feeders) : 0;
    const int D = has_out ? std::max(1, job.drainers) : 0;
    const int R = std::max(2, job.ring);
    const int in_slot_rows = has_in ? (int)std::max<size_t>(1, job.slot_bytes / std::max<size_t>(1, job.d_in_pitch)) : 0;
    const int out_slot_rows = has_out ? (int)std::max<size_t>(1, job.slot_bytes / std::max<size_t>(1, job.d_out_pitch)) : 0;
    // per-row shapes (uniform jobs: the pitch and the one size)
    auto in_size = [&](int r) { return job.in_row_sizes ? job.in_row_sizes[r] : job.in_row_bytes; };
    auto in_off = [&](int r) { return job.d_in_offsets ? job.d_in_offsets[r] : (size_t)r * job.d_in_pitch; };
    auto out_size = [&](int r) { return job.out_row_sizes ? job.out_row_sizes[r] : job.out_row_bytes; };
    auto out_off = [&](int r) { return job.d_out_offsets ? job.d_out_offsets[r] : (size_t)r * job.d_out_pitch; };
    // What a worker takes at a time: a group of consecutive rows of one chunk whose device extent fits a ring slot
    // (uniform jobs: in_slot_rows rows).  groups of chunk k = [gfirst[k], gfirst[k + 1]).
    struct Groups { std::vector<int> begin; std::vector<int> gfirst; size_t slot_cap = 0; };
    auto make_groups = [&](bool have, int rows_per_unit, int slot_rows, size_t pitch, const size_t *sizes, const size_t *offs) {
        Groups g;
        g.slot_cap = (size_t)std::max(slot_rows, 1) * pitch;
        g.gfirst.push_back(0);
        for (int k = 0; k < chunks && have; k++) {
            const int row0 = cbegin[k] * rows_per_unit, row1 = cbegin[k + 1] * rows_per_unit;
            int r = row0;
            while (r < row1) {
                g.begin.push_back(r);
                if (!sizes) {
                    r = std::min(r + slot_rows, row1);
                } else {
                    int e = r + 1;                             // at least one row, then as many as fit the slot
                    while (e < row1 && offs[e] + sizes[e] - offs[r] <= g.slot_cap) e++;
                    r = e;
                }
            }
            g.gfirst.push_back((int)g.begin.size());
        }
        if (!have)
            for (int k = 0; k < chunks; k++) g.gfirst.push_back(0);
        g.begin.push_back(has_in || has_out ? job.units * rows_per_unit : 0);   // sentinel: one past the last row
        return g;
    };
    const Groups gin = make_groups(has_in, job.in_rows_per_unit, in_slot_rows, job.d_in_pitch, job.in_row_sizes, job.d_in_offsets);
    const Groups gout = make_groups(has_out, job.out_rows_per_unit, out_slot_rows, job.d_out_pitch, job.out_row_sizes, job.d_out_offsets);
    // rows of group i of a side: [begin[i], min(begin[i + 1], the chunk's last row))
    auto group_rows = [&](const Groups &g, int k, int i, int rows_per_unit, int &r, int &n) {
        r = g.begin[i];
        const int row1 = cbegin[k + 1] * rows_per_unit;
        n = std::min(g.begin[i + 1], row1) - r;
    };

    std::vector<std::atomic<int>> next_in(chunks), next_out(chunks);   // per chunk: first group not yet taken by a worker
    for (auto &a : next_in) a.store(0);
    for (auto &a : next_out) a.store(0);
    Shared sh;
    sh.uploaded.assign(chunks, 0);
    sh.launched.assign(chunks, 0);

    // everything HIP-side is created up front on the calling thread, destroyed after every thread has joined
    std::vector<hipStream_t> fstream(F, nullptr), dstream(D, nullptr);
    std::vector<hipEvent_t> fslot(F * R, nullptr), dslot(D * R, nullptr), upl(F * chunks, nullptr), comp(chunks, nullptr);
    const int CL = std::max(1, std::min(job.compute_lanes, 4));
    std::vector<hipStream_t> cstreams(CL, nullptr);
    PinnedBlock in_ring, out_ring;
    bool ok = true;
    auto check = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && ok) {
            ok = false;
            res.code = -5;
            res.why = hip_msg(what, e);
        }
    };
    // the transfer kernels' own compute units (Job::transfer_cus): bits [0, transfer_cus) of the CU mask for the transfer
    // streams, the rest for the compute streams
    const bool gather_in = has_in && job.direct && job.gather_in && job.transfer && F == 1;
    const bool scatter_out = has_out && job.direct_out && job.scatter_out && job.transfer;
    // staged downloads: the ring slot's one contiguous copy by a transfer launch as well (the ring is page-locked and
    // device-visible already: nothing of the caller's has to be page-locked for it)
    const bool scatter_ring = has_out && !job.direct_out && job.scatter_out && job.transfer;
    const bool masked = (gather_in || scatter_out || scatter_ring) && job.transfer_cus > 0 && job.total_cus > 2 * job.transfer_cus;
    std::vector<uint32_t> cmask, tmask;
    if (masked) {
        const int words = (job.total_cus + 31) / 32;
        cmask.assign(words, 0xFFFFFFFFu);
        tmask.assign(words, 0u);
        for (int b = 0; b < job.transfer_cus; b++) {
            cmask[b / 32] &= ~(1u << (b % 32));
            tmask[b / 32] |= 1u << (b % 32);
        }
    }
    auto make_stream = [&](hipStream_t *st, bool transfer_side) {
        if (masked) {
            return MaskedStreamPool::get().acquire(st, job.device, transfer_side ? tmask : cmask);
        }
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    for (auto &c : cstreams) check(make_stream(&c, false), "hipStreamCreate");
    const bool shared_streams = job.shared_streams;
    auto make_side = [&](hipStream_t *st, bool uses_transfer) {
        return uses_transfer ? make_stream(st, true) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    if (shared_streams) {
        if (F > 0) check(make_side(&fstream[0], gather_in), "hipStreamCreate");
        if (D > 0) check(make_side(&dstream[0], scatter_out || scatter_ring), "hipStreamCreate");
        for (auto &s : fstream) s = fstream[0];
        for (auto &s : dstream) s = dstream[0];
    } else {
        for (auto &s : fstream) check(make_side(&s, gather_in), "hipStreamCreate");
        for (auto &s : dstream) check(make_side(&s, scatter_out || scatter_ring), "hipStreamCreate");
    }
    // piece tables (page-locked, device-visible): every row in pieces of at most piece_bytes; a drainer copies whatever
    // rows it claims, so each drainer's table can hold them all
    const size_t piece = std::max<size_t>(job.piece_bytes, 4096);
    auto pieces_of = [&](bool have, int rows_per_unit, const size_t *sizes, size_t uniform) {
        size_t n = 0;
        if (!have) return n;
        const int rows = job.units * rows_per_unit;
        for (int r = 0; r < rows; r++) n += ((sizes ? sizes[r] : uniform) + piece - 1) / piece;
        return n;
    };
    const size_t in_pieces = gather_in ? pieces_of(has_in, job.in_rows_per_unit, job.in_row_sizes, job.in_row_bytes) : 0;
    size_t out_pieces = scatter_out ? pieces_of(has_out, job.out_rows_per_unit, job.out_row_sizes, job.out_row_bytes) : 0;
    if (scatter_ring)                                  // a slot's copy covers its rows' device extent, gaps included
        for (int k = 0; k < chunks; k++)
            for (int gi = gout.gfirst[k]; gi < gout.gfirst[k + 1]; gi++) {
                int r, n;
                group_rows(gout, k, gi, job.out_rows_per_unit, r, n);
                out_pieces += (out_off(r + n - 1) - out_off(r) + out_size(r + n - 1) + piece - 1) / piece;
            }
    PinnedBlock in_table, out_table;
    if (ok && in_pieces && !in_table.alloc(in_pieces * sizeof(Job::TransferPiece))) check(hipErrorOutOfMemory, "hipHostMalloc (piece table)");
    if (ok && out_pieces && !out_table.alloc((size_t)D * out_pieces * sizeof(Job::TransferPiece))) check(hipErrorOutOfMemory, "hipHostMalloc (piece table)");
    // output rows page-locked and mapped so far (scatter_out): the device-visible address, nullptr until some drainer has it
    std::vector<std::atomic<void *>> mapped_out(scatter_out ? (size_t)job.units * job.out_rows_per_unit : 0);
    for (auto &m : mapped_out) m.store(nullptr);
    auto append_pieces = [&](Job::TransferPiece *table, size_t &np, const char *src, char *dst, size_t bytes) {
        for (size_t o = 0; o < bytes; o += piece)
            table[np++] = Job::TransferPiece{src + o, dst + o, (unsigned)std::min(piece, bytes - o), 0u};
    };
    const bool timeline = std::getenv("VGA_HIP_PIPELINE_TIMELINE") != nullptr;
    std::vector<hipEvent_t> cstart(chunks + 1, nullptr), dlev(timeline ? D * chunks : 0, nullptr);
    for (auto *v : {&fslot, &dslot})
        for (auto &e : *v) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    for (auto *v : {&upl, &comp, &cstart, &dlev})
        for (auto &e : *v) check(hipEventCreateWithFlags(&e, timeline ? hipEventDefault : hipEventDisableTiming), "hipEventCreate");
    if (timeline) check(hipEventRecord(cstart[chunks], cstreams[0]), "hipEventRecord");
    if (ok && has_in && !job.direct && !in_ring.alloc((size_t)F * R * gin.slot_cap)) check(hipErrorOutOfMemory, "hipHostMalloc (input ring)");
    if (ok && has_out && !job.direct_out && !out_ring.alloc((size_t)D * R * gout.slot_cap)) check(hipErrorOutOfMemory, "hipHostMalloc (output ring)");

    // progress (job.chunk_done): every drainer marks the end of its copies of chunk k with an event and, while it waits for
    // the next chunk's kernels anyway, for that event; the drainer that is last to see chunk k complete reports it, after
    // the chunks before it (report_mutex / next_report keep the order)
    const bool progress = (bool)job.chunk_done;
    std::vector<hipEvent_t> dend(progress ? (size_t)D * chunks : 0, nullptr);
    for (auto &e : dend) check(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    std::vector<int> chunk_seen(chunks, 0);            // drainers that have seen chunk k's copies complete (under report_mutex)
    std::mutex report_mutex;
    int next_report = 0;
    auto chunk_complete = [&](int k) {
        std::lock_guard<std::mutex> g(report_mutex);
        chunk_seen[k]++;
        while (next_report < chunks && chunk_seen[next_report] >= std::max(D, 1)) {
            job.chunk_done(cbegin[next_report], cbegin[next_report + 1] - cbegin[next_report]);
            next_report++;
        }
    };
    auto chunk_units_of = [&](int k) { return cbegin[k + 1] - cbegin[k]; };
    std::vector<std::vector<void *>> locked_in(F), locked_out(D);      // rows page-locked for the call, per worker (see below)

    // ---------------------------------------------------------------- feeder t
    auto feeder = [&](int t) {
        VGA_PIPE_TRY(hipSetDevice(job.device));
        char *ring = static_cast<char *>(in_ring.p) + (size_t)t * R * gin.slot_cap;
        int64_t used = 0;                                  // slots handed to the DMA engine so far
        std::vector<void *> &registered = locked_in[t];    // direct mode: rows this thread page-locked for the call
        double t_copy = 0, t_wait = 0, t_issue = 0, t_boundary = 0, t_final = 0;
        const double t_start = now();
        struct Report {
            Shared &sh; double &c, &w, &i, &b, &f; const double &t0;
            ~Report() { std::lock_guard<std::mutex> g(sh.m); sh.st.feed_copy += c; sh.st.feed_wait_slot += w; sh.st.feed_issue += i;
                        sh.st.feed_boundary += b; sh.st.feed_final += f;
                        sh.st.feed_max = std::max(sh.st.feed_max, now() - t0); }
        } report{sh, t_copy, t_wait, t_issue, t_boundary, t_final, t_start};
        Job::TransferPiece *table = static_cast<Job::TransferPiece *>(in_table.p);   // gather_in: this (the only) feeder's pieces
        size_t np = 0, np_sent = 0;
        for (int k = 0; k < chunks && !sh.err.load(); k++) {
            // row groups are handed out on demand: a feeder that runs on a core far from the caller's pages (or shares
            // its core) simply takes fewer of them -- with a fixed split the slowest thread set the upload time
            for (;;) {
                if (sh.err.load()) break;
                const int gi = gin.gfirst[k] + next_in[k].fetch_add(1);
                if (gi >= gin.gfirst[k + 1]) break;
                int r, n;
                group_rows(gin, k, gi, job.in_rows_per_unit, r, n);
                if (job.direct) {
                    const double ta = now();
                    for (int i = 0; i < n; i++) {
                        // registered rows go through the DMA engines asynchronously (and next to running kernels); a row
                        // that cannot be registered (already registered by the caller, shares a page with its neighbour,
                        // ...) is copied by the runtime's pageable path, which is correct but blocks this thread
                        void *row = const_cast<void *>(job.in_rows[r + i]);
                        const size_t bytes = in_size(r + i);
                        if (bytes == 0) continue;
                        bool locked = false;
                        if (job.register_rows && hipHostRegister(row, bytes, gather_in ? hipHostRegisterMapped : hipHostRegisterDefault) == hipSuccess) {
                            registered.push_back(row);
                            locked = true;
                        } else
                            (void)hipGetLastError();
                        if (gather_in && locked) {         // the chunk's gather kernel fetches it (below)
                            void *dev_view = nullptr;
                            if (hipHostGetDevicePointer(&dev_view, row, 0) == hipSuccess && dev_view) {
                                append_pieces(table, np, static_cast<const char *>(dev_view), job.d_in + in_off(r + i), bytes);
                                continue;
                            }
                            (void)hipGetLastError();
                        }
                        VGA_PIPE_TRY(hipMemcpyAsync(job.d_in + in_off(r + i), row, bytes, hipMemcpyHostToDevice, fstream[t]));
                    }
                    t_issue += now() - ta;
                    continue;
                }
                const int s = (int)(used % R);
                double ta = now();
                if (used >= R) VGA_PIPE_TRY(hipEventSynchronize(fslot[t * R + s]));      // the slot's previous upload is done
                double tb = now();
                t_wait += tb - ta;
                char *slot = ring + (size_t)s * gin.slot_cap;
                for (int i = 0; i < n; i++) {
                    const size_t at = in_off(r + i) - in_off(r);
                    if (in_size(r + i)) std::memcpy(slot + at, job.in_rows[r + i], in_size(r + i));
                    // ragged rows: the slot travels as one copy, gaps included -- what lies between two rows on the device
                    // (padding the kernels may rely on being zero) must not receive the slot's previous contents
                    if (job.in_row_sizes && i + 1 < n) std::memset(slot + at + in_size(r + i), 0, in_off(r + i + 1) - in_off(r) - at - in_size(r + i));
                }
                ta = now();
                t_copy += ta - tb;
                const size_t bytes = in_off(r + n - 1) - in_off(r) + in_size(r + n - 1);
                if (bytes) VGA_PIPE_TRY(hipMemcpyAsync(job.d_in + in_off(r), slot, bytes, hipMemcpyHostToDevice, fstream[t]));
                VGA_PIPE_TRY(hipEventRecord(fslot[t * R + s], fstream[t]));
                t_issue += now() - ta;
                used++;
            }
            const double tb0 = now();
            if (gather_in && np > np_sent && !sh.err.load()) {
                std::string why;
                const int rc = job.transfer(table + np_sent, (int)(np - np_sent), fstream[t], why);
                if (rc) {
                    sh.fail(rc, why);
                    return;
                }
                np_sent = np;
            }
            VGA_PIPE_TRY(hipEventRecord(upl[t * chunks + k], fstream[t]));
            {
                std::lock_guard<std::mutex> g(sh.m);
                sh.uploaded[k]++;
            }
            sh.cv.notify_all();
            t_boundary += now() - tb0;
        }
        const double tf0 = now();
        if (!sh.err.load()) VGA_PIPE_TRY(hipStreamSynchronize(fstream[t]));             // nothing in flight reads the ring after this
        t_final = now() - tf0;
    };

    // ---------------------------------------------------------------- drainer u
    auto drainer = [&](int u) {
        VGA_PIPE_TRY(hipSetDevice(job.device));
        char *ring = static_cast<char *>(out_ring.p) + (size_t)u * R * gout.slot_cap;
        struct Pending { int row = -1, n = 0; };
        std::vector<Pending> pend(R);
        int64_t used = 0;
        int reported = 0;                                  // progress: chunks [0, reported) of this drainer are complete
        std::vector<void *> &registered = locked_out[u];
        Job::TransferPiece *otable = static_cast<Job::TransferPiece *>(out_table.p) + (size_t)u * out_pieces;   // scatter_out: this drainer's pieces
        size_t onp = 0;
        double t_wait_comp = 0, t_wait_copy = 0, t_copy = 0, t_register = 0;
        const double t_start = now();
        struct Report {
            Shared &sh; double &a, &b, &c, &r; const double &t0;
            ~Report() { std::lock_guard<std::mutex> g(sh.m); sh.st.drain_wait_compute += a; sh.st.drain_wait_copy += b; sh.st.drain_copy += c;
                        sh.st.drain_register += r;
                        sh.st.drain_max = std::max(sh.st.drain_max, now() - t0); }
        } report{sh, t_wait_comp, t_wait_copy, t_copy, t_register, t_start};
        // direct mode: page-lock this drainer's share of the output rows now, while it has nothing else to do (rows it
        // cannot lock are copied by the runtime's pageable path later, which is correct but blocks)
        if (job.direct_out && job.register_rows) {
            const double ta = now();
            const int total_rows = job.units * job.out_rows_per_unit;
            for (int r = u; r < total_rows; r += D) {
                void *row = job.out_rows[r];
                if (out_size(r) == 0) continue;
                if (hipHostRegister(row, out_size(r), scatter_out ? hipHostRegisterMapped : hipHostRegisterDefault) == hipSuccess) {
                    registered.push_back(row);
                    void *dev_view = nullptr;
                    if (scatter_out && hipHostGetDevicePointer(&dev_view, row, 0) == hipSuccess && dev_view) mapped_out[r].store(dev_view);
                    else if (scatter_out) (void)hipGetLastError();
                } else
                    (void)hipGetLastError();
            }
            t_register = now() - ta;
        }
        auto flush = [&](int s) -> bool {                  // slot s: wait for its download, hand the rows to the caller
            if (pend[s].row < 0) return true;
            const double ta = now();
            hipError_t e = hipEventSynchronize(dslot[u * R + s]);
            if (e != hipSuccess) {
                sh.fail(-5, hip_msg("hipEventSynchronize", e));
                return false;
            }
            const double tb = now();
            t_wait_copy += tb - ta;
            const char *slot = ring + (size_t)s * gout.slot_cap;
            for (int i = 0; i < pend[s].n; i++)
                if (out_size(pend[s].row + i))
                    std::memcpy(job.out_rows[pend[s].row + i], slot + (out_off(pend[s].row + i) - out_off(pend[s].row)), out_size(pend[s].row + i));
            t_copy += now() - tb;
            pend[s].row = -1;
            return true;
        };
        for (int k = 0; k < chunks; k++) {
            {
                const double ta = now();
                std::unique_lock<std::mutex> g(sh.m);
                sh.cv.wait(g, [&] { return sh.launched[k] || sh.err.load(); });
                t_wait_comp += now() - ta;
            }
            if (sh.err.load()) break;
            VGA_PIPE_TRY(hipStreamWaitEvent(dstream[u], comp[k], 0));
            {                                              // not needed for ordering (the copies are stream-ordered behind the
                const double ta = now();                   // event); it only attributes the wait to "compute" in the stats
                VGA_PIPE_TRY(hipEventSynchronize(comp[k]));
                t_wait_comp += now() - ta;
            }
            const int row0 = cbegin[k] * job.out_rows_per_unit;
            for (;;) {
                if (sh.err.load()) break;
                const int gi = gout.gfirst[k] + next_out[k].fetch_add(1);
                if (gi >= gout.gfirst[k + 1]) break;
                int r, n;
                group_rows(gout, k, gi, job.out_rows_per_unit, r, n);
                if (job.direct_out) {
                    const double ta = now();
                    const size_t np0 = onp;
                    for (int i = 0; i < n; i++) {
                        void *row = job.out_rows[r + i];
                        if (out_size(r + i) == 0) continue;
                        void *dev_view = scatter_out ? mapped_out[r + i].load() : nullptr;
                        if (dev_view) {                    // this group's scatter kernel writes it (below)
                            append_pieces(otable, onp, job.d_out + out_off(r + i), static_cast<char *>(dev_view), out_size(r + i));
                            continue;
                        }
                        VGA_PIPE_TRY(hipMemcpyAsync(row, job.d_out + out_off(r + i), out_size(r + i), hipMemcpyDeviceToHost, dstream[u]));
                    }
                    if (onp > np0) {
                        std::string why;
                        const int rc = job.transfer(otable + np0, (int)(onp - np0), dstream[u], why);
                        if (rc) {
                            sh.fail(rc, why);
                            return;
                        }
                    }
                    t_copy += now() - ta;
                    continue;
                }
                const int s = (int)(used % R);
                if (!flush(s)) return;
                char *slot = ring + (size_t)s * gout.slot_cap;
                const size_t bytes = out_off(r + n - 1) - out_off(r) + out_size(r + n - 1);
                if (bytes && scatter_ring) {
                    const size_t np0 = onp;
                    append_pieces(otable, onp, job.d_out + out_off(r), slot, bytes);
                    std::string why;
                    const int rc = job.transfer(otable + np0, (int)(onp - np0), dstream[u], why);
                    if (rc) {
                        sh.fail(rc, why);
                        return;
                    }
                } else if (bytes)
                    VGA_PIPE_TRY(hipMemcpyAsync(slot, job.d_out + out_off(r), bytes, hipMemcpyDeviceToHost, dstream[u]));
                VGA_PIPE_TRY(hipEventRecord(dslot[u * R + s], dstream[u]));
                pend[s].row = r;
                pend[s].n = n;
                used++;
            }
            if (timeline) VGA_PIPE_TRY(hipEventRecord(dlev[u * chunks + k], dstream[u]));
            if (progress && !sh.err.load()) {
                // this chunk's copies are queued; the ones of the chunk before have had its kernels' time to finish: see
                // them complete (staged mode: hand their rows out now rather than when the slot is needed again) and
                // say so.  Nothing is waiting for this thread until the next chunk's kernels have run.
                VGA_PIPE_TRY(hipEventRecord(dend[u * chunks + k], dstream[u]));
                if (k > 0) {
                    const double ta = now();
                    VGA_PIPE_TRY(hipEventSynchronize(dend[u * chunks + k - 1]));
                    t_wait_copy += now() - ta;
                    for (int s = 0; s < R; s++)
                        if (pend[s].row >= 0 && pend[s].row < row0 && !flush(s)) return;
                    chunk_complete(k - 1);
                    reported = k;
                }
            }
        }
        if (!sh.err.load())
            for (int s = 0; s < R; s++)
                if (!flush((int)((used + s) % R))) return;  // oldest first
        if (!sh.err.load() && job.direct_out) {
            const double ta = now();
            VGA_PIPE_TRY(hipStreamSynchronize(dstream[u]));   // the caller's rows are complete when run() returns
            t_wait_copy += now() - ta;
        }
        if (progress && !sh.err.load())
            for (; reported < chunks; reported++) chunk_complete(reported);
    };

    // Rows page-locked for the call (direct mode), per worker.  They are unlocked only at the end of run(), after every
    // thread has joined and every stream has drained: a drainer locks its share of the rows up front but copies whatever
    // rows it claims, so a worker that unlocked "its" rows when it finished could pull them from under another worker's
    // copy (hipHostUnregister happens to synchronise every stream on ROCm, which hid that; tests/host/mockhip counts it).
    std::vector<std::thread> threads;
    const double t_setup_done = now();
    if (ok) {
        for (int t = 0; t < F; t++) threads.emplace_back(feeder, t);
        for (int u = 0; u < D; u++) threads.emplace_back(drainer, u);
        // ---------------------------------------------------------------- the calling thread: kernels, chunk by chunk
        [&] {
            double t_wait = 0, t_launch = 0;
            struct Report {
                Shared &sh; double &a, &b;
                ~Report() { std::lock_guard<std::mutex> g(sh.m); sh.st.main_wait_upload = a; sh.st.main_launch = b; }
            } report{sh, t_wait, t_launch};
            for (int k = 0; k < chunks; k++) {
                double ta = now();
                if (F > 0) {
                    std::unique_lock<std::mutex> g(sh.m);
                    sh.cv.wait(g, [&] { return sh.uploaded[k] == F || sh.err.load(); });
                }
                if (sh.err.load()) return;
                double tb = now();
                t_wait += tb - ta;
                hipStream_t cstream = cstreams[k % CL];
                compute_lane() = k % CL;
                for (int t = 0; t < F; t++) VGA_PIPE_TRY(hipStreamWaitEvent(cstream, upl[t * chunks + k], 0));
                if (timeline) VGA_PIPE_TRY(hipEventRecord(cstart[k], cstream));
                std::string why;
                const int rc = job.compute(cbegin[k], chunk_units_of(k), cstream, why);
                if (rc) {
                    sh.fail(rc, why);
                    return;
                }
                VGA_PIPE_TRY(hipEventRecord(comp[k], cstream));
                t_launch += now() - tb;
                {
                    std::lock_guard<std::mutex> g(sh.m);
                    sh.launched[k] = 1;
                }
                sh.cv.notify_all();
            }
            const double tc = now();
            for (auto c : cstreams) VGA_PIPE_TRY(hipStreamSynchronize(c));
            {
                std::lock_guard<std::mutex> g(sh.m);
                sh.st.main_tail_sync = now() - tc;
            }
            if (progress && D == 0 && !sh.err.load())
                for (int k = 0; k < chunks; k++) chunk_complete(k);
        }();
        for (auto &th : threads) th.join();
        if (sh.err.load()) {
            res.code = sh.err.load();
            res.why = sh.why;
        }
    }
    // after a failure, operations may still be in flight on the rings: drain every stream before anything is released
    for (auto c : cstreams) if (c) (void)hipStreamSynchronize(c);
    for (auto s : fstream) if (s) (void)hipStreamSynchronize(s);
    for (auto s : dstream) if (s) (void)hipStreamSynchronize(s);
    for (auto *side : {&locked_in, &locked_out})
        for (auto &rows : *side)
            for (void *p : rows) (void)hipHostUnregister(p);
    if (timeline && ok && !sh.err.load()) {
        auto at = [&](hipEvent_t e) { float ms = -1; (void)hipEventElapsedTime(&ms, cstart[chunks], e); return ms; };
        for (int k = 0; k < chunks; k++) {
            std::fprintf(stderr, "timeline chunk %d: uploaded", k);
            for (int t = 0; t < F; t++) std::fprintf(stderr, " %.1f", at(upl[t * chunks + k]));
            std::fprintf(stderr, "  compute %.1f .. %.1f  downloaded", at(cstart[k]), at(comp[k]));
            for (int u = 0; u < D; u++) std::fprintf(stderr, " %.1f", at(dlev[u * chunks + k]));
            std::fprintf(stderr, " ms\n");
        }
    }
    for (auto *v : {&fslot, &dslot, &upl, &comp, &cstart, &dlev, &dend})
        for (auto e : *v) if (e) (void)hipEventDestroy(e);
    if (shared_streams) { fstream.resize(F > 0 ? 1 : 0); dstream.resize(D > 0 ? 1 : 0); }
    for (auto *v : {&fstream, &dstream, &cstreams})      // (masked streams go back to their pool, idle: everything above has synchronised them)
        for (auto st : *v)
            if (st && !MaskedStreamPool::get().release(st)) (void)hipStreamDestroy(st);
    res.stats = sh.st;
    res.stats.setup = t_setup_done - t_begin;
    res.stats.total = now() - t_begin;
    res.stats.feeders = F;
    res.stats.drainers = D;
    res.stats.chunks = chunks;
    res.stats.chunk_units = chunk_units;
    return res;
}

#undef VGA_PIPE_TRY

// ---------------------------------------------------------------- several GPUs behind one call
// The reference's parallelism lives inside one process (Parallel.For over channels, GcAdpcmFormat.cs:65-68; a worker per
// file, Cli/Batch.cs:24-25), and so does a P/Invoke host: one call, one caller, N GPUs.  Units are independent, every
// result lands in the caller's own rows, and each GPU has its own PCIe link -- so the units are cut into contiguous
// shares, one per listed device, and every share runs the whole single-device entry point (its own device buffers,
// feeder / drainer threads and streams: everything above) on its own host thread with that device current.  No
// collective, no peer copies.  The calling thread takes share 0.
struct Share {
    int index = 0, device = 0, first = 0, count = 0;
};

// shares of `units` over `devices` (a device may be listed more than once); shares would be smaller than `min_units`
// -> fewer of them.  Blocks differ by at most one unit.
inline std::vector<Share> plan_shares(const std::vector<int> &devices, int units, int min_units)
{
    std::vector<Share> out;
    if (units <= 0 || devices.empty()) return out;
    int n = (int)devices.size();
    if (min_units > 0) n = std::max(1, std::min(n, units / min_units));
    n = std::max(1, std::min(n, units));
    const int base = units / n, extra = units % n;
    int first = 0;
    for (int k = 0; k < n; k++) {
        const int count = base + (k < extra ? 1 : 0);
        out.push_back({k, devices[k], first, count});
        first += count;
    }
    return out;
}

// body(const Share &, std::string &why) -> 0 or an error code.  Returns the first failing share's code and message
// (by share index), after every share has finished.  With one share the body runs inline and the device is not touched.
template <class Body>
inline Result run_on_devices(const std::vector<Share> &shares, Body &&body)
{
    Result res;
    if (shares.empty()) return res;
    if (shares.size() == 1) {
        res.code = body(shares[0], res.why);
        return res;
    }
    int caller_device = 0;
    (void)hipGetDevice(&caller_device);
    std::vector<Result> each(shares.size());
    auto one = [&](size_t k) {
        const hipError_t e = hipSetDevice(shares[k].device);
        if (e != hipSuccess) {
            each[k].code = -5;
            each[k].why = detail::hip_msg("hipSetDevice", e);
            return;
        }
        each[k].code = body(shares[k], each[k].why);
    };
    std::vector<std::thread> threads;
    for (size_t k = 1; k < shares.size(); k++) threads.emplace_back(one, k);
    one(0);
    for (auto &t : threads) t.join();
    (void)hipSetDevice(caller_device);
    for (auto &r : each)
        if (r.code) return r;
    return res;
}

}  // namespace pipe
}  // namespace vga
