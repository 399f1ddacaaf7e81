// common.hpp -- shared host-side helpers for libvgaudio_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/vgaudio_hip.h"

namespace vga {

void set_error(const char *fmt, ...);
// true once after set_error() on this thread (DevBuf: synchronise before a block goes back to the pool)
bool take_error_pending();

// status helper: HIP failure -> VGA_ERR_DEVICE with message
#define VGA_HIP_TRY(expr)                                                              \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            vga::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                           __FILE__, __LINE__);                                        \
            return VGA_ERR_DEVICE;                                                     \
        }                                                                              \
    } while (0)

// Process-wide cache of the host-buffer entry points' device allocations.  hipMalloc maps memory at ~25 GB/s (1.4 s for
// the 44 GB of a configs[1] call: more than the call's transfers and kernels together); a caller converting batch after
// batch gets the previous call's buffers back instead.  Blocks of at least 1 MiB are parked on release (up to 64 GiB PER
// DEVICE -- a process that drives eight GPUs keeps eight working sets; VGA_HIP_POOL_GIB changes the figure, 0 turns the
// cache off -- largest dropped first) and handed to the next request they fit without wasting more than half; smaller
// ones go straight to hipMalloc / hipFree.  vga_release_cached_memory() empties the cache.  A block released while the
// calling thread is unwinding from a failed call (set_error() was called) may still be read or written by work that call
// enqueued: the device is synchronised first (hipFree used to do that implicitly).
class DevicePool {
public:
    static DevicePool &get()
    {
        static DevicePool pool;
        return pool;
    }
    hipError_t acquire(void **out, size_t bytes)
    {
        if (bytes < kMinPooled) return hipMalloc(out, bytes ? bytes : 1);
        int device = 0;
        (void)hipGetDevice(&device);
        {
            std::lock_guard<std::mutex> g(m_);
            int best = -1;
            for (int i = 0; i < (int)blocks_.size(); i++) {
                const Block &b = blocks_[i];
                if (!b.busy && b.device == device && b.bytes >= bytes && b.bytes / 2 <= bytes &&
                    (best < 0 || b.bytes < blocks_[best].bytes))
                    best = i;
            }
            if (best >= 0) {
                blocks_[best].busy = true;
                *out = blocks_[best].p;
                return hipSuccess;
            }
        }
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {                             // out of memory with blocks parked: drop them and retry once
            trim();
            e = hipMalloc(out, bytes);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(m_);
            blocks_.push_back({*out, bytes, device, true});
        }
        return e;
    }
    void release(void *p)
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(m_);
            bool pooled = false;
            int device = -1;
            for (auto &b : blocks_)
                if (b.p == p) { b.busy = false; pooled = true; device = b.device; }
            size_t idle = 0;                               // parked on the released block's device
            for (auto &b : blocks_)
                if (!b.busy && b.device == device) idle += b.bytes;
            if (!pooled) drop.push_back(p);
            const size_t keep = keep_bytes();
            while (idle > keep) {
                int big = -1;
                for (int i = 0; i < (int)blocks_.size(); i++)
                    if (!blocks_[i].busy && blocks_[i].device == device && (big < 0 || blocks_[i].bytes > blocks_[big].bytes)) big = i;
                if (big < 0) break;
                idle -= blocks_[big].bytes;
                drop.push_back(blocks_[big].p);
                blocks_.erase(blocks_.begin() + big);
            }
        }
        for (void *q : drop) (void)hipFree(q);
    }
    void trim()
    {
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = (int)blocks_.size() - 1; i >= 0; i--)
                if (!blocks_[i].busy) {
                    drop.push_back(blocks_[i].p);
                    blocks_.erase(blocks_.begin() + i);
                }
        }
        for (void *q : drop) (void)hipFree(q);
    }

    // device the block was allocated on (-1: not pooled)
    int device_of(void *p)
    {
        std::lock_guard<std::mutex> g(m_);
        for (auto &b : blocks_)
            if (b.p == p) return b.device;
        return -1;
    }

private:
    static constexpr size_t kMinPooled = (size_t)1 << 20;
    static size_t keep_bytes()
    {
        static const size_t v = [] {
            const char *e = std::getenv("VGA_HIP_POOL_GIB");
            const long gib = e ? std::atol(e) : 64;
            return (size_t)(gib < 0 ? 0 : gib) << 30;
        }();
        return v;
    }
    struct Block { void *p; size_t bytes; int device; bool busy; };
    std::mutex m_;
    std::vector<Block> blocks_;
};

// RAII device buffer (host-buffer entry points only)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { drop(); }
    void drop()
    {
        if (!p) return;
        if (take_error_pending()) {                        // a failed call's work may still touch the block: wait for ITS device
            const int dev = DevicePool::get().device_of(p);
            int cur = -1;
            if (dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) {
                (void)hipDeviceSynchronize();
                (void)hipSetDevice(cur);
            } else
                (void)hipDeviceSynchronize();
        }
        DevicePool::get().release(p);
        p = nullptr;
    }
    hipError_t alloc(size_t n) {
        drop();
        bytes = n;
        return DevicePool::get().acquire(&p, n ? n : 1);
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// stream-ordered scratch of the segmented launchers: freed on every exit path (hipFreeAsync on the same stream)
struct AsyncBuf {
    void *p = nullptr;
    hipStream_t stream = nullptr;
    AsyncBuf() = default;
    AsyncBuf(const AsyncBuf &) = delete;
    AsyncBuf &operator=(const AsyncBuf &) = delete;
    ~AsyncBuf() { if (p) (void)hipFreeAsync(p, stream); }
    hipError_t alloc(size_t n, hipStream_t s) {
        stream = s;
        return hipMallocAsync(&p, n ? n : 1, s);
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of the function: set it on every launch (a
// host-side table update, microseconds) instead of caching a process-wide "done" flag, which would be wrong after
// vga_set_device() picks another GPU and racy between threads.
template <class F>
inline hipError_t allow_dynamic_lds(F *kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct Stream {
    hipStream_t s = nullptr;
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
    hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
};

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Workgroup barrier for hand-offs that go through LDS only.  __syncthreads() also drains vmcnt, i.e. every
// global load still in flight (tile prefetches) and every store (tile flushes) -- a full HBM round trip per
// tile for the helper waves.  Here only the LDS counter is waited for; the "memory" clobber keeps the compiler
// from moving memory operations across.
__device__ __forceinline__ void lds_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// fails loudly when no gfx950 device is present: there is no CPU fallback.
int require_device();

// compute units of the current device (256 on an MI355X): how many time pieces fill the chip (LABNOTES.md 4.3)
inline int device_cu_count()
{
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    return cus > 0 ? cus : 256;
}

// test hook (vga_testing_force_open_seams_this_thread, include/vgaudio_hip_testing.h): the seam kernels of the time-segmented codecs then never accept a seam as
// closed, so that their fall-back (re-computing the rest of the channel serially) is what produces the output
int force_open_seams();          // 0 = off, 1 = every seam, 2 = seams with an even index of every third channel, 3 = every seam, and the
                                 // decoders count them as seams that would not close (their REPAIR launch takes over)

// lane layout of the GC-ADPCM encoder wave (gc_encode_kernel.hip): 8 = (channel, predictor); 4 = (channel, predictor,
// candidate); 0 = the launcher's choice (4 for batches too small for persistent workgroups, 8 otherwise).  Thread-local
// test hook vga_testing_gc_encoder_layout_this_thread: both must give the same bytes.
int encoder_layout();
int coefs_kernel_variant();               // 0: four channels + summing wave per workgroup (product); 1: one wave per channel
int encoder_segments_override();   // > 0: time pieces per channel forced by the test hook
int encoder_persistent_mode();     // 0: the launcher decides; 1: one workgroup per (channel group, piece); 2: persistent workgroups + queue
int hca_frames_per_group_override();   // > 0: frames per workgroup of hca_frames_kernel forced by the test hook

// the per-(channel, seam) reading of that mode inside the seam kernels
__host__ __device__ inline bool seam_forced_open(int mode, int channel, int seam)
{
    return mode == 1 || mode == 3 || (mode == 2 && channel % 3 == 0 && seam % 2 == 0);
}

}  // namespace vga
