// common.hpp -- shared host-side helpers for libvgaudio_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/vgaudio_hip.h"

namespace vga {

void set_error(const char *fmt, ...);

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

// RAII device buffer (host-buffer entry points only)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) {
        bytes = n;
        return hipMalloc(&p, n ? n : 1);
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

// compute units of the current device (256 on an MI355X): how many time pieces fill the chip
inline int device_cu_count()
{
    int device = 0, cus = 256;
    if (hipGetDevice(&device) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    return cus > 0 ? cus : 256;
}

// test hook (vga_testing_force_open_seams_this_thread, include/vgaudio_hip_testing.h): the seam kernels of the time-segmented codecs then never accept a seam as
// closed, so that their fall-back (re-computing the rest of the channel serially) is what produces the output
int force_open_seams();          // 0 = off, 1 = every seam, 2 = seams with an even index of every third channel

// the per-(channel, seam) reading of that mode inside the seam kernels
__host__ __device__ inline bool seam_forced_open(int mode, int channel, int seam)
{
    return mode == 1 || (mode == 2 && channel % 3 == 0 && seam % 2 == 0);
}

}  // namespace vga
