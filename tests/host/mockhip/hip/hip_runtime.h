// Mock of the handful of HIP runtime entry points host_pipeline.hpp uses -- TEST INFRASTRUCTURE for the CPU-only
// suite (tests/test_host_pipeline.py).  A stream is a worker thread draining a FIFO of closures, so copies, event
// records and stream-waits are genuinely asynchronous and ordered as on a device: a missing wait in the pipeline shows
// up as corrupted data, a lost wake-up as a hang, a data race under -fsanitize=thread.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDefault = 0, hipHostMallocPortable = 1;

struct MockStream {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool stop = false, idle = true;
    std::thread th;
    MockStream() : th([this] { loop(); }) {}
    ~MockStream()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void push(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> g(m);
            q.push_back(std::move(f));
        }
        cv.notify_all();
    }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(m);
                idle = q.empty();
                if (idle) cv.notify_all();
                cv.wait(g, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
                idle = false;
            }
            f();
        }
    }
    void sync()
    {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return q.empty() && idle; });
    }
};
struct MockEvent {
    std::mutex m;
    std::condition_variable cv;
    long issued = 0, done = 0;
};
typedef MockStream *hipStream_t;
typedef MockEvent *hipEvent_t;

// knobs of the test driver
inline std::atomic<int> &mock_fail_memcpy_after() { static std::atomic<int> n{-1}; return n; }      // >= 0: the n-th hipMemcpyAsync from now fails
inline std::atomic<int> &mock_copy_delay_us() { static std::atomic<int> us{0}; return us; }
inline std::mutex &mock_mutex() { static std::mutex m; return m; }

inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : (e == hipErrorOutOfMemory ? "out of memory" : "mock failure"); }
// four mock devices; the current device is per host thread, as in HIP
inline int &mock_current_device() { static thread_local int d = 0; return d; }
inline hipError_t hipSetDevice(int d) { if (d < 0 || d >= 4) return hipErrorInvalidValue; mock_current_device() = d; return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = mock_current_device(); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static const unsigned hipHostRegisterDefault = 0, hipHostRegisterMapped = 2;
// a page-locked row's device-visible address: the mock's "device" is the host
inline hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
inline int &mock_registered() { static int n = 0; return n; }
// page-locked rows -> copies issued on them that have not completed yet.  Unregistering a row with a copy pending is the
// bug the real runtime only hides by synchronising every stream inside hipHostUnregister: counted as a violation here.
inline std::map<const void *, int> &mock_registered_rows() { static std::map<const void *, int> m; return m; }
inline int &mock_unregister_violations() { static int n = 0; return n; }
// every third registration "fails" (already registered / shared page): the caller must fall back to a plain copy
inline hipError_t hipHostRegister(void *p, size_t, unsigned)
{
    std::lock_guard<std::mutex> g(mock_mutex());
    static int calls = 0;
    if (++calls % 3 == 0) return hipErrorInvalidValue;
    mock_registered()++;
    mock_registered_rows()[p] = 0;
    return hipSuccess;
}
inline hipError_t hipHostUnregister(void *p)
{
    std::lock_guard<std::mutex> g(mock_mutex());
    mock_registered()--;
    auto it = mock_registered_rows().find(p);
    if (it != mock_registered_rows().end()) {
        if (it->second > 0) mock_unregister_violations()++;
        mock_registered_rows().erase(it);
    }
    return hipSuccess;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new MockStream; return hipSuccess; }
inline int &mock_masked_streams() { static int n = 0; return n; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *)
{
    {
        std::lock_guard<std::mutex> g(mock_mutex());
        mock_masked_streams()++;
    }
    *s = new MockStream;
    return hipSuccess;
}
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t s) { s->sync(); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new MockEvent; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    long gen;
    {
        std::lock_guard<std::mutex> g(e->m);
        gen = ++e->issued;
    }
    s->push([e, gen] {
        {
            std::lock_guard<std::mutex> g(e->m);
            if (e->done < gen) e->done = gen;
        }
        e->cv.notify_all();
    });
    return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e)
{
    std::unique_lock<std::mutex> g(e->m);
    const long gen = e->issued;                              // an event never recorded is "complete", as in HIP
    e->cv.wait(g, [&] { return e->done >= gen; });
    return hipSuccess;
}
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }     // the mock keeps no clock
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    long gen;
    {
        std::lock_guard<std::mutex> g(e->m);
        gen = e->issued;                                     // waits for the record issued BEFORE this call only
    }
    s->push([e, gen] {
        std::unique_lock<std::mutex> g(e->m);
        e->cv.wait(g, [&] { return e->done >= gen; });
    });
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{
    {
        std::lock_guard<std::mutex> g(mock_mutex());
        if (mock_fail_memcpy_after() == 0) {
            mock_fail_memcpy_after() = -1;
            return hipErrorInvalidValue;
        }
        if (mock_fail_memcpy_after() > 0) mock_fail_memcpy_after()--;
    }
    const int us = mock_copy_delay_us();
    const void *host = nullptr;                              // the page-locked row this copy reads or writes, if any
    {
        std::lock_guard<std::mutex> g(mock_mutex());
        for (const void *cand : {(const void *)dst, src}) {
            auto it = mock_registered_rows().find(cand);
            if (it != mock_registered_rows().end()) { it->second++; host = cand; break; }
        }
    }
    s->push([=] {
        if (us) std::this_thread::sleep_for(std::chrono::microseconds(us));
        std::memcpy(dst, src, n);
        if (host) {
            std::lock_guard<std::mutex> g(mock_mutex());
            auto it = mock_registered_rows().find(host);
            if (it != mock_registered_rows().end()) it->second--;
        }
    });
    return hipSuccess;
}
// test-only: run host code in stream order (stands in for a kernel launch)
inline void mockLaunch(hipStream_t s, std::function<void()> f) { s->push(std::move(f)); }
