// bench_h2d_modes.hip -- how should pageable caller rows reach HBM on this box?  Times, for R separately malloc'ed rows of
// 5.76 MB (one 60 s channel), host -> device with: (a) hipMemcpyAsync straight from the pageable rows, one stream;
// (b) the same from T threads with a stream each; (c) hipHostRegister per row (T threads), async copies from the
// registered rows, unregister; (d) staging through a pinned ring with T memcpy threads (what host_pipeline.hpp does).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/bench_h2d_modes.hip -o tools/variants/bench_h2d_modes -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? std::atoi(argv[1]) : 1024;
    const int T = argc > 2 ? std::atoi(argv[2]) : 4;
    const size_t row_bytes = 2880000 * 2;
    std::vector<char *> host(rows);
    for (int r = 0; r < rows; r++) {
        host[r] = static_cast<char *>(std::malloc(row_bytes));
        std::memset(host[r], r & 0xff, row_bytes);
    }
    char *dev;
    CHECK(hipMalloc(&dev, (size_t)rows * row_bytes));
    const double gb = (double)rows * row_bytes / 1e9;
    auto report = [&](const char *name, double s) { std::printf("\"%s\": {\"ms\": %.1f, \"GBps\": %.1f}, ", name, s * 1e3, gb / s); };
    std::printf("{\"rows\": %d, \"threads\": %d, \"GB\": %.2f, ", rows, T, gb);
    {   // (a)
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            for (int r = 0; r < rows; r++) CHECK(hipMemcpyAsync(dev + (size_t)r * row_bytes, host[r], row_bytes, hipMemcpyHostToDevice, s));
            CHECK(hipStreamSynchronize(s));
            if (rep) report("pageable_async_1_stream", now() - t0);
        }
        CHECK(hipStreamDestroy(s));
    }
    auto threaded = [&](const char *name, auto body) {
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] { CHECK(hipSetDevice(0)); body(t); });
            for (auto &x : th) x.join();
            if (rep) report(name, now() - t0);
        }
    };
    threaded("pageable_async_T_streams", [&](int t) {
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int r = t; r < rows; r += T) CHECK(hipMemcpyAsync(dev + (size_t)r * row_bytes, host[r], row_bytes, hipMemcpyHostToDevice, s));
        CHECK(hipStreamSynchronize(s));
        CHECK(hipStreamDestroy(s));
    });
    threaded("register_per_row_T_threads", [&](int t) {
        hipStream_t s;
        CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (int r = t; r < rows; r += T) {
            CHECK(hipHostRegister(host[r], row_bytes, 0));
            CHECK(hipMemcpyAsync(dev + (size_t)r * row_bytes, host[r], row_bytes, hipMemcpyHostToDevice, s));
        }
        CHECK(hipStreamSynchronize(s));
        for (int r = t; r < rows; r += T) CHECK(hipHostUnregister(host[r]));
        CHECK(hipStreamDestroy(s));
    });
    {   // (d) pinned ring, 3 slots of 4 rows per thread
        constexpr int SR = 4, R = 3;
        char *ring;
        CHECK(hipHostMalloc(&ring, (size_t)T * R * SR * row_bytes, 0));
        threaded("pinned_ring_T_threads", [&](int t) {
            hipStream_t s;
            CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            hipEvent_t ev[R];
            for (auto &evt : ev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
            int used = 0;
            for (int g = t * SR; g < rows; g += T * SR) {
                const int n = std::min(SR, rows - g), sl = used % R;
                if (used >= R) CHECK(hipEventSynchronize(ev[sl]));
                char *slot = ring + ((size_t)t * R + sl) * SR * row_bytes;
                for (int i = 0; i < n; i++) std::memcpy(slot + (size_t)i * row_bytes, host[g + i], row_bytes);
                CHECK(hipMemcpyAsync(dev + (size_t)g * row_bytes, slot, (size_t)n * row_bytes, hipMemcpyHostToDevice, s));
                CHECK(hipEventRecord(ev[sl], s));
                used++;
            }
            CHECK(hipStreamSynchronize(s));
            for (auto &evt : ev) CHECK(hipEventDestroy(evt));
            CHECK(hipStreamDestroy(s));
        });
        CHECK(hipHostFree(ring));
    }
    std::printf("\"done\": 1}\n");
    return 0;
}
