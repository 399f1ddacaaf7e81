// bench_h2d_batch.hip -- do the ragged host calls' 10 008 uploads have to leave the copy engine idle ~11 us each?  The files of
// bench.py's mixed-lengths set (log-uniform 1-120 s of 48 kHz PCM, 23.6 GB; --rows / --gb shrink it) as separately malloc'ed,
// page-locked rows, host -> device with (a) one hipMemcpyAsync per row on one stream (what host_pipeline.hpp does), (b)
// hipMemcpyBatchAsync over 512 rows at a time (ROCm >= 7.1's runtime; the 7.0 runtime PyTorch brings into every Python process
// here does not export it, which is why the library cannot use it where it is measured), (c) one row per call alternating over
// two streams.  A standalone program: it links the ROCm 7.2 runtime of /opt/rocm.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/bench_h2d_batch.hip -o tools/variants/bench_h2d_batch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const double target_gb = argc > 1 ? std::atof(argv[1]) : 23.6;
    std::mt19937_64 rng(0xBA7C4);
    std::uniform_real_distribution<double> u(std::log(48000.0), std::log(120 * 48000.0));
    std::vector<size_t> bytes;
    size_t total = 0;
    while ((double)total < target_gb * 1e9) {
        const size_t n = (size_t)std::exp(u(rng)) * 2;
        bytes.push_back(n);
        total += (n + 15) / 16 * 16;
    }
    const int rows = (int)bytes.size();
    std::vector<char *> host(rows);
    std::vector<size_t> off(rows);
    size_t at = 0;
    for (int r = 0; r < rows; r++) {
        host[r] = static_cast<char *>(std::malloc(bytes[r]));
        std::memset(host[r], r & 0xff, bytes[r]);
        CHECK(hipHostRegister(host[r], bytes[r], hipHostRegisterDefault));
        off[r] = at;
        at += (bytes[r] + 15) / 16 * 16;
    }
    char *dev;
    CHECK(hipMalloc(&dev, at + 64));
    const double gb = (double)total / 1e9;
    std::printf("{\"rows\": %d, \"GB\": %.2f", rows, gb);
    hipStream_t s[2];
    for (auto &q : s) CHECK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    auto report = [&](const char *name, double t) { std::printf(", \"%s\": {\"ms\": %.1f, \"GBps\": %.1f}", name, t * 1e3, gb / t); std::fflush(stdout); };
    for (int rep = 0; rep < 3; rep++) {                                   // (a)
        const double t0 = now();
        for (int r = 0; r < rows; r++) CHECK(hipMemcpyAsync(dev + off[r], host[r], bytes[r], hipMemcpyHostToDevice, s[0]));
        CHECK(hipStreamSynchronize(s[0]));
        if (rep) report(rep == 1 ? "one_copy_per_row_1" : "one_copy_per_row_2", now() - t0);
    }
    for (int rep = 0; rep < 3; rep++) {                                   // (b)
        const double t0 = now();
        std::vector<void *> d(rows), h(rows);
        for (int r = 0; r < rows; r++) { d[r] = dev + off[r]; h[r] = host[r]; }
        hipError_t e = hipSuccess;
        for (int r = 0; r < rows && e == hipSuccess; r += 512) {
            const size_t n = (size_t)std::min(512, rows - r);
            size_t fail = 0;
            e = hipMemcpyBatchAsync(d.data() + r, h.data() + r, bytes.data() + r, n, nullptr, nullptr, 0, &fail, s[0]);
        }
        if (e != hipSuccess) { std::printf(", \"batch_of_512_rows\": \"%s\"", hipGetErrorString(e)); (void)hipGetLastError(); break; }
        CHECK(hipStreamSynchronize(s[0]));
        if (rep) report(rep == 1 ? "batch_of_512_rows_1" : "batch_of_512_rows_2", now() - t0);
    }
    for (int rep = 0; rep < 4; rep++) {                                   // (c)
        const double t0 = now();
        for (int r = 0; r < rows; r++) CHECK(hipMemcpyAsync(dev + off[r], host[r], bytes[r], hipMemcpyHostToDevice, s[r & 1]));
        CHECK(hipStreamSynchronize(s[0]));
        CHECK(hipStreamSynchronize(s[1]));
        if (rep) { char name[40]; std::snprintf(name, sizeof name, "alternating_two_streams_%d", rep); report(name, now() - t0); }
    }
    std::printf("}\n");
    return 0;
}
