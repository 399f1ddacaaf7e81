// bench_h2d_gather.hip -- VERDICT r05 item 4: can the CUs pull a ragged batch's rows over PCIe faster than 10 008 hipMemcpyAsync
// calls do?  The rows of bench.py's mixed-lengths set (log-uniform 1-120 s of 48 kHz PCM; --gb shrinks it) as separately
// malloc'ed, page-locked rows (hipHostRegister, as host_pipeline.hpp does), host -> device
//   (a) one hipMemcpyAsync per row on one stream (what the library does now),
//   (b) ONE gather kernel over a device table of (host-mapped pointer, bytes, destination offset): a workgroup per 64 KB piece
//       of a row, 16-byte loads, `waves` x `unroll` loads in flight per CU,
//   (c) the same for device -> host (a scatter kernel storing into the registered rows) against one copy per row.
// Prints GB/s of each.  Standalone: links the ROCm 7.2 runtime of /opt/rocm.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/bench_h2d_gather.hip -o tools/variants/bench_h2d_gather
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Piece { const char *src; char *dst; unsigned bytes; unsigned pad; };   // <= 64 KB of one row

template <int UNROLL, bool TO_HOST>
__global__ __launch_bounds__(256) void gather_kernel(const Piece *__restrict__ pieces, int n)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const Piece p = pieces[i];
        const uint4 *s = reinterpret_cast<const uint4 *>(p.src);
        uint4 *d = reinterpret_cast<uint4 *>(p.dst);
        const unsigned n16 = p.bytes / 16;
        for (unsigned k = threadIdx.x; k < n16; k += 256 * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (k + 256 * u < n16) v[u] = s[k + 256 * u];
#pragma unroll
            for (int u = 0; u < UNROLL; u++)
                if (k + 256 * u < n16) d[k + 256 * u] = v[u];
        }
        // (rows are multiples of 16 bytes here; the library would move a row's last odd bytes with byte accesses)
    }
}

// a kernel that fills the chip the way the persistent encoder does: every CU's registers and LDS taken, for `ms` milliseconds
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(3, 3))) void busy_kernel(long long cycles, float *sink)
{
    __shared__ float pad[9000];                        // ~36 KB: four workgroups per CU
    pad[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float acc[96];
#pragma unroll
    for (int i = 0; i < 96; i++) acc[i] = pad[(threadIdx.x + i) % 192];
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {
#pragma unroll
        for (int i = 0; i < 96; i++) acc[i] = acc[i] * 1.0001f + acc[(i + 1) % 96];
    }
    float t = 0;
#pragma unroll
    for (int i = 0; i < 96; i++) t += acc[i];
    if (t == 12345.678f) sink[0] = t;
}

int main(int argc, char **argv)
{
    const double target_gb = argc > 1 ? std::atof(argv[1]) : 8.0;
    std::mt19937_64 rng(0xBA7C4);
    std::uniform_real_distribution<double> u(std::log(48000.0), std::log(120 * 48000.0));
    std::vector<size_t> bytes;
    size_t total = 0;
    while ((double)total < target_gb * 1e9) {
        const size_t n = ((size_t)std::exp(u(rng)) * 2 + 15) / 16 * 16;
        bytes.push_back(n);
        total += n;
    }
    const int rows = (int)bytes.size();
    std::vector<char *> host(rows), mapped(rows);
    std::vector<size_t> off(rows);
    size_t at = 0;
    for (int r = 0; r < rows; r++) {
        void *p = nullptr;
        if (posix_memalign(&p, 4096, bytes[r])) return 2;
        host[r] = static_cast<char *>(p);
        std::memset(host[r], (r * 7 + 1) & 0xff, bytes[r]);
        CHECK(hipHostRegister(host[r], bytes[r], hipHostRegisterMapped));
        void *dp = nullptr;
        CHECK(hipHostGetDevicePointer(&dp, host[r], 0));
        mapped[r] = static_cast<char *>(dp);
        off[r] = at;
        at += bytes[r];
    }
    char *dev;
    CHECK(hipMalloc(&dev, at + 64));
    CHECK(hipMemset(dev, 0, at + 64));
    const double gb = (double)total / 1e9;
    std::printf("{\"rows\": %d, \"GB\": %.2f", rows, gb);
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto report = [&](const char *name, double t) { std::printf(", \"%s\": {\"ms\": %.1f, \"GBps\": %.1f}", name, t * 1e3, gb / t); std::fflush(stdout); };
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        for (int r = 0; r < rows; r++) CHECK(hipMemcpyAsync(dev + off[r], host[r], bytes[r], hipMemcpyHostToDevice, s));
        CHECK(hipStreamSynchronize(s));
        if (rep) report(rep == 1 ? "h2d_one_copy_per_row_1" : "h2d_one_copy_per_row_2", now() - t0);
    }
    // the piece table
    constexpr unsigned PIECE = 64 * 1024;
    std::vector<Piece> up, down;
    for (int r = 0; r < rows; r++)
        for (size_t o = 0; o < bytes[r]; o += PIECE) {
            const unsigned nb = (unsigned)std::min<size_t>(PIECE, bytes[r] - o);
            up.push_back({mapped[r] + o, dev + off[r] + o, nb, 0});
            down.push_back({dev + off[r] + o, mapped[r] + o, nb, 0});
        }
    Piece *d_up, *d_down;
    CHECK(hipMalloc(&d_up, up.size() * sizeof(Piece)));
    CHECK(hipMalloc(&d_down, down.size() * sizeof(Piece)));
    CHECK(hipMemcpy(d_up, up.data(), up.size() * sizeof(Piece), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_down, down.data(), down.size() * sizeof(Piece), hipMemcpyHostToDevice));
    CHECK(hipMemset(dev, 0, at + 64));
    for (int grid : {64, 256, 1024, 4096}) {
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            hipLaunchKernelGGL((gather_kernel<4, false>), dim3(grid), dim3(256), 0, s, d_up, (int)up.size());
            CHECK(hipStreamSynchronize(s));
            char name[64];
            std::snprintf(name, sizeof name, "h2d_gather_kernel_grid%d_%d", grid, rep);
            if (rep) report(name, now() - t0);
        }
    }
    // check what arrived
    {
        std::vector<char> back(std::min<size_t>(at, 64 << 20));
        CHECK(hipMemcpy(back.data(), dev, back.size(), hipMemcpyDeviceToHost));
        size_t bad = 0, pos = 0;
        for (int r = 0; r < rows && pos + bytes[r] <= back.size(); pos += bytes[r], r++)
            for (size_t i = 0; i < bytes[r]; i += 4097) bad += back[pos + i] != (char)((r * 7 + 1) & 0xff);
        std::printf(", \"gather_mismatches\": %zu", bad);
    }
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        for (int r = 0; r < rows; r++) CHECK(hipMemcpyAsync(host[r], dev + off[r], bytes[r], hipMemcpyDeviceToHost, s));
        CHECK(hipStreamSynchronize(s));
        if (rep) report(rep == 1 ? "d2h_one_copy_per_row_1" : "d2h_one_copy_per_row_2", now() - t0);
    }
    for (int grid : {256, 1024}) {
        for (int rep = 0; rep < 2; rep++) {
            const double t0 = now();
            hipLaunchKernelGGL((gather_kernel<4, true>), dim3(grid), dim3(256), 0, s, d_down, (int)down.size());
            CHECK(hipStreamSynchronize(s));
            char name[64];
            std::snprintf(name, sizeof name, "d2h_scatter_kernel_grid%d_%d", grid, rep);
            if (rep) report(name, now() - t0);
        }
    }
    // (d) the gather next to a kernel that holds every CU: same stream priorities, then with CU masks (compute: all but the
    // first `reserve` mask bits, transfer: those bits)
    {
        hipDeviceProp_t prop;
        CHECK(hipGetDeviceProperties(&prop, 0));
        const int cus = prop.multiProcessorCount;
        float *sink;
        CHECK(hipMalloc(&sink, 4));
        const long long cycles = 100000000ll * 3 / 10;             // wall_clock64 ticks at 100 MHz: 0.3 s
        auto trial = [&](const char *name, hipStream_t cs, hipStream_t ts) {
            hipLaunchKernelGGL(busy_kernel, dim3(cus * 4), dim3(192), 0, cs, cycles, sink);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            const double t0 = now();
            hipLaunchKernelGGL((gather_kernel<4, false>), dim3(64), dim3(256), 0, ts, d_up, (int)up.size());
            CHECK(hipStreamSynchronize(ts));
            const double t1 = now();
            CHECK(hipStreamSynchronize(cs));
            std::printf(", \"%s\": {\"gather_ms\": %.1f, \"GBps\": %.1f, \"busy_kernel_ms_after\": %.1f}", name, (t1 - t0) * 1e3, gb / (t1 - t0),
                        (now() - t1) * 1e3);
            std::fflush(stdout);
        };
        hipStream_t c0, t0s;
        CHECK(hipStreamCreateWithFlags(&c0, hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&t0s, hipStreamNonBlocking));
        trial("next_to_busy_chip_no_masks", c0, t0s);
        for (int reserve : {8, 16}) {
            const int words = (cus + 31) / 32;
            std::vector<uint32_t> cmask(words, 0xFFFFFFFFu), tmask(words, 0u);
            for (int b = 0; b < reserve; b++) { cmask[b / 32] &= ~(1u << (b % 32)); tmask[b / 32] |= 1u << (b % 32); }
            hipStream_t cm, tm;
            CHECK(hipExtStreamCreateWithCUMask(&cm, (uint32_t)words, cmask.data()));
            CHECK(hipExtStreamCreateWithCUMask(&tm, (uint32_t)words, tmask.data()));
            char name[64];
            std::snprintf(name, sizeof name, "next_to_busy_chip_%d_cus_reserved", reserve);
            trial(name, cm, tm);
        }
    }
    std::printf("}\n");
    return 0;
}
