// bench_overlap.hip -- do host<->device copies overlap chip-filling kernels on this box, and under which issue patterns?
// (evidence for the host pipeline's shape; GPU box only)
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/bench_overlap.hip -o tools/variants/bench_overlap
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                            \
    do {                                                                                                    \
        hipError_t err_ = (x);                                                                              \
        if (err_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(err_)); std::exit(2); } \
    } while (0)

__global__ __launch_bounds__(256) void spin_kernel(unsigned *out, int iters)
{
    unsigned v = threadIdx.x + blockIdx.x * 977u;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x12345u) out[0] = v;                       // never true in practice; keeps the loop
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = (size_t)8 << 30, piece = (size_t)32 << 20;
    const int pieces = (int)(total / piece), groups = 8, per_group = pieces / groups;
    char *h = nullptr, *d = nullptr, *d2 = nullptr, *hout = nullptr;
    unsigned *dummy = nullptr;
    CHECK(hipHostMalloc(&h, total, hipHostMallocDefault));
    CHECK(hipHostMalloc(&hout, total / 4, hipHostMallocDefault));
    std::memset(h, 1, total);
    CHECK(hipMalloc(&d, total));
    CHECK(hipMalloc(&d2, total / 4));
    CHECK(hipMalloc(&dummy, 4096));
    hipStream_t sc, sk, sd, sm[4];
    CHECK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
    for (auto &s : sm) CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(groups), evk(groups);
    for (auto &e : ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : evk) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    // calibrate: kernels worth ~300 ms in `groups` batches of `per_batch` launches
    int iters = 200000;
    const int per_batch = 6;
    auto kernels = [&](hipStream_t s, int batches) {
        for (int b = 0; b < batches * per_batch; b++) hipLaunchKernelGGL(spin_kernel, dim3(4096), dim3(256), 0, s, dummy, iters);
    };
    kernels(sk, 1);
    CHECK(hipStreamSynchronize(sk));
    double t0 = now();
    kernels(sk, groups);
    CHECK(hipStreamSynchronize(sk));
    double tk = now() - t0;
    iters = (int)(iters * 0.300 / tk);
    t0 = now();
    kernels(sk, groups);
    CHECK(hipStreamSynchronize(sk));
    tk = now() - t0;

    auto sync_all = [&]() { CHECK(hipDeviceSynchronize()); };
    auto run = [&](const char *name, auto &&issue) {
        double best = 1e9, best_c = 0, best_k = 0;
        for (int rep = 0; rep < 3; rep++) {
            sync_all();
            const double a = now();
            double tc = 0, tkk = 0;
            issue(a, tc, tkk);
            sync_all();
            const double t = now() - a;
            if (t < best) { best = t; best_c = tc; best_k = tkk; }
        }
        std::printf("{\"case\": \"%s\", \"total_ms\": %.1f, \"copies_done_ms\": %.1f, \"kernels_done_ms\": %.1f}\n", name, best * 1e3, best_c * 1e3, best_k * 1e3);
        std::fflush(stdout);
    };
    std::printf("{\"kernels_alone_ms\": %.1f, \"pieces\": %d, \"piece_MB\": %zu}\n", tk * 1e3, pieces, piece >> 20);

    run("h2d_one_copy_alone", [&](double a, double &tc, double &) {
        CHECK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, sc));
        CHECK(hipStreamSynchronize(sc)); tc = now() - a; });
    run("h2d_pieces_one_stream_alone", [&](double a, double &tc, double &) {
        for (int i = 0; i < pieces; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
        CHECK(hipStreamSynchronize(sc)); tc = now() - a; });
    run("h2d_one_copy_with_kernels", [&](double a, double &tc, double &tkk) {
        CHECK(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, sc));
        kernels(sk, groups);
        CHECK(hipStreamSynchronize(sc)); tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    run("h2d_pieces_one_stream_with_kernels", [&](double a, double &tc, double &tkk) {
        for (int i = 0; i < pieces; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
        kernels(sk, groups);
        CHECK(hipStreamSynchronize(sc)); tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    run("h2d_pieces_four_streams_with_kernels", [&](double a, double &tc, double &tkk) {
        for (int i = 0; i < pieces; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sm[i & 3]));
        kernels(sk, groups);
        for (auto &s : sm) CHECK(hipStreamSynchronize(s));
        tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    // the pipeline's pattern: kernel batch g waits for copy group g (event), copies of group g+1 run under it
    run("pipeline_pattern_h2d_events", [&](double a, double &tc, double &tkk) {
        for (int g = 0; g < groups; g++) {
            for (int i = g * per_group; i < (g + 1) * per_group; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
            CHECK(hipEventRecord(ev[g], sc));
            CHECK(hipStreamWaitEvent(sk, ev[g], 0));
            kernels(sk, 1);
        }
        CHECK(hipStreamSynchronize(sc)); tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    // ... plus the downloads: group g's output (a quarter of the input size) leaves after kernel batch g
    run("pipeline_pattern_h2d_kernels_d2h", [&](double a, double &tc, double &tkk) {
        const size_t og = total / 4 / groups;
        for (int g = 0; g < groups; g++) {
            for (int i = g * per_group; i < (g + 1) * per_group; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
            CHECK(hipEventRecord(ev[g], sc));
            CHECK(hipStreamWaitEvent(sk, ev[g], 0));
            kernels(sk, 1);
            CHECK(hipEventRecord(evk[g], sk));
            CHECK(hipStreamWaitEvent(sd, evk[g], 0));
            CHECK(hipMemcpyAsync(hout + g * og, d2 + g * og, og, hipMemcpyDeviceToHost, sd));
        }
        CHECK(hipStreamSynchronize(sc)); tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    // the same, but each group's copies are issued only once the previous group's copies are done (a feeder that
    // blocks), so that the waits are enqueued when little else is
    run("pipeline_pattern_issue_late", [&](double a, double &tc, double &tkk) {
        for (int g = 0; g < groups; g++) {
            for (int i = g * per_group; i < (g + 1) * per_group; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
            CHECK(hipEventRecord(ev[g], sc));
            CHECK(hipStreamWaitEvent(sk, ev[g], 0));
            kernels(sk, 1);
            CHECK(hipEventSynchronize(ev[g]));
        }
        CHECK(hipStreamSynchronize(sc)); tc = now() - a;
        CHECK(hipStreamSynchronize(sk)); tkk = now() - a; });
    // downloads next to uploads and kernels: when are the downloads done?  (2 GiB in 8 MB pieces, no dependencies)
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));            // lo = least, hi = greatest priority (numerically lower)
    hipStream_t sd_hi, sk_lo;
    CHECK(hipStreamCreateWithPriority(&sd_hi, hipStreamNonBlocking, hi));
    CHECK(hipStreamCreateWithPriority(&sk_lo, hipStreamNonBlocking, lo));
    const size_t dpiece = (size_t)8 << 20;
    const int dpieces = (int)(total / 4 / dpiece);
    auto d2h_case = [&](const char *name, bool with_h2d, hipStream_t ks, hipStream_t ds) {
        double best = 1e9, best_d = 0, best_c = 0;
        for (int rep = 0; rep < 3; rep++) {
            sync_all();
            const double a = now();
            if (with_h2d)
                for (int i = 0; i < pieces; i++) CHECK(hipMemcpyAsync(d + i * piece, h + i * piece, piece, hipMemcpyHostToDevice, sc));
            kernels(ks, groups);
            for (int i = 0; i < dpieces; i++) CHECK(hipMemcpyAsync(hout + i * dpiece, d2 + i * dpiece, dpiece, hipMemcpyDeviceToHost, ds));
            CHECK(hipStreamSynchronize(ds));
            const double td = now() - a;
            CHECK(hipStreamSynchronize(sc));
            const double tc = now() - a;
            sync_all();
            const double t = now() - a;
            if (t < best) { best = t; best_d = td; best_c = tc; }
        }
        std::printf("{\"case\": \"%s\", \"total_ms\": %.1f, \"downloads_done_ms\": %.1f, \"uploads_done_ms\": %.1f}\n", name, best * 1e3, best_d * 1e3, best_c * 1e3);
        std::fflush(stdout);
    };
    d2h_case("d2h_with_kernels", false, sk, sd);
    d2h_case("d2h_with_kernels_and_h2d", true, sk, sd);
    d2h_case("d2h_high_priority_stream_with_kernels_and_h2d", true, sk, sd_hi);
    d2h_case("d2h_with_low_priority_kernels_and_h2d", true, sk_lo, sd);
    return 0;
}
