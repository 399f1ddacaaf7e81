// ubench_valu.hip -- VALU issue/latency microbenchmark for gfx950 (design input for the
// GC-ADPCM encode kernel): cycles per wave64 instruction for dependent vs independent
// integer chains at 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void k(long long *out, int iters, int seed)
{
    int a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x + c;
    int m = seed | 3, b = seed + 77;
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32 / CHAINS; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = __mul24(a[c], m) + b;                 // v_mad_i32_i24
                else if (KIND == 1) a[c] = min(max(a[c] + b, -32768), 32767); // add + med3
                else if (KIND == 2) a[c] = (int)(float)a[c] + b;            // cvt, cvt, add
                else if (KIND == 3) a[c] = (a[c] << 3) + b;                 // v_lshl_add_u32
                else if (KIND == 5) {                                        // the decoders' recurrence: mad, shift, clamp
                    a[c] = min(max((__mul24(a[c], m) + b) >> 11, -32768), 32767);
                } else if (KIND == 6) {                                      // the same with dot2 + saturating pack
                    int t;
                    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(t) : "v"(a[c]), "v"(m), "v"(b));
                    t >>= 11;
                    asm volatile("v_cvt_pk_i16_i32 %0, %1, %2" : "=v"(a[c]) : "v"(b), "v"(t));
                }
                else { double d = (double)(float)a[c] + 0.4999999; a[c] = (int)d + b; } // f64 detour
                asm volatile("" : "+v"(a[c]));
            }
        }
    }
    long long t1 = clock64();
    int s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += a[c];
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (s == 123456789 ? 1 : 0);
}

template <int CHAINS, int KIND>
void run(const char *name, int ops_per_elem)
{
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = 256 * 4 * waves_per_simd;
        long long *d;
        hipMalloc(&d, blocks * sizeof(long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<CHAINS, KIND><<<blocks, 64>>>(d, 10, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<CHAINS, KIND><<<blocks, 64>>>(d, iters, 1);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), d, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= blocks;
        const double instr = (double)iters * 32 * ops_per_elem;
        printf("%-28s chains=%d waves/SIMD=%d : %.2f clk64-ticks/instr per wave, wall %.3f ms -> %.2f ns/instr/wave\n",
               name, CHAINS, waves_per_simd, avg / instr, ms, ms * 1e6 / instr);
        hipFree(d);
    }
}

int main()
{
    run<1, 0>("mad_i24 dependent", 1);
    run<4, 0>("mad_i24 4 chains", 1);
    run<8, 0>("mad_i24 8 chains", 1);
    run<1, 1>("add+med3 dependent", 2);
    run<4, 1>("add+med3 4 chains", 2);
    run<1, 2>("cvt+cvt+add dependent", 3);
    run<4, 2>("cvt+cvt+add 4 chains", 3);
    run<1, 3>("lshl_add dependent", 1);
    run<1, 5>("mad+ashr+med3 dependent", 3);
    run<2, 5>("mad+ashr+med3 2 chains", 3);
    run<1, 6>("dot2+ashr+cvt_pk dependent", 3);
    run<2, 6>("dot2+ashr+cvt_pk 2 chains", 3);
    run<1, 4>("f64 detour dependent", 5);
    run<4, 4>("f64 detour 4 chains", 5);
    return 0;
}
