// ubench_f64.hip -- f64 VALU issue / dependent-chain cost on gfx950 (design input for the GC-ADPCM coefficient kernel,
// whose ordered bucket sums are chains of dependent v_add_f64): ns per wave64 instruction for 1 / 2 / 4 independent
// chains at 1 / 2 / 4 waves per SIMD, for add, mul + add (no contraction), divide, and the ordered_sum pattern
// (ds_read_b128 + two dependent adds).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_f64.hip -o tools/variants/ubench_f64 && tools/variants/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void k(double *out, int iters, double seed)
{
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = seed * i;
    __syncthreads();
    double a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x + c;
    double m = seed * 1.0000001, b = seed + 0.77;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32 / CHAINS; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = a[c] + b;                              // v_add_f64
                else if (KIND == 1) a[c] = a[c] * m + b;                     // v_mul_f64, v_add_f64
                else if (KIND == 2) a[c] = b / a[c];                         // IEEE divide
                else {                                                       // ordered_sum: one b128 read feeds two dependent adds
                    const double2 v = *reinterpret_cast<const double2 *>(&s[((i * 32 + u * CHAINS + c) * 2) & 1022]);
                    a[c] += v.x;
                    a[c] += v.y;
                }
                asm volatile("" : "+v"(a[c]));
            }
        }
    }
    double t = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) t += a[c];
    if (t == 1.2345) out[blockIdx.x] = t;
}

template <int CHAINS, int KIND>
void run(const char *name, double instr_per_elem)
{
    const int iters = KIND == 2 ? 200 : 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = 256 * 4 * waves_per_simd;
        double *d;
        hipMalloc(&d, blocks * sizeof(double));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<CHAINS, KIND><<<blocks, 64>>>(d, 10, 1.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<CHAINS, KIND><<<blocks, 64>>>(d, iters, 1.5);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double elems = (double)iters * 32;
        printf("%-34s chains=%d waves/SIMD=%d : %.2f ns per element per wave; per SIMD %.2f ns per element (= %.1f cycles at 2.4 GHz; %.1f per instruction)\n",
               name, CHAINS, waves_per_simd, ms * 1e6 / elems, ms * 1e6 / elems / waves_per_simd, ms * 1e6 / elems / waves_per_simd * 2.4,
               ms * 1e6 / elems / waves_per_simd * 2.4 / instr_per_elem);
        hipFree(d);
    }
}

int main()
{
    run<1, 0>("add_f64 dependent", 1);
    run<2, 0>("add_f64 2 chains", 1);
    run<4, 0>("add_f64 4 chains", 1);
    run<1, 1>("mul+add f64 dependent", 2);
    run<4, 1>("mul+add f64 4 chains", 2);
    run<1, 2>("div f64 dependent", 1);
    run<4, 2>("div f64 4 chains", 1);
    run<1, 3>("b128 read + 2 adds dependent", 3);
    run<2, 3>("b128 read + 2 adds 2 chains", 3);
    return 0;
}
