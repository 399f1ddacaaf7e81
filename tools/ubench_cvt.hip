// ubench_cvt.hip -- what the ADX quantiser's f64 step costs on gfx950: (int)((double)raw * gain) with the conversion
// instructions (v_cvt_f64_i32, v_mul_f64, v_cvt_i32_f64), with the int -> double conversion done as a bit pattern plus
// one v_add_f64 (2^52 + 2^31 + raw is exact), and an integer-only chain of the same length for comparison.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_cvt.hip -o tools/variants/ubench_cvt && tools/variants/ubench_cvt
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void k(int *out, int iters, double gain, int seed)
{
    int a[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = seed + threadIdx.x * 7 + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32 / CHAINS; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (KIND == 0) a[c] = (int)((double)a[c] * gain) + 3;                               // cvt, mul, cvt, add
                else if (KIND == 1) {                                                               // xor, add_f64, mul, cvt, add
                    const double d = __hiloint2double(0x43300000, a[c] ^ (int)0x80000000) - 4503601774854144.0;
                    a[c] = (int)(d * gain) + 3;
                } else if (KIND == 2) a[c] = (int)(double)a[c] + 3;                                 // cvt, cvt, add
                else a[c] = ((a[c] * 3) ^ 5) + 3;                                                   // mul_lo? (3 int ops)
                asm volatile("" : "+v"(a[c]));
            }
        }
    }
    int t = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) t += a[c];
    if (t == 12345) out[blockIdx.x] = t;
}

template <int CHAINS, int KIND>
void run(const char *name)
{
    const int iters = 2000;
    for (int waves_per_simd : {1, 2}) {
        const int blocks = 256 * 4 * waves_per_simd;
        int *d;
        hipMalloc(&d, blocks * sizeof(int));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<CHAINS, KIND><<<blocks, 64>>>(d, 10, 0.37, 5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<CHAINS, KIND><<<blocks, 64>>>(d, iters, 0.37, 5);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double elems = (double)iters * 32;
        printf("%-44s chains=%d waves/SIMD=%d : per SIMD %.2f ns per element\n", name, CHAINS, waves_per_simd, ms * 1e6 / elems / waves_per_simd);
        hipFree(d);
    }
}

int main()
{
    run<1, 0>("cvt_f64_i32, mul_f64, cvt_i32_f64, add");
    run<4, 0>("cvt_f64_i32, mul_f64, cvt_i32_f64, add");
    run<1, 1>("xor, add_f64, mul_f64, cvt_i32_f64, add");
    run<4, 1>("xor, add_f64, mul_f64, cvt_i32_f64, add");
    run<1, 2>("cvt_f64_i32, cvt_i32_f64, add");
    run<4, 2>("cvt_f64_i32, cvt_i32_f64, add");
    run<1, 3>("mul_lo, xor, add (integer)");
    run<4, 3>("mul_lo, xor, add (integer)");
    return 0;
}
