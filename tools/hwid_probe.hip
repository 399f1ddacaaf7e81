// hwid_probe: where do the waves of a (threads, lds) workgroup land?  Each wave records HW_ID
// (CU / SIMD / wave slot) while all workgroups are resident (they spin until everyone has arrived
// or a timeout passes).  Built by tools/build_variants.sh-style command into tools/variants/libhwid.so.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void probe_kernel(uint32_t *out, int *arrived, int total_wgs, int lds_words)
{
    extern __shared__ int lds[];
    if (threadIdx.x < (unsigned)lds_words) lds[threadIdx.x] = threadIdx.x;
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[2 * w] = hwid;
        out[2 * w + 1] = xcc;
    }
    if (threadIdx.x == 0) atomicAdd(arrived, 1);
    // keep the workgroup resident for a while so the placement reflects a full machine
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000000) {          // 20 ms at 100 MHz
        if (__atomic_load_n(arrived, __ATOMIC_RELAXED) >= total_wgs) break;
    }
    __syncthreads();
    if (lds_words > 0 && lds[0] == 12345) out[0] = 0;
}

extern "C" int hwid_probe(int blocks, int threads, int lds_bytes, uint32_t *host_out)
{
    uint32_t *d_out; int *d_arr;
    const int waves = blocks * (threads / 64);
    if (hipMalloc(&d_out, waves * 8) != hipSuccess) return -1;
    if (hipMalloc(&d_arr, 4) != hipSuccess) return -1;
    hipMemset(d_arr, 0, 4);
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(threads), lds_bytes, 0, d_out, d_arr, blocks, lds_bytes / 4 > threads ? threads : lds_bytes / 4);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    hipMemcpy(host_out, d_out, waves * 8, hipMemcpyDeviceToHost);
    hipFree(d_out); hipFree(d_arr);
    return 0;
}
