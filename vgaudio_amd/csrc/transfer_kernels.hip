// transfer_kernels.hip -- the host pipeline's rows over PCIe by a kernel instead of one hipMemcpyAsync per row (round 6).
//
// What it replaces: host_pipeline.hpp's direct mode page-locks the caller's rows (the managed arrays of
// GcAdpcmFormat.EncodeFromPcm16 / CriAdxFormat / CriHcaFormat, VGAudio/Formats/GcAdpcm/GcAdpcmFormat.cs:58-74 -- one array per
// channel, and one per FILE under Cli/Batch.cs:24-25) and moved each with its own copy.  The copy engine idles ~11 us between
// two such copies: 45 GB/s for the 10 008 rows of a ragged batch against 57 GB/s when the compute units fetch the same
// page-locked rows themselves (tools/bench_h2d_gather.hip, profiles/r06_h_h2d_gather.json.log).  One launch moves a chunk's
// pieces (Job::TransferPiece: at most 256 KB of one row, device-visible addresses on both sides); a workgroup takes a piece
// at a time, 16-byte accesses at whatever byte alignment the caller's rows have, four loads in flight per thread.  The launch
// runs on a stream whose CU mask reserves its compute units (host_pipeline.hpp: Job::transfer_cus) -- next to kernels that
// hold every CU it would otherwise start when their workgroups end.
#include "common.hpp"
#include "host_pipeline.hpp"

namespace vga {

namespace {

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
constexpr int TRANSFER_THREADS = 256;
constexpr int TRANSFER_UNROLL = 4;

__global__ __launch_bounds__(TRANSFER_THREADS) void transfer_pieces_kernel(const pipe::Job::TransferPiece *__restrict__ pieces, int n)
{
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const char *src = static_cast<const char *>(pieces[i].src);
        char *dst = static_cast<char *>(pieces[i].dst);
        const unsigned bytes = pieces[i].bytes;
        const unsigned n16 = bytes / 16;
        for (unsigned k = threadIdx.x; k < n16; k += TRANSFER_THREADS * TRANSFER_UNROLL) {
            u32x4_a1 v[TRANSFER_UNROLL];
#pragma unroll
            for (int u = 0; u < TRANSFER_UNROLL; u++)
                if (k + TRANSFER_THREADS * u < n16) v[u] = *reinterpret_cast<const u32x4_a1 *>(src + (size_t)(k + TRANSFER_THREADS * u) * 16);
#pragma unroll
            for (int u = 0; u < TRANSFER_UNROLL; u++)
                if (k + TRANSFER_THREADS * u < n16) *reinterpret_cast<u32x4_a1 *>(dst + (size_t)(k + TRANSFER_THREADS * u) * 16) = v[u];
        }
        const unsigned tail = bytes - n16 * 16;             // a row's last odd bytes
        if (threadIdx.x < tail) dst[n16 * 16 + threadIdx.x] = src[n16 * 16 + threadIdx.x];
    }
}

}  // namespace

// pieces: page-locked, device-visible (host_pipeline.hpp keeps the table alive until its run() returns)
int launch_transfer(const pipe::Job::TransferPiece *pieces, int n, hipStream_t stream)
{
    if (n <= 0) return VGA_OK;
    void *dev_view = nullptr;
    VGA_HIP_TRY(hipHostGetDevicePointer(&dev_view, const_cast<pipe::Job::TransferPiece *>(pieces), 0));
    // 64 workgroups saturate the link (57 GB/s with 64 or 256, 53 GB/s with 1024: bench_h2d_gather.hip) and fit the sixteen
    // compute units the pipeline reserves for them
    const int grid = n < 64 ? n : 64;
    hipLaunchKernelGGL(transfer_pieces_kernel, dim3(grid), dim3(TRANSFER_THREADS), 0, stream,
                       static_cast<const pipe::Job::TransferPiece *>(dev_view), n);
    VGA_HIP_TRY(hipGetLastError());
    return VGA_OK;
}

}  // namespace vga
