// im_results.hip -- the realign results' trip to the host.
//
// attempt_pe_alignment (src/alignment.c:764-799) returns NULL for most candidates; the host reads a candidate's
// 512-byte im_read_result only when it holds realigned evidence (src/indelminer.c:494-502).  So the records with
// status == IM_ST_EVIDENCE and n_ev > 0 are packed into one array, and per candidate only its status and its place in
// that array cross PCIe: 8 bytes instead of 512 for the reads without evidence.

#include "im_device.hpp"

namespace im {
namespace {

static_assert(sizeof(im_read_result) == 512, "a record is 64 lanes x 8 bytes");

__global__ __launch_bounds__(256) void compact_results_kernel(const im_read_result* __restrict__ res, int32_t n_cap, const int32_t* __restrict__ n_dev,
                                                             int32_t* __restrict__ status, int32_t* __restrict__ slot,
                                                             im_read_result* __restrict__ compact, int32_t* __restrict__ count)
{
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int base = wave * 64; base < n; base += n_waves * 64) {
        const int c = base + lane;
        int st = 0, nev = 0;
        if (c < n) { st = res[c].status; nev = res[c].n_ev; }
        const bool evid = st == IM_ST_EVIDENCE && nev > 0;
        uint64_t m = __ballot(evid);
        int first = 0;
        if (lane == 0 && m) first = atomicAdd(count, __popcll(m));
        first = __shfl(first, 0);
        const int mine = evid ? first + __popcll(m & ((1ull << lane) - 1ull)) : -1;
        if (c < n) { status[c] = st; slot[c] = mine; }
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const uint2* src = reinterpret_cast<const uint2*>(res + base + j);
            uint2* dst = reinterpret_cast<uint2*>(compact + __shfl(mine, j));
            dst[lane] = src[lane];
        }
    }
}

}  // namespace

hipError_t launch_compact_results(const im_read_result* res, int32_t n_cap, const int32_t* n_dev, int32_t* status, int32_t* slot,
                                  im_read_result* compact, int32_t* count, int n_cu, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int32_t), stream);
    if (e != hipSuccess || n_cap <= 0) return e;
    int64_t blocks = ((int64_t)n_cap + 255) / 256;
    if (blocks > (int64_t)n_cu * 8) blocks = (int64_t)n_cu * 8;
    hipLaunchKernelGGL(compact_results_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, res, n_cap, n_dev, status, slot, compact, count);
    return hipGetLastError();
}

}  // namespace im
