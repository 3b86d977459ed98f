// im_depth.hip -- region depth for the DP= field on gfx950.
//
// Replaces calculate_cov_params (src/shared.c:151-212), which re-opens the BAM, reloads the
// whole index and runs a samtools pileup for EVERY printed variant.  Pileup semantics kept
// (bam_pileup.c:67-143,238-265): a position counts a read iff the read passes
// the default mask (unmapped / secondary / QC-fail / duplicate are skipped) and its covering
// CIGAR op is M, = or X.  The host hands over those match segments once per contig;
//   depth_scatter   +1 / -1 into a difference array (device atomics)
//   depth_scan_*    three-phase prefix sum -> depth per position (HBM streaming)
//   depth_query     one wave per printed variant: sum over [start-lw-1, stop+rw+1)
// The host takes floor(sum / length) like the reference (src/shared.c:205).

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kScanBlock = 1024;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;
constexpr int kScanGroup = 32;      // tiles that count their arrival into one word (depth_scan_tiled_kernel)

__global__ __launch_bounds__(256) void depth_scatter_kernel(int32_t n_seg, const int32_t* __restrict__ start,
                                                           const int32_t* __restrict__ len, int64_t clen, int32_t* __restrict__ diff)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_seg; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t a = start[i], b = (int64_t)start[i] + len[i];
        if (a < 0) a = 0;
        if (b > clen) b = clen;
        if (a >= b) continue;
        atomicAdd(&diff[a], 1);
        atomicAdd(&diff[b], -1);
    }
}

// phase 1: inclusive scan inside a tile of 8192 elements, tile total to sums[]
__global__ __launch_bounds__(kScanBlock) void depth_scan_tiles_kernel(int32_t* __restrict__ data, int64_t n, int32_t* __restrict__ sums)
{
    __shared__ int32_t wsum[kScanBlock / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
    int32_t v[kScanItems];
    int32_t run = 0;
#pragma unroll
    for (int e = 0; e < kScanItems; e++) { const int64_t i = base + e; v[e] = (i < n) ? data[i] : 0; run += v[e]; v[e] = run; }
    int32_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const int32_t excl = woff + x - run;
#pragma unroll
    for (int e = 0; e < kScanItems; e++) { const int64_t i = base + e; if (i < n) data[i] = v[e] + excl; }
    if (tid == kScanBlock - 1) sums[blockIdx.x] = woff + x;
}

// Genome-wide form, ONE launch per contig: the tile scan as above, and the workgroup that finishes last turns the tile totals
// into exclusive offsets in place.  The depths are left tile-local; depth_query_tiled adds a position's tile offset when it
// reads it -- no third pass over the contig.  Totals travel through returning agent-scope atomics (written and read back
// on the same path, no fence: im_triage.hip uses the same hand-over); sums[tiles] is the arrival counter, zero between launches.
__global__ __launch_bounds__(kScanBlock) void depth_scan_tiled_kernel(int32_t* __restrict__ data, int64_t n, int32_t* __restrict__ sums, int32_t tiles)
{
    __shared__ int32_t wsum[kScanBlock / 64];
    __shared__ int32_t s_last, carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
    int32_t v[kScanItems];
    int32_t run = 0;
    static_assert(kScanItems == 8, "a thread's eight items travel as two 16-byte accesses");
    const bool whole = base + kScanItems <= n;          // the contig's run starts on a 256-byte boundary: base is 32-byte aligned
    if (whole) {
        const int4 a = *reinterpret_cast<const int4*>(data + base), c = *reinterpret_cast<const int4*>(data + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
    } else {
#pragma unroll
        for (int e = 0; e < kScanItems; e++) { const int64_t i = base + e; v[e] = (i < n) ? data[i] : 0; }
    }
#pragma unroll
    for (int e = 0; e < kScanItems; e++) { run += v[e]; v[e] = run; }
    int32_t x = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const int32_t excl = woff + x - run;
    if (whole) {
        *reinterpret_cast<int4*>(data + base) = make_int4(v[0] + excl, v[1] + excl, v[2] + excl, v[3] + excl);
        *reinterpret_cast<int4*>(data + base + 4) = make_int4(v[4] + excl, v[5] + excl, v[6] + excl, v[7] + excl);
    } else {
#pragma unroll
        for (int e = 0; e < kScanItems; e++) { const int64_t i = base + e; if (i < n) data[i] = v[e] + excl; }
    }
    if (tid == kScanBlock - 1) {
        const int32_t seen = atomicExch(&sums[blockIdx.x], woff + x);
        asm volatile("" :: "v"(seen));                  // the total is in before this workgroup is counted
        // counted in two levels -- a group of kScanGroup tiles, then the groups: same-address atomics are served one after the
        // other (~9 ns each: 7 us of the 6.25 Mb contig's launch when every tile counted into one word, im_triage.hip has the measurement)
        const int32_t g = (int32_t)blockIdx.x / kScanGroup, groups = (tiles + kScanGroup - 1) / kScanGroup;
        const int32_t g_n = min(kScanGroup, tiles - g * kScanGroup);
        int32_t last = 0;
        if (atomicAdd(&sums[tiles + 1 + g], 1) == g_n - 1) {
            atomicExch(&sums[tiles + 1 + g], 0);        // ready for the next launch
            last = atomicAdd(&sums[tiles], 1) == groups - 1 ? 1 : 0;
        }
        s_last = last;
        carry_s = 0;
    }
    __syncthreads();
    if (!s_last) return;
    for (int32_t b0 = 0; b0 < tiles; b0 += kScanBlock) {
        const int32_t j = b0 + tid;
        const int32_t t = j < tiles ? atomicAdd(&sums[j], 0) : 0;
        int32_t y = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int32_t u = __shfl_up(y, o); if (lane >= o) y += u; }
        if (lane == 63) wsum[wave] = y;
        __syncthreads();
        int32_t wo = 0;
        for (int w = 0; w < wave; w++) wo += wsum[w];
        const int32_t carry = carry_s;
        if (j < tiles) sums[j] = carry + wo + y - t;
        __syncthreads();
        if (tid == kScanBlock - 1) carry_s = carry + wo + y;
        __syncthreads();
    }
    if (tid == 0) sums[tiles] = 0;
}

__global__ __launch_bounds__(256) void depth_query_tiled_kernel(int32_t nq, const int32_t* __restrict__ beg, const int32_t* __restrict__ end,
                                                               const int32_t* __restrict__ depth, const int32_t* __restrict__ sums,
                                                               int64_t clen, uint32_t* __restrict__ out, uint32_t* __restrict__ out_max)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int q = wave; q < nq; q += nwaves) {
        int64_t a = beg[q], b = end[q];
        if (a < 0) a = 0;
        if (b > clen) b = clen;
        // out_max: the deepest position of [beg - 1, end] (the host asks the file about loci deep enough for samtools' pileup cap)
        const int64_t a1 = out_max && a > 0 ? a - 1 : a, b1 = out_max && b < clen ? b + 1 : b;
        uint32_t s = 0, mx = 0;
        for (int64_t p = a1 + lane; p < b1; p += 64) {
            const uint32_t v = (uint32_t)(depth[p] + sums[p / kScanTile]);
            if (p >= a && p < b) s += v;
            mx = max(mx, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += (uint32_t)__shfl_xor((int)s, o); mx = max(mx, (uint32_t)__shfl_xor((int)mx, o)); }
        if (lane == 0) { out[q] = s; if (out_max) out_max[q] = mx; }
    }
}

// phase 2: exclusive scan of the tile totals by one workgroup
__global__ __launch_bounds__(kScanBlock) void depth_scan_sums_kernel(int32_t* __restrict__ sums, int64_t n)
{
    __shared__ int32_t wsum[kScanBlock / 64];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += kScanBlock) {
        const int64_t i = base + tid;
        const int32_t v = (i < n) ? sums[i] : 0;
        int32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int32_t carry = carry_s;
        if (i < n) sums[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == kScanBlock - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
}

// phase 3: add the scanned tile offsets
__global__ __launch_bounds__(kScanBlock) void depth_scan_add_kernel(int32_t* __restrict__ data, int64_t n, const int32_t* __restrict__ sums)
{
    const int32_t off = sums[blockIdx.x];
    if (off == 0) return;
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int e = 0; e < kScanItems; e++) { const int64_t i = base + e; if (i < n) data[i] += off; }
}

__global__ __launch_bounds__(256) void depth_query_kernel(int32_t nq, const int32_t* __restrict__ beg, const int32_t* __restrict__ end,
                                                         const int32_t* __restrict__ depth, int64_t clen, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int q = wave; q < nq; q += nwaves) {
        int64_t a = beg[q], b = end[q];
        if (a < 0) a = 0;
        if (b > clen) b = clen;
        uint32_t s = 0;
        for (int64_t p = a + lane; p < b; p += 64) s += (uint32_t)depth[p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += (uint32_t)__shfl_xor((int)s, o);
        if (lane == 0) out[q] = s;
    }
}

}  // namespace

hipError_t launch_depth_build(int64_t clen, int32_t n_seg, const int32_t* seg_start, const int32_t* seg_len,
                              int32_t* depth /* clen + 1 */, int32_t* sums /* tiles */, hipStream_t stream)
{
    const int64_t n = clen + 1;
    hipError_t e = hipMemsetAsync(depth, 0, (size_t)n * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    if (n_seg > 0) {
        int64_t b = ((int64_t)n_seg + 255) / 256;
        if (b > 4096) b = 4096;
        hipLaunchKernelGGL(depth_scatter_kernel, dim3((int)b), dim3(256), 0, stream, n_seg, seg_start, seg_len, clen, depth);
    }
    return launch_depth_scan(depth, n, sums, stream);
}

// difference array -> depths, in place: n elements, sums holds depth_tiles(n - 1) entries
hipError_t launch_depth_scan(int32_t* depth, int64_t n, int32_t* sums, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(depth_scan_tiles_kernel, dim3((int)tiles), dim3(kScanBlock), 0, stream, depth, n, sums);
    hipLaunchKernelGGL(depth_scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, stream, sums, tiles);
    hipLaunchKernelGGL(depth_scan_add_kernel, dim3((int)tiles), dim3(kScanBlock), 0, stream, depth, n, sums);
    return hipGetLastError();
}

int64_t depth_tiles(int64_t clen) { return (clen + 1 + kScanTile - 1) / kScanTile; }
// ints of a contig's run of tile sums in the genome-wide form: the tiles' totals, the arrival counter of the groups, one arrival counter per group
int64_t depth_sums_ints(int64_t clen) { const int64_t t = depth_tiles(clen); return t + 1 + (t + kScanGroup - 1) / kScanGroup; }

// genome-wide form: n elements, sums holds depth_sums_ints(n - 1) entries (behind the tiles' totals the arrival counters, zero between launches)
hipError_t launch_depth_scan_tiled(int32_t* depth, int64_t n, int32_t* sums, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int64_t tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL(depth_scan_tiled_kernel, dim3((int)tiles), dim3(kScanBlock), 0, stream, depth, n, sums, (int32_t)tiles);
    return hipGetLastError();
}

hipError_t launch_depth_query_tiled(int32_t nq, const int32_t* beg, const int32_t* end, const int32_t* depth, const int32_t* sums,
                                    int64_t clen, uint32_t* out, uint32_t* out_max, hipStream_t stream)
{
    if (nq <= 0) return hipSuccess;
    int b = (nq + 3) / 4;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(depth_query_tiled_kernel, dim3(b), dim3(256), 0, stream, nq, beg, end, depth, sums, clen, out, out_max);
    return hipGetLastError();
}

hipError_t launch_depth_query(int32_t nq, const int32_t* beg, const int32_t* end, const int32_t* depth, int64_t clen,
                              uint32_t* out, hipStream_t stream)
{
    if (nq <= 0) return hipSuccess;
    int b = (nq + 3) / 4;
    if (b > 2048) b = 2048;
    hipLaunchKernelGGL(depth_query_kernel, dim3(b), dim3(256), 0, stream, nq, beg, end, depth, clen, out);
    return hipGetLastError();
}

}  // namespace im
