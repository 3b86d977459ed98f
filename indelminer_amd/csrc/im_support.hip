// im_support.hip -- annotate mode: does a read support a known indel?  (K7)
//
// Replaces the alignment inside realign_with_indel (src/variant.c:1246-1424), which
// check_for_indel (1427-1556) runs for every read overlapping a known split-read variant that
// the discovery pass did not re-find: a full affine-gap local Smith-Waterman of the read against
// the reference window WITH the variant applied (match +2, mismatch -1, gap open 4, extend 1),
// a traceback from the first strictly-greatest cell while the score stays positive, and three
// counts over the traced path (substitutions, inserted + deleted bases, "aligned" bases).
//
// The reference stores the whole (len2+1) x (len1+1) score and direction matrices per call (and
// leaks them).  The counts are a function of the traced path only, and the path from any cell is
// fixed by the direction choices, so they can be carried FORWARD with the scores:
//     stats(i,j) = step(dir(i,j)) + (V(pred) > 0 ? stats(pred) : 0)
// and the answer is stats at the best cell.  No matrices, no traceback.
//
// One wavefront per (read, variant) task.  Rows (read bases) map to lanes, 64 rows per block;
// the block sweeps the columns skewed by the lane index so that every lane has its upper and
// diagonal neighbours from the lane below one and two steps earlier (DPP wave_shr:1), its left
// neighbour from itself.  The last row of a block is parked in LDS for the next block.
// Tie rules kept: substitution unless strictly smaller than the best gap; insertion over deletion
// on equality (1337-1342); first strictly-greater maximum in row-major order (1346-1350).

//
// Two instances.  <false>: windows of up to IM_MAX_SW_TARGET bytes and queries of up to IM_MAX_READ bases -- target and
// boundary row in LDS, the three statistics packed into 32 bits: every read a sequencer pairs and every indel up to ~1500
// bases.  <true>: anything longer (the reference takes any -p, src/variant.c:841-921,1246-1312): the target is read where
// it lies, the boundary row is parked in device memory (one stretch per workgroup), the statistics are packed into 64 bits.

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kSwMaxTarget = IM_MAX_SW_TARGET;
constexpr int kDppWaveShr1S = 0x138;

// LDS of one workgroup, sized by the launch for the longest target of the batch (cap = that length + 1,
// rounded up to 4): the target bytes and the parked boundary row (V, F, stats).  A 150-base window costs
// 2 KiB, so the CU fills with waves; sizing for kSwMaxTarget would leave three.
template <typename ST>
struct SwRow {
    const uint8_t* t1;
    int32_t*  rowV;
    int32_t*  rowF;
    ST*       rowS;
};
__device__ __forceinline__ SwRow<uint32_t> sw_carve(unsigned char* base, int cap)
{
    SwRow<uint32_t> s;
    s.t1 = base;
    s.rowV = reinterpret_cast<int32_t*>(base + cap);
    s.rowF = s.rowV + cap;
    s.rowS = reinterpret_cast<uint32_t*>(s.rowF + cap);
    return s;
}
inline size_t sw_lds_bytes(int cap) { return (size_t)cap * 13; }
__host__ __device__ inline size_t sw_big_row_bytes(int64_t cap) { return (size_t)cap * 16; }      // V, F, 64-bit statistics per target base

__device__ __forceinline__ int sw_shr1(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, kDppWaveShr1S, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t up8(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }

// packed path statistics: aligned [0,10) | indels [10,21) | substitutions [21,32).  Statistics only run along cells of
// positive score: with m matches (+2 each, m <= IM_MAX_READ = 1020) the substitutions and gap bases of such a path stay
// below 2m <= 2040 < 2^11, and "aligned" counts read bases (<= 1020 < 2^10).
// The long form packs aligned [0,21) | indels [21,42) | substitutions [42,64): queries of up to 2^20 bases.
template <bool BIG> struct SwStat;
template <> struct SwStat<false> {
    typedef uint32_t T;
    static constexpr T kAligned = 1u, kIndel = 1u << 10, kSub = 1u << 21;
    static __device__ __forceinline__ int subs(T s) { return (int)(s >> 21); }
    static __device__ __forceinline__ int indels(T s) { return (int)((s >> 10) & 0x7FFu); }
    static __device__ __forceinline__ int aligned(T s) { return (int)(s & 0x3FFu); }
};
template <> struct SwStat<true> {
    typedef uint64_t T;
    static constexpr T kAligned = 1ull, kIndel = 1ull << 21, kSub = 1ull << 42;
    static __device__ __forceinline__ int subs(T s) { return (int)(s >> 42); }
    static __device__ __forceinline__ int indels(T s) { return (int)((s >> 21) & 0x1FFFFFull); }
    static __device__ __forceinline__ int aligned(T s) { return (int)(s & 0x1FFFFFull); }
};
static_assert(IM_MAX_READ <= 1023, "the packed statistics of the short form are sized for reads of up to 1023 bases");
constexpr int kSwBigMaxQuery = 1 << 20;

__device__ __forceinline__ uint32_t stat_shr1(uint32_t v) { return (uint32_t)sw_shr1(0, (int)v); }
__device__ __forceinline__ uint64_t stat_shr1(uint64_t v)
{
    const uint32_t lo = (uint32_t)sw_shr1(0, (int)(uint32_t)v), hi = (uint32_t)sw_shr1(0, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t stat_xor(uint32_t v, int o) { return (uint32_t)__shfl_xor((int)v, o); }
__device__ __forceinline__ uint64_t stat_xor(uint64_t v, int o)
{
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
    return ((uint64_t)hi << 32) | lo;
}

template <bool BIG>
__global__ __launch_bounds__(64) void support_kernel(int32_t n_tasks,
                                                    const uint8_t* __restrict__ targets, const int64_t* __restrict__ t_off,
                                                    const uint8_t* __restrict__ queries, const int64_t* __restrict__ q_off,
                                                    int32_t* __restrict__ out /* n x 4: subs, indels, aligned, status */,
                                                    int cap /* row capacity in target bases, multiple of 4 */,
                                                    unsigned char* __restrict__ big_rows /* BIG: gridDim.x stretches of sw_big_row_bytes(cap) */)
{
    typedef SwStat<BIG> St;
    typedef typename St::T stat_t;
    constexpr stat_t kStAligned = St::kAligned, kStIndel = St::kIndel, kStSub = St::kSub;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    SwRow<stat_t> s;
    if constexpr (BIG) {
        unsigned char* mine = big_rows + (size_t)blockIdx.x * sw_big_row_bytes(cap);
        s.rowS = reinterpret_cast<stat_t*>(mine);
        s.rowV = reinterpret_cast<int32_t*>(mine + (size_t)cap * 8);
        s.rowF = s.rowV + cap;
        s.t1 = targets;
    } else {
        s = sw_carve(smem_raw, cap);
    }
    const int lane = threadIdx.x;
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int64_t to = t_off[task], qo = q_off[task];
        const int64_t l1 = t_off[task + 1] - to, l2 = q_off[task + 1] - qo;
        const bool fits_short = l1 >= 0 && l2 >= 0 && l1 <= kSwMaxTarget && l2 <= IM_MAX_READ;
        if constexpr (BIG) {
            if (fits_short) continue;                                   // the short form has done it
            if (l1 < 0 || l2 < 0 || l1 >= cap || l2 > kSwBigMaxQuery) continue;      // stays IM_ST_UNSUPPORTED
        } else {
            if (!fits_short || l1 >= cap) {
                if (lane == 0) { out[4 * task] = 0; out[4 * task + 1] = 0; out[4 * task + 2] = 0; out[4 * task + 3] = IM_ST_UNSUPPORTED; }
                continue;
            }
        }
        const int len1 = (int)l1, len2 = (int)l2;
        if constexpr (BIG) s.t1 = targets + to;
        else {
            uint8_t* t1w = smem_raw;
            for (int j = lane; j < len1; j += 64) t1w[j] = targets[to + j];
        }
        __syncthreads();

        int g_best = 0; stat_t g_stats = 0;              // max_score starts at 0: an all-nonpositive matrix traces nothing
        for (int i0 = 0; i0 < len2; i0 += 64) {
            const int i = i0 + 1 + lane;                    // this lane's row (1-based)
            const bool row_ok = i <= len2;
            const uint32_t qc_raw = row_ok ? queries[qo + i - 1] : 0u;
            const uint32_t qc = up8(qc_raw);
            // own previous cell (i, j-1): column 0 to start with (1297-1303)
            int v_left = -4 - i, e_left = 0; stat_t s_left = 0;
            // what the lane below produced one and two steps ago
            int v_out = 0, f_out = 0; stat_t s_out = 0;     // this lane's newest cell
            int v_diag_in = -4 - (i - 1);                   // (i-1, 0)
            stat_t s_diag_in = 0;
            int best_v = 0; stat_t best_s = 0;
            const int steps = len1 + 63;
            for (int t = 0; t < steps; t++) {
                // neighbours from the lane below: its newest cell is (i-1, j)
                int v_up = sw_shr1(0, v_out), f_up = sw_shr1(0, f_out);
                stat_t s_up = stat_shr1(s_out);
                const int j = t - lane + 1;
                if (lane == 0) {
                    if (j >= 1 && j <= len1) {
                        if (i0 == 0) { v_up = -4 - j; f_up = 0; s_up = 0; }          // row 0 (1301-1303), F zero-filled (1307)
                        else { v_up = s.rowV[j]; f_up = s.rowF[j]; s_up = s.rowS[j]; }
                    }
                }
                const bool act = row_ok && j >= 1 && j <= len1;
                if (act) {
                    const uint32_t tc_raw = s.t1[j - 1];
                    const int sub = v_diag_in + ((up8(tc_raw) == qc) ? 2 : -1);
                    const int ins = max(f_up, v_up - 4) - 1;
                    const int del = max(e_left, v_left - 4) - 1;
                    const int indel = max(ins, del);
                    int v = sub;
                    stat_t st = (v_diag_in > 0 ? s_diag_in : (stat_t)0) + kStAligned + ((tc_raw != qc_raw) ? kStSub : (stat_t)0);
                    if (v < indel) {
                        v = indel;
                        if (ins >= del) st = (v_up > 0 ? s_up : (stat_t)0) + kStIndel + kStAligned;
                        else            st = (v_left > 0 ? s_left : (stat_t)0) + kStIndel;
                    }
                    if (v > best_v) { best_v = v; best_s = st; }
                    // becomes "left" for the next column and "up" for the lane above
                    v_left = v; e_left = del; s_left = st;
                    v_out = v; f_out = ins; s_out = st;
                    if (lane == 63) { s.rowV[j] = v; s.rowF[j] = ins; s.rowS[j] = st; }
                }
                // the cell above-left of the next column is the cell above of this one
                v_diag_in = (j >= 1 && j <= len1) ? v_up : v_diag_in;
                s_diag_in = (j >= 1 && j <= len1) ? s_up : s_diag_in;
                if (j == 0) { v_diag_in = -4 - (i - 1); s_diag_in = 0; }
                if (lane == 0 && j >= 1 && j <= len1 && i0 > 0) { /* boundary row diag comes from LDS too */ }
            }
            // block maximum: larger score, then smaller row (rows ascend with the lane)
            int bv = best_v; stat_t bs = best_s; int bl = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ov = __shfl_xor(bv, o); const stat_t os = stat_xor(bs, o); const int ol = __shfl_xor(bl, o);
                if (ov > bv || (ov == bv && ol < bl)) { bv = ov; bs = os; bl = ol; }
            }
            if (bv > g_best) { g_best = bv; g_stats = bs; }
            __syncthreads();
        }
        if (lane == 0) {
            out[4 * task]     = St::subs(g_stats);
            out[4 * task + 1] = St::indels(g_stats);
            out[4 * task + 2] = St::aligned(g_stats) + 1;          // the NUL position is counted too (1392-1404)
            out[4 * task + 3] = IM_ST_EVIDENCE;
        }
        __syncthreads();
    }
}

}  // namespace

size_t support_big_scratch_bytes(int64_t max_target, int64_t max_query, int32_t n_tasks, int32_t* grid_out)
{
    *grid_out = 0;
    if (max_target <= kSwMaxTarget && max_query <= IM_MAX_READ) return 0;
    const int64_t cap = (max_target + 1 + 3) & ~(int64_t)3;
    const size_t per = sw_big_row_bytes(cap);
    int64_t grid = n_tasks < 2048 ? n_tasks : 2048;
    const size_t budget = (size_t)2 << 30;
    if (per * (size_t)grid > budget) grid = (int64_t)(budget / per);
    if (grid < 1) grid = 1;
    *grid_out = (int32_t)grid;
    return per * (size_t)grid;
}

hipError_t launch_support(int32_t n_tasks, const uint8_t* targets, const int64_t* t_off,
                          const uint8_t* queries, const int64_t* q_off, int32_t* out, int64_t max_target, int64_t max_query,
                          void* big_scratch, int32_t big_grid, int n_cu, hipStream_t stream)
{
    if (n_tasks <= 0) return hipSuccess;
    if (max_target < 0) max_target = 0;
    const int64_t whole_target = max_target;
    if (max_target > kSwMaxTarget) max_target = kSwMaxTarget;      // longer targets take the second launch
    const int cap = (int)((max_target + 1 + 3) & ~(int64_t)3);
    const size_t lds = sw_lds_bytes(cap);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(support_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sw_lds_bytes((kSwMaxTarget + 4) & ~3));
        if (e != hipSuccess) return e;
        attr_bytes = sw_lds_bytes((kSwMaxTarget + 4) & ~3);
    }
    // one task per workgroup while that stays a sane grid: the dispatcher balances tasks of different cost
    (void)n_cu;
    const int grid = n_tasks < (1 << 20) ? n_tasks : (1 << 20);
    hipLaunchKernelGGL(support_kernel<false>, dim3(grid), dim3(64), lds, stream, n_tasks, targets, t_off, queries, q_off, out, cap, (unsigned char*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || big_grid <= 0) return e;
    if (whole_target >= 0x7ffffff0LL) return hipErrorInvalidValue;
    const int big_cap = (int)((whole_target + 1 + 3) & ~(int64_t)3);
    (void)max_query;
    hipLaunchKernelGGL(support_kernel<true>, dim3(big_grid), dim3(64), 0, stream, n_tasks, targets, t_off, queries, q_off, out, big_cap,
                       static_cast<unsigned char*>(big_scratch));
    return hipGetLastError();
}

}  // namespace im
