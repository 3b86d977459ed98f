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

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kSwMaxTarget = IM_MAX_SW_TARGET;
constexpr int kDppWaveShr1S = 0x138;

// LDS of one workgroup, sized by the launch for the longest target of the batch (cap = that length + 1,
// rounded up to 4): the target bytes and the parked boundary row (V, F, stats).  A 150-base window costs
// 2 KiB, so the CU fills with waves; sizing for kSwMaxTarget would leave three.
struct SwLds {
    uint8_t*  t1;
    int32_t*  rowV;
    int32_t*  rowF;
    uint32_t* rowS;
};
__device__ __forceinline__ SwLds sw_carve(unsigned char* base, int cap)
{
    SwLds s;
    s.t1 = base;
    s.rowV = reinterpret_cast<int32_t*>(base + cap);
    s.rowF = s.rowV + cap;
    s.rowS = reinterpret_cast<uint32_t*>(s.rowF + cap);
    return s;
}
inline size_t sw_lds_bytes(int cap) { return (size_t)cap * 13; }

__device__ __forceinline__ int sw_shr1(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, kDppWaveShr1S, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t up8(uint32_t c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }

// packed path statistics: aligned [0,10) | indels [10,21) | substitutions [21,32).  Statistics only run along cells of
// positive score: with m matches (+2 each, m <= IM_MAX_READ = 1020) the substitutions and gap bases of such a path stay
// below 2m <= 2040 < 2^11, and "aligned" counts read bases (<= 1020 < 2^10).
constexpr uint32_t kStAligned = 1u, kStIndel = 1u << 10, kStSub = 1u << 21;
static_assert(IM_MAX_READ <= 1023, "the packed statistics are sized for reads of up to 1023 bases");

__global__ __launch_bounds__(64) void support_kernel(int32_t n_tasks,
                                                    const uint8_t* __restrict__ targets, const int64_t* __restrict__ t_off,
                                                    const uint8_t* __restrict__ queries, const int64_t* __restrict__ q_off,
                                                    int32_t* __restrict__ out /* n x 4: subs, indels, aligned, status */,
                                                    int cap /* LDS capacity in target bases, multiple of 4 */)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const SwLds s = sw_carve(smem_raw, cap);
    const int lane = threadIdx.x;
    for (int task = blockIdx.x; task < n_tasks; task += gridDim.x) {
        const int64_t to = t_off[task], qo = q_off[task];
        const int len1 = (int)(t_off[task + 1] - to), len2 = (int)(q_off[task + 1] - qo);
        if (len1 > kSwMaxTarget || len1 >= cap || len2 > IM_MAX_READ || len1 < 0 || len2 < 0) {
            if (lane == 0) { out[4 * task] = 0; out[4 * task + 1] = 0; out[4 * task + 2] = 0; out[4 * task + 3] = IM_ST_UNSUPPORTED; }
            continue;
        }
        for (int j = lane; j < len1; j += 64) s.t1[j] = targets[to + j];
        __syncthreads();

        int g_best = 0; uint32_t g_stats = 0;            // max_score starts at 0: an all-nonpositive matrix traces nothing
        for (int i0 = 0; i0 < len2; i0 += 64) {
            const int i = i0 + 1 + lane;                    // this lane's row (1-based)
            const bool row_ok = i <= len2;
            const uint32_t qc_raw = row_ok ? queries[qo + i - 1] : 0u;
            const uint32_t qc = up8(qc_raw);
            // own previous cell (i, j-1): column 0 to start with (1297-1303)
            int v_left = -4 - i, e_left = 0; uint32_t s_left = 0;
            // what the lane below produced one and two steps ago
            int v_out = 0, f_out = 0; uint32_t s_out = 0;   // this lane's newest cell
            int v_diag_in = -4 - (i - 1);                   // (i-1, 0)
            uint32_t s_diag_in = 0;
            int best_v = 0; uint32_t best_s = 0;
            const int steps = len1 + 63;
            for (int t = 0; t < steps; t++) {
                // neighbours from the lane below: its newest cell is (i-1, j)
                int v_up = sw_shr1(0, v_out), f_up = sw_shr1(0, f_out);
                uint32_t s_up = (uint32_t)sw_shr1(0, (int)s_out);
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
                    uint32_t st = (v_diag_in > 0 ? s_diag_in : 0u) + kStAligned + ((tc_raw != qc_raw) ? kStSub : 0u);
                    if (v < indel) {
                        v = indel;
                        if (ins >= del) st = (v_up > 0 ? s_up : 0u) + kStIndel + kStAligned;
                        else            st = (v_left > 0 ? s_left : 0u) + kStIndel;
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
            int bv = best_v; uint32_t bs = best_s; int bl = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const int ov = __shfl_xor(bv, o); const uint32_t os = (uint32_t)__shfl_xor((int)bs, o); const int ol = __shfl_xor(bl, o);
                if (ov > bv || (ov == bv && ol < bl)) { bv = ov; bs = os; bl = ol; }
            }
            if (bv > g_best) { g_best = bv; g_stats = bs; }
            __syncthreads();
        }
        if (lane == 0) {
            out[4 * task]     = (int32_t)(g_stats >> 21);
            out[4 * task + 1] = (int32_t)((g_stats >> 10) & 0x7FFu);
            out[4 * task + 2] = (int32_t)(g_stats & 0x3FFu) + 1;   // the NUL position is counted too (1392-1404)
            out[4 * task + 3] = IM_ST_EVIDENCE;
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_support(int32_t n_tasks, const uint8_t* targets, const int64_t* t_off,
                          const uint8_t* queries, const int64_t* q_off, int32_t* out, int32_t max_target, int n_cu, hipStream_t stream)
{
    if (n_tasks <= 0) return hipSuccess;
    if (max_target < 0) max_target = 0;
    if (max_target > kSwMaxTarget) max_target = kSwMaxTarget;      // longer targets are reported per task
    const int cap = (max_target + 1 + 3) & ~3;
    const size_t lds = sw_lds_bytes(cap);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(support_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sw_lds_bytes((kSwMaxTarget + 4) & ~3));
        if (e != hipSuccess) return e;
        attr_bytes = sw_lds_bytes((kSwMaxTarget + 4) & ~3);
    }
    // one task per workgroup while that stays a sane grid: the dispatcher balances tasks of different cost
    (void)n_cu;
    const int grid = n_tasks < (1 << 20) ? n_tasks : (1 << 20);
    hipLaunchKernelGGL(support_kernel, dim3(grid), dim3(64), lds, stream, n_tasks, targets, t_off, queries, q_off, out, cap);
    return hipGetLastError();
}

}  // namespace im
