// im_cluster.hip -- split-read evidence clustering on gfx950.
//
// Replaces the split-read part of process_evidence (src/indelminer.c:117-209):
//   slsort(allevidence, sort_evidence)      src/indelminer.c:123, src/evidence.c:50-58
//   nodes for the sorted prefix b2 < marker src/indelminer.c:137-146
//   SR edges: same class, b1, b2            src/graph.c:122-127
//   components -> variants                  src/indelminer.c:159-204
// The reference builds the graph with an O(N^2) list scan; the SR rule only
// joins identical keys, so the same partition is a stable sort by (b1,b2)
// followed by a run-length pass.  Evidence arrives in ARRIVAL order; the
// reference's prepend list + glibc's stable merge sort leave the members of a
// cluster in ascending arrival order, which is exactly what a
// stable LSD radix sort of the arrival-ordered input gives.
//
// For identical (b1,b2) every record has the same class by construction
// (insertions have b1 == b2, deletions b2 > b1), so class is checked between
// sorted neighbours instead of being a sort key.
//
// Kernels: key build, 8 x (tile histogram, offset scan, stable scatter),
// marker cut (min-reduction), run heads, head scan, cluster table, order.
// All memory-streaming; the radix passes move 12 B per record per pass.

#include "im_device.hpp"

#include <mutex>

namespace im {
namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kSortThreads = 256;
constexpr int kSortItems = 8;
constexpr int kSortTile = kSortThreads * kSortItems;     // 2048 records per workgroup
constexpr int kScanThreads = 1024;

__device__ __forceinline__ uint64_t make_key(int32_t b1, int32_t b2)
{
    // order-preserving for signed values
    return ((uint64_t)((uint32_t)b1 ^ 0x80000000u) << 32) | (uint64_t)((uint32_t)b2 ^ 0x80000000u);
}
__device__ __forceinline__ int32_t key_b2(uint64_t k) { return (int32_t)((uint32_t)k ^ 0x80000000u); }

__global__ __launch_bounds__(256) void build_keys_kernel(const int32_t* __restrict__ n_dev, int32_t n_cap, const int32_t* __restrict__ b1, const int32_t* __restrict__ b2,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int32_t n = min(*n_dev, n_cap);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = make_key(b1[i], b2[i]);
        vals[i] = (uint32_t)i;
    }
}

// per-tile digit histogram, stored digit-major: hist[d * nblocks + tile]
__global__ __launch_bounds__(kSortThreads) void radix_hist_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ n_dev, int32_t n_cap, int shift,
                                                                 uint32_t* __restrict__ hist, int nblocks)
{
    const int32_t n = min(*n_dev, n_cap);
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
#pragma unroll
    for (int r = 0; r < kSortItems; r++) {
        const int64_t e = base + r * kSortThreads + threadIdx.x;
        if (e < n) atomicAdd(&h[(uint32_t)(keys[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of a u32 array by ONE workgroup (arrays here are small: 256 x tiles,
// or one flag per evidence record).  total (optional) receives the sum.
__global__ __launch_bounds__(kScanThreads) void scan_excl_kernel(const uint32_t* in, uint32_t* out,
                                                                int64_t n_host, const int32_t* __restrict__ n_dev,
                                                                uint32_t* __restrict__ total)
{
    // n_dev (if given) bounds the scan from device memory; n_host is then the capacity
    const int64_t n = n_dev ? min((int64_t)*n_dev, n_host) : n_host;
    __shared__ uint32_t wsum[kScanThreads / 64];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += kScanThreads) {
        const int64_t i = base + tid;
        const uint32_t v = (i < n) ? in[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t carry = carry_s;
        if (i < n) out[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == kScanThreads - 1) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0 && total) *total = carry_s;
}

// stable scatter of one radix pass
__global__ __launch_bounds__(kSortThreads) void radix_scatter_kernel(const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
                                                                    uint64_t* __restrict__ kout, uint32_t* __restrict__ vout,
                                                                    const int32_t* __restrict__ n_dev, int32_t n_cap, int shift,
                                                                    const uint32_t* __restrict__ offs, int nblocks)
{
    const int32_t n = min(*n_dev, n_cap);
    __shared__ uint32_t running[256];
    __shared__ uint32_t wcnt[kSortThreads / 64][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    running[tid] = offs[(int64_t)tid * nblocks + blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kSortTile;
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < kSortItems; r++) {
        const int64_t e = base + r * kSortThreads + tid;
        const bool valid = e < n;
        uint64_t key = 0; uint32_t val = 0;
        if (valid) { key = kin[e]; val = vin[e]; }
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
#pragma unroll
        for (int w = 0; w < kSortThreads / 64; w++) wcnt[w][tid] = 0;
        // lanes of this wave holding the same digit
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = __ballot(valid && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t cnt = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid && rank == 0) wcnt[wave][d] = cnt;
        __syncthreads();
        {
            // digit `tid`: turn the per-wave counts into per-wave output offsets
            uint32_t run = running[tid];
#pragma unroll
            for (int w = 0; w < kSortThreads / 64; w++) { const uint32_t c = wcnt[w][tid]; wcnt[w][tid] = run; run += c; }
            running[tid] = run;
        }
        __syncthreads();
        if (valid) {
            const uint32_t dst = wcnt[wave][d] + rank;
            kout[dst] = key; vout[dst] = val;
        }
        __syncthreads();
    }
}

// first sorted position whose b2 >= marker (the reference stops making nodes there,
// src/indelminer.c:140-142); *cut starts at n
__global__ __launch_bounds__(256) void cut_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ n_dev, int32_t n_cap, int32_t marker, uint32_t* __restrict__ cut)
{
    const int32_t n = min(*n_dev, n_cap);
    uint32_t best = 0xFFFFFFFFu;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (key_b2(keys[i]) >= marker) { best = (uint32_t)i; break; }     // positions ascend per thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
    if ((threadIdx.x & 63) == 0 && best != 0xFFFFFFFFu) atomicMin(cut, best);
}

__global__ __launch_bounds__(256) void heads_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                   const int32_t* __restrict__ cls, const int32_t* __restrict__ n_dev, int32_t n_cap,
                                                   const uint32_t* __restrict__ cut,
                                                   uint32_t* __restrict__ head, uint8_t* __restrict__ used)
{
    const int32_t n = min(*n_dev, n_cap);
    const uint32_t m = min(*cut, (uint32_t)n);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = 0;
        if (i < m) h = (i == 0) || keys[i] != keys[i - 1] || cls[vals[i]] != cls[vals[i - 1]];
        head[i] = h;
        used[vals[i]] = (i < m) ? 1 : 0;
    }
}

// cl_first[c] for every head; cl_first[n_clusters] = cut as a sentinel (array has n+1 slots in scratch)
__global__ __launch_bounds__(256) void cluster_first_kernel(const uint32_t* __restrict__ head, const uint32_t* __restrict__ cid,
                                                           const int32_t* __restrict__ n_dev, int32_t n_cap,
                                                           const uint32_t* __restrict__ cut, const uint32_t* __restrict__ ncl,
                                                           int32_t* __restrict__ first_tmp, int32_t* __restrict__ n_clusters)
{
    const int32_t n = min(*n_dev, n_cap);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (head[i]) first_tmp[cid[i]] = (int32_t)i;
    if (blockIdx.x == 0 && threadIdx.x == 0) { first_tmp[*ncl] = (int32_t)min(*cut, (uint32_t)n); *n_clusters = (int32_t)*ncl; }
}

__global__ __launch_bounds__(256) void cluster_finish_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ head,
                                                            const uint32_t* __restrict__ cid, const int32_t* __restrict__ n_dev, int32_t n_cap,
                                                            const uint32_t* __restrict__ cut, const int32_t* __restrict__ first_tmp,
                                                            int32_t tie_desc,
                                                            int32_t* __restrict__ order, int32_t* __restrict__ cl_first, int32_t* __restrict__ cl_count)
{
    const int32_t n = min(*n_dev, n_cap);
    const uint32_t m = min(*cut, (uint32_t)n);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= m) { order[i] = (int32_t)vals[i]; continue; }
        // cid[] is the exclusive scan of head[], so a head's own id is cid, a member's is cid - 1
        const uint32_t c = head[i] ? cid[i] : cid[i] - 1;
        const int32_t f = first_tmp[c], cnt = first_tmp[c + 1] - f;
        if (head[i]) { cl_first[c] = f; cl_count[c] = cnt; }
        const int32_t pos = tie_desc ? (f + (cnt - 1 - ((int32_t)i - f))) : (int32_t)i;
        order[pos] = (int32_t)vals[i];
    }
}

// ---- single-workgroup path ----------------------------------------------------
//
// One launch for the whole of process_evidence's split-read work when at most
// kSmallMax evidence records are live (a READCHUNK flush of the reference holds a
// few thousand): compaction of the evidence slots in arrival order, bitonic sort
// of (b1, b2, slot) in LDS, marker cut, run heads, head scan, cluster table.
// Input is a SLOT array: slot i is live iff cls[i] >= 0 (the realign kernel writes
// IM_MAX_EV slots per read, empty ones with cls = -1); slot order = arrival order.

constexpr int kSmallThreads = 1024;
constexpr int kSmallMax = 8192;

struct SmallLds {
    int32_t  b1[kSmallMax];
    int32_t  b2[kSmallMax];
    uint32_t slot[kSmallMax];
    int32_t  first[kSmallMax + 1];
    uint32_t wsum[kSmallThreads / 64];
    uint32_t carry;
    uint32_t cut;
};

// exclusive scan of one value per thread across the workgroup; returns the exclusive
// prefix and leaves the block total in *total_out (read by every thread)
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* wsum, uint32_t* total_out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t t = __shfl_up(x, o); if (lane >= o) x += t; }
    __syncthreads();
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kSmallThreads / 64; w++) { const uint32_t c = wsum[w]; if (w < wave) woff += c; tot += c; }
    *total_out = tot;
    return woff + x - v;
}

__device__ __forceinline__ bool tuple_less(int32_t a1, int32_t a2, uint32_t as, int32_t b1, int32_t b2, uint32_t bs)
{
    if (a1 != b1) return a1 < b1;
    if (a2 != b2) return a2 < b2;
    return as < bs;
}

// out_counts[0] = clusters (or -1 on overflow), out_counts[1] = live evidence records
__global__ __launch_bounds__(kSmallThreads) void cluster_small_kernel(
    int32_t n_slots_host, const int32_t* __restrict__ n_slots_dev,
    const int32_t* __restrict__ cls, const int32_t* __restrict__ b1, const int32_t* __restrict__ b2,
    int32_t marker, int32_t tie_desc,
    int32_t* __restrict__ order, int32_t* __restrict__ cl_first, int32_t* __restrict__ cl_count,
    uint8_t* __restrict__ used, int32_t* __restrict__ out_counts)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    SmallLds& s = *reinterpret_cast<SmallLds*>(smem_raw);
    const int tid = threadIdx.x;
    const int32_t n_slots = n_slots_dev ? min(*n_slots_dev, n_slots_host) : n_slots_host;

    // 1. compaction in arrival order
    uint32_t carry = 0;
    for (int32_t base = 0; base < n_slots; base += kSmallThreads) {
        const int32_t i = base + tid;
        const bool live = (i < n_slots) && (cls[i] >= 0);
        uint32_t tot;
        const uint32_t r = carry + block_scan_excl(live ? 1u : 0u, s.wsum, &tot);
        if (live && r < (uint32_t)kSmallMax) { s.b1[r] = b1[i]; s.b2[r] = b2[i]; s.slot[r] = (uint32_t)i; }
        if (i < n_slots && used) used[i] = 0;
        carry += tot;
    }
    const uint32_t nv = carry;
    if (nv > (uint32_t)kSmallMax) {                  // caller must take the multi-kernel path
        if (tid == 0) { out_counts[0] = -1; out_counts[1] = (int32_t)nv; }
        return;
    }
    uint32_t P = 2;
    while (P < nv) P <<= 1;
    for (uint32_t i = nv + tid; i < P; i += kSmallThreads) { s.b1[i] = INT_MAX; s.b2[i] = INT_MAX; s.slot[i] = 0xFFFFFFFFu; }
    __syncthreads();

    // 2. bitonic sort, ascending in (b1, b2, slot); slot order = arrival order, so equal
    //    keys stay in arrival order like the reference's stable sort
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (P >> 1); t += kSmallThreads) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j clear
                const uint32_t hi = lo | j;
                const bool asc = (lo & k) == 0;
                const int32_t a1 = s.b1[lo], a2 = s.b2[lo], c1 = s.b1[hi], c2 = s.b2[hi];
                const uint32_t as = s.slot[lo], cs = s.slot[hi];
                const bool hi_lt_lo = tuple_less(c1, c2, cs, a1, a2, as);
                if (hi_lt_lo == asc) {
                    s.b1[lo] = c1; s.b2[lo] = c2; s.slot[lo] = cs;
                    s.b1[hi] = a1; s.b2[hi] = a2; s.slot[hi] = as;
                }
            }
            __syncthreads();
        }
    }

    // 3. marker cut: nodes are made only for the sorted prefix before the first b2 >= marker
    if (tid == 0) s.cut = nv;
    __syncthreads();
    {
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t p = tid; p < nv; p += kSmallThreads) if (s.b2[p] >= marker) { best = p; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
        if ((tid & 63) == 0 && best != 0xFFFFFFFFu) atomicMin(&s.cut, best);
    }
    __syncthreads();
    const uint32_t m = s.cut;

    // 4. run heads + cluster ids (8 consecutive positions per thread)
    constexpr int kPer = kSmallMax / kSmallThreads;
    uint32_t head[kPer];
    uint32_t nh = 0;
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t p = (uint32_t)tid * kPer + e;
        uint32_t h = 0;
        if (p < m) {
            h = (p == 0) || s.b1[p] != s.b1[p - 1] || s.b2[p] != s.b2[p - 1] ||
                cls[s.slot[p]] != cls[s.slot[p - 1]];
        }
        head[e] = h; nh += h;
    }
    uint32_t ncl;
    uint32_t cid = block_scan_excl(nh, s.wsum, &ncl);
    uint32_t mycid[kPer];
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t p = (uint32_t)tid * kPer + e;
        if (head[e]) { s.first[cid] = (int32_t)p; mycid[e] = cid; cid++; }
        else mycid[e] = cid - 1;
    }
    if (tid == 0) s.first[ncl] = (int32_t)m;
    __syncthreads();

    // 5. outputs
#pragma unroll
    for (int e = 0; e < kPer; e++) {
        const uint32_t p = (uint32_t)tid * kPer + e;
        if (p >= nv) continue;
        const uint32_t sl = s.slot[p];
        if (p >= m) { order[p] = (int32_t)sl; continue; }
        const uint32_t c = mycid[e];
        const int32_t f = s.first[c], cnt = s.first[c + 1] - f;
        if (head[e]) { cl_first[c] = f; cl_count[c] = cnt; }
        const int32_t pos = tie_desc ? (f + (cnt - 1 - ((int32_t)p - f))) : (int32_t)p;
        order[pos] = (int32_t)sl;
        if (used) used[sl] = 1;
    }
    if (tid == 0) { out_counts[0] = (int32_t)ncl; out_counts[1] = (int32_t)nv; }
}

// ---- breakpoint-histogram path ------------------------------------------------
//
// Split-read clusters are exact-key groups, so sorting every evidence record is more
// than the problem needs: hash the records into a breakpoint histogram (one table
// entry per distinct (b1,b2,class) with its support count), order only the DISTINCT
// breakpoints (typically 10x fewer than records), then drop every record into its
// cluster's slice and order each slice by arrival.  Four launches, every one of them
// many-workgroup; the table cleans itself for the next call.
//
//   hist_insert   live slot -> table entry (64-bit CAS), support count, dense list of new keys
//   hist_rank     distinct key -> its position in key order and the start of its slice, by
//                 counting the smaller keys and their supports (v_readlane broadcasts);
//                 the marker cut is a key comparison.  No sort network, no scan, no LDS.
//   hist_place    record -> order[first[entry] + cursor[cluster]++]
//   hist_finish   one wave per cluster: order the slice by slot (= arrival), reset the entry
//
// Limits (else counts[0] = -1 and the caller takes the radix path): kHistMaxKeys distinct
// breakpoints, kHistMaxSupport records per breakpoint.

constexpr int kHistMaxKeys = 8192;
constexpr int kHistMaxSupport = 1024;
constexpr uint64_t kEmptyKey = ~0ull;

struct HistScratch {
    uint64_t* keys;         // [H]
    uint32_t* cnt;          // [H]
    uint32_t* rank;         // [H] cluster id of a table entry, ~0 = behind the marker cut
    uint32_t* first_h;      // [H] where the entry's slice of order[] starts
    uint32_t* slot_h;       // [n_slots]
    uint32_t* uniq;         // [kHistMaxKeys] table positions in first-touch order
    uint64_t* ukey;         // [kHistMaxKeys] their keys, same order
    uint32_t* sorted_h;     // [kHistMaxKeys] table positions in key order
    uint32_t* cursor;       // [kHistMaxKeys] zero between calls
    uint32_t* acc_r;        // [kHistMaxKeys] hist_rank accumulators, zero between calls
    uint32_t* acc_f;        // [kHistMaxKeys]
    uint64_t* acc_kcut;     // [kHistMaxKeys / 64] kEmptyKey between calls
    uint32_t* acc_live;     // [kHistMaxKeys / 64]
    uint32_t* acc_done;     // [kHistMaxKeys / 64]
    uint32_t* misc;         // [0] distinct keys, [2] overflow, [3] clusters after the cut, [4] finish workgroups done
    uint32_t  H;
};

__device__ __forceinline__ uint64_t hist_key(int32_t cls, int32_t b1, int32_t b2)
{
    // ascending key order == ascending (b1, b2, class)
    return ((uint64_t)(uint32_t)b1 << 32) | (uint64_t)(((uint32_t)b2 << 1) | (uint32_t)(cls & 1));
}
__device__ __forceinline__ int32_t hist_key_b1(uint64_t k) { return (int32_t)(k >> 32); }
__device__ __forceinline__ int32_t hist_key_b2(uint64_t k) { return (int32_t)((uint32_t)k >> 1); }
__device__ __forceinline__ uint32_t hist_hash(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

__global__ __launch_bounds__(64) void hist_insert_kernel(int32_t n_slots, const int32_t* __restrict__ cls,
                                                         const int32_t* __restrict__ b1, const int32_t* __restrict__ b2,
                                                         HistScratch s, uint8_t* __restrict__ used)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        if (used) used[i] = 0;
        const int32_t c = cls[i], vb1 = b1[i], vb2 = b2[i];        // one round trip, live or not
        if (c < 0) { s.slot_h[i] = 0xFFFFFFFFu; continue; }
        const uint64_t key = hist_key(c, vb1, vb2);
        uint32_t h = hist_hash(key) & (s.H - 1);
        for (;;) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&s.keys[h]), kEmptyKey, key);
            if (old == kEmptyKey) {
                const uint32_t u = atomicAdd(&s.misc[0], 1u);
                if (u < (uint32_t)kHistMaxKeys) { s.uniq[u] = h; s.ukey[u] = key; }
                break;
            }
            if (old == key) break;
            h = (h + 1) & (s.H - 1);
        }
        atomicAdd(&s.cnt[h], 1u);
        s.slot_h[i] = h;
    }
}

// Every kernel of this path runs one-wave workgroups with no or little LDS, so that its workgroups fit
// the single wave slots a chip-filling realign launch of another stream leaves as its own one-wave
// workgroups retire (a four-wave workgroup needs a free slot on all four SIMDs of a CU at once and
// starves until that launch drains).

// One wave per (64 distinct keys, 64 distinct keys) tile: lane <-> key K of the first block, the second
// block's (key, support) pairs are fetched one per lane and broadcast lane by lane through v_readlane into
// scalar registers.  Every lane counts the keys smaller than K (= K's position in key order; keys are
// distinct) and the records those keys hold (= where K's slice of order[] starts); the tiles of one key
// block add their partial counts into acc_r / acc_f and the LAST tile to arrive (acc_done) writes the
// results and leaves the accumulators zero for the next call.  The marker cut (clusters exist only before
// the first key, in key order, whose b2 >= marker) is "K < kcut", kcut accumulated the same way per key
// block.  No sort, no scan, no LDS; everything cross-workgroup goes through agent-scope atomics.
__global__ __launch_bounds__(64) void hist_rank_kernel(HistScratch s, int32_t marker,
                                                      int32_t* __restrict__ cl_first, int32_t* __restrict__ cl_count,
                                                      int32_t* __restrict__ out_counts)
{
    const int lane = threadIdx.x;
    const uint32_t nu = s.misc[0];
    if (nu > (uint32_t)kHistMaxKeys) {
        if (blockIdx.x == 0 && lane == 0) { s.misc[2] = 1; s.misc[3] = 0; out_counts[0] = -1; out_counts[1] = 0; }
        return;
    }
    if (nu == 0) {
        if (blockIdx.x == 0 && lane == 0) { s.misc[3] = 0; out_counts[0] = 0; out_counts[1] = 0; }
        return;
    }
    const uint32_t nblk = (nu + 63u) / 64u;
    for (uint32_t w = blockIdx.x; w < nblk * nblk; w += gridDim.x) {
        const uint32_t pb = w / nblk, qb = w - pb * nblk;
        const uint32_t p = pb * 64u + (uint32_t)lane, q = qb * 64u + (uint32_t)lane;
        const uint64_t key = (p < nu) ? s.ukey[p] : kEmptyKey;
        const uint64_t kq = (q < nu) ? s.ukey[q] : kEmptyKey;               // the pad compares as "not smaller"
        const uint32_t cq = (q < nu) ? s.cnt[s.uniq[q]] : 0u;
        const uint32_t klo = (uint32_t)kq, khi = (uint32_t)(kq >> 32);
        uint32_t r = 0, f = 0;
#pragma unroll
        for (int e = 0; e < 64; e++) {
            const uint64_t ke = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)khi, e) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)klo, e);
            const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)cq, e);
            const bool lt = ke < key;
            r += lt ? 1u : 0u;
            f += lt ? ce : 0u;
        }
        // this tile's share of the cut key and of the live-record count
        uint64_t kc = (kq != kEmptyKey && hist_key_b2(kq) >= marker) ? kq : kEmptyKey;
        uint32_t live = cq;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)kc, o), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(kc >> 32), o);
            const uint64_t ok = ((uint64_t)ohi << 32) | olo;
            if (ok < kc) kc = ok;
            live += (uint32_t)__shfl_xor((int)live, o);
        }
        uint32_t seen = 0;
        if (p < nu) {
            seen = atomicAdd(&s.acc_r[p], r);
            seen |= atomicAdd(&s.acc_f[p], f);
        }
        if (lane == 0) {
            if (kc != kEmptyKey) seen |= (uint32_t)atomicMin(reinterpret_cast<unsigned long long*>(&s.acc_kcut[pb]), (unsigned long long)kc);
            seen |= atomicAdd(&s.acc_live[pb], live);
        }
        asm volatile("" :: "v"(seen));      // the returns are in: those atomics are performed before the count below
        uint32_t done = 0;
        if (lane == 0) done = atomicAdd(&s.acc_done[pb], 1u);
        done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);
        if (done != nblk - 1) continue;
        // last tile of key block pb: collect, publish, and zero the accumulators
        uint64_t kcut = kEmptyKey; uint32_t tot_live = 0;
        if (lane == 0) {
            kcut = atomicExch(reinterpret_cast<unsigned long long*>(&s.acc_kcut[pb]), (unsigned long long)kEmptyKey);
            tot_live = atomicExch(&s.acc_live[pb], 0u);
            s.acc_done[pb] = 0;
        }
        kcut = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kcut >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kcut);
        if (p < nu) {
            const uint32_t rr = atomicExch(&s.acc_r[p], 0u), ff = atomicExch(&s.acc_f[p], 0u);
            const uint32_t my_h = s.uniq[p];
            const uint32_t c = s.cnt[my_h];
            const bool valid = key < kcut;
            s.sorted_h[rr] = my_h;
            s.rank[my_h] = valid ? rr : 0xFFFFFFFFu;
            s.first_h[my_h] = ff;
            if (valid) { cl_first[rr] = (int32_t)ff; cl_count[rr] = (int32_t)c; if (c > (uint32_t)kHistMaxSupport) s.misc[2] = 1; }
            if (key == kcut) { s.misc[3] = rr; out_counts[0] = (int32_t)rr; }
        }
        if (pb == 0 && lane == 0) {
            if (kcut == kEmptyKey) { s.misc[3] = nu; out_counts[0] = (int32_t)nu; }
            out_counts[1] = (int32_t)tot_live;
        }
    }
}

__global__ __launch_bounds__(64) void hist_place_kernel(int32_t n_slots, HistScratch s,
                                                        int32_t* __restrict__ order, uint8_t* __restrict__ used)
{
    if (s.misc[0] > (uint32_t)kHistMaxKeys) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = s.slot_h[i];
        if (h == 0xFFFFFFFFu) continue;
        const uint32_t c = s.rank[h];
        if (c == 0xFFFFFFFFu) continue;
        const uint32_t p = atomicAdd(&s.cursor[c], 1u);
        order[s.first_h[h] + p] = (int32_t)i;
        if (used) used[i] = 1;
    }
}

// one wave per distinct key: order the cluster's slice by slot index, then reset the table entry.  The call's
// counters are cleared by whichever workgroup finishes LAST (misc[4] counts them): a workgroup scheduled
// late must still find the counters of this call.
__global__ __launch_bounds__(64) void hist_finish_kernel(HistScratch s, const int32_t* __restrict__ cl_first,
                                                        const int32_t* __restrict__ cl_count, int32_t tie_desc,
                                                        int32_t* __restrict__ order, int32_t* __restrict__ out_counts)
{
    __shared__ int32_t st[kHistMaxSupport];
    const int lane = threadIdx.x;
    const uint32_t nu_raw = s.misc[0];
    const bool overflow_keys = nu_raw > (uint32_t)kHistMaxKeys;
    const uint32_t nu = overflow_keys ? 0u : nu_raw;
    const uint32_t m = s.misc[3];
    const bool bad = s.misc[2] != 0;
    for (uint32_t p = blockIdx.x; p < nu; p += gridDim.x) {
        const uint32_t h = s.sorted_h[p];
        if (p < m && !bad) {
            const int32_t f = cl_first[p], cnt = cl_count[p];
            if (cnt > 1 || tie_desc) {
                for (int32_t t = lane; t < cnt; t += 64) st[t] = order[f + t];
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (int32_t t = lane; t < cnt; t += 64) {
                    const int32_t v = st[t];
                    int32_t r = 0;
                    for (int32_t j = 0; j < cnt; j++) r += (st[j] < v) ? 1 : 0;
                    order[f + (tie_desc ? (cnt - 1 - r) : r)] = v;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) { s.keys[h] = kEmptyKey; s.cnt[h] = 0; s.cursor[p] = 0; }
    }
    if (lane == 0) {
        if (blockIdx.x == 0 && bad && !overflow_keys) out_counts[0] = -1;
        // no fence: the count only has to follow this workgroup's own reads of the counters, which its
        // loop bounds consumed long ago (an agent-scope fence here is a whole-L2 write-back per workgroup)
        const uint32_t done = atomicAdd(&s.misc[4], 1u);
        if (done == gridDim.x - 1) {
            s.misc[4] = 0;
            if (!overflow_keys) { s.misc[0] = 0; s.misc[1] = 0; s.misc[2] = 0; s.misc[3] = 0; }
        }
    }
}

inline uint32_t hist_table_size(int32_t n_slots)
{
    uint32_t H = 1024;
    while (H < 2u * (uint32_t)(n_slots > 0 ? n_slots : 1)) H <<= 1;
    return H;
}

inline size_t hist_carve(HistScratch* hs, void* base, int32_t n_slots)
{
    const uint32_t H = hist_table_size(n_slots);
    const size_t nn = (size_t)(n_slots > 0 ? n_slots : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t oK = take((size_t)H * 8), oC = take((size_t)H * 4), oR = take((size_t)H * 4), oS = take(nn * 4);
    const size_t oU = take((size_t)kHistMaxKeys * 4), oSh = take((size_t)kHistMaxKeys * 4), oCu = take((size_t)kHistMaxKeys * 4), oM = take(64);
    const size_t oUk = take((size_t)kHistMaxKeys * 8), oFh = take((size_t)H * 4);
    const size_t oAr = take((size_t)kHistMaxKeys * 4), oAf = take((size_t)kHistMaxKeys * 4), oAk = take((size_t)kHistMaxKeys / 64 * 8);
    const size_t oAl = take((size_t)kHistMaxKeys / 64 * 4), oAd = take((size_t)kHistMaxKeys / 64 * 4);
    if (hs) {
        char* b = static_cast<char*>(base);
        hs->keys = (uint64_t*)(b + oK); hs->cnt = (uint32_t*)(b + oC); hs->rank = (uint32_t*)(b + oR); hs->slot_h = (uint32_t*)(b + oS);
        hs->uniq = (uint32_t*)(b + oU); hs->sorted_h = (uint32_t*)(b + oSh); hs->cursor = (uint32_t*)(b + oCu); hs->misc = (uint32_t*)(b + oM);
        hs->ukey = (uint64_t*)(b + oUk); hs->first_h = (uint32_t*)(b + oFh);
        hs->acc_r = (uint32_t*)(b + oAr); hs->acc_f = (uint32_t*)(b + oAf); hs->acc_kcut = (uint64_t*)(b + oAk);
        hs->acc_live = (uint32_t*)(b + oAl); hs->acc_done = (uint32_t*)(b + oAd);
        hs->H = H;
    }
    return off;
}

// ---- evidence gather ----------------------------------------------------------

__global__ __launch_bounds__(256) void count_ev_kernel(const im_read_result* __restrict__ res, int32_t n, uint32_t* __restrict__ cnt)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        cnt[i] = (res[i].status == IM_ST_EVIDENCE) ? (uint32_t)res[i].n_ev : 0u;
}

__global__ __launch_bounds__(256) void write_ev_kernel(const im_read_result* __restrict__ res, int32_t n, const uint32_t* __restrict__ offs,
                                                      const uint32_t* __restrict__ total, int32_t cap,
                                                      int32_t* __restrict__ cls, int32_t* __restrict__ b1, int32_t* __restrict__ b2,
                                                      int32_t* __restrict__ src, int32_t* __restrict__ n_out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (res[i].status != IM_ST_EVIDENCE) continue;
        const int ne = res[i].n_ev;
        for (int k = 0; k < ne && k < IM_MAX_EV; k++) {
            const int64_t o = (int64_t)offs[i] + k;
            if (o >= cap) break;
            cls[o] = res[i].ev[k].cls; b1[o] = res[i].ev[k].b1; b2[o] = res[i].ev[k].b2;
            src[o] = (int32_t)(i * IM_MAX_EV + k);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = (int32_t)min(*total, (uint32_t)cap);
}

inline int grid_for(int64_t n, int threads)
{
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > 256 * 8) b = 256 * 8;
    return (int)b;
}


struct ClusterScratch {
    uint64_t *keysA, *keysB;
    uint32_t *valsA, *valsB, *hist, *head, *cid, *misc;      // misc[0] = cut, misc[1] = n heads
    int32_t* first_tmp;
    int nblocks;
};

inline size_t carve(ClusterScratch* cs, void* base, int32_t n)
{
    const size_t nn = (size_t)(n > 0 ? n : 1);
    const int nblocks = (int)((nn + kSortTile - 1) / kSortTile);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t oKA = take(nn * 8), oKB = take(nn * 8), oVA = take(nn * 4), oVB = take(nn * 4);
    const size_t oH = take((size_t)nblocks * 256 * 4), oHead = take(nn * 4), oCid = take(nn * 4);
    const size_t oMisc = take(64), oFirst = take((nn + 1) * 4);
    if (cs) {
        char* b = static_cast<char*>(base);
        cs->keysA = (uint64_t*)(b + oKA); cs->keysB = (uint64_t*)(b + oKB);
        cs->valsA = (uint32_t*)(b + oVA); cs->valsB = (uint32_t*)(b + oVB);
        cs->hist = (uint32_t*)(b + oH); cs->head = (uint32_t*)(b + oHead); cs->cid = (uint32_t*)(b + oCid);
        cs->misc = (uint32_t*)(b + oMisc); cs->first_tmp = (int32_t*)(b + oFirst);
        cs->nblocks = nblocks;
    }
    return off;
}

}  // namespace

size_t cluster_scratch_bytes(int32_t n) { return carve(nullptr, nullptr, n); }

hipError_t launch_cluster_sr(int32_t n_cap, const int32_t* n_dev,
                             const int32_t* cls, const int32_t* b1, const int32_t* b2,
                             int32_t marker, int32_t tie_desc,
                             int32_t* order, int32_t* cl_first, int32_t* cl_count,
                             uint8_t* used, int32_t* n_clusters,
                             void* scratch, size_t scratch_bytes, hipStream_t stream)
{
    if (n_cap <= 0) return hipMemsetAsync(n_clusters, 0, sizeof(int32_t), stream);
    ClusterScratch cs;
    if (carve(&cs, scratch, n_cap) > scratch_bytes) return hipErrorInvalidValue;
    const int g256 = grid_for(n_cap, 256);
    hipLaunchKernelGGL(build_keys_kernel, dim3(g256), dim3(256), 0, stream, n_dev, n_cap, b1, b2, cs.keysA, cs.valsA);
    uint64_t *kin = cs.keysA, *kout = cs.keysB;
    uint32_t *vin = cs.valsA, *vout = cs.valsB;
    const int64_t nh = (int64_t)cs.nblocks * 256;
    for (int pass = 0; pass < 8; pass++) {
        const int shift = pass * 8;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(cs.nblocks), dim3(kSortThreads), 0, stream, kin, n_dev, n_cap, shift, cs.hist, cs.nblocks);
        hipLaunchKernelGGL(scan_excl_kernel, dim3(1), dim3(kScanThreads), 0, stream, cs.hist, cs.hist, nh, (const int32_t*)nullptr, (uint32_t*)nullptr);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(cs.nblocks), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n_dev, n_cap, shift, cs.hist, cs.nblocks);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    // 8 passes: the sorted data is back in keysA / valsA (kin / vin)
    hipError_t e = hipMemsetAsync(cs.misc, 0xFF, sizeof(uint32_t), stream);      // cut = "none"
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cut_kernel, dim3(g256), dim3(256), 0, stream, kin, n_dev, n_cap, marker, cs.misc);
    hipLaunchKernelGGL(heads_kernel, dim3(g256), dim3(256), 0, stream, kin, vin, cls, n_dev, n_cap, cs.misc, cs.head, used);
    hipLaunchKernelGGL(scan_excl_kernel, dim3(1), dim3(kScanThreads), 0, stream, cs.head, cs.cid, (int64_t)n_cap, n_dev, cs.misc + 1);
    hipLaunchKernelGGL(cluster_first_kernel, dim3(g256), dim3(256), 0, stream, cs.head, cs.cid, n_dev, n_cap, cs.misc, cs.misc + 1, cs.first_tmp, n_clusters);
    hipLaunchKernelGGL(cluster_finish_kernel, dim3(g256), dim3(256), 0, stream, vin, cs.head, cs.cid, n_dev, n_cap, cs.misc, cs.first_tmp, tie_desc, order, cl_first, cl_count);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void cluster_records_kernel(int32_t tid, const int32_t* __restrict__ counts,
                                                             const int32_t* __restrict__ order, const int32_t* __restrict__ cl_first,
                                                             const int32_t* __restrict__ cl_count,
                                                             const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                             const int32_t* __restrict__ b2, int4* __restrict__ recs, int32_t cap)
{
    const int32_t ncl = max(counts[0], 0);
    const int32_t lim = min(ncl, cap - 1);
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < lim; c += (int64_t)gridDim.x * blockDim.x) {
        const int32_t e = order[cl_first[c]];
        recs[1 + c] = make_int4(tid, b1[e], b2[e], (cls[e] << 24) | (cl_count[c] & 0xFFFFFF));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) recs[0] = make_int4(counts[0], counts[1], tid, ncl > cap - 1 ? 1 : 0);
}

hipError_t launch_cluster_records(int32_t tid, const int32_t* counts, const int32_t* order, const int32_t* cl_first,
                                  const int32_t* cl_count, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                                  int32_t* recs, int32_t cap, hipStream_t stream)
{
    hipLaunchKernelGGL(cluster_records_kernel, dim3(grid_for(cap, 256)), dim3(256), 0, stream,
                       tid, counts, order, cl_first, cl_count, cls, b1, b2, reinterpret_cast<int4*>(recs), cap);
    return hipGetLastError();
}

int cluster_small_max() { return kSmallMax; }

size_t cluster_hist_scratch_bytes(int32_t n_slots) { return hist_carve(nullptr, nullptr, n_slots); }

// scratch must be prepared once with launch_cluster_hist_init (and again after an overflow return)
hipError_t launch_cluster_hist_init(int32_t n_slots, void* scratch, size_t scratch_bytes, hipStream_t stream)
{
    HistScratch hs;
    if (hist_carve(&hs, scratch, n_slots) > scratch_bytes) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(hs.keys, 0xFF, (size_t)hs.H * 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.cnt, 0, (size_t)hs.H * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.misc, 0, 64, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.cursor, 0, (size_t)kHistMaxKeys * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.acc_r, 0, (size_t)kHistMaxKeys * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.acc_f, 0, (size_t)kHistMaxKeys * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.acc_kcut, 0xFF, (size_t)kHistMaxKeys / 64 * 8, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.acc_live, 0, (size_t)kHistMaxKeys / 64 * 4, stream);
    if (e == hipSuccess) e = hipMemsetAsync(hs.acc_done, 0, (size_t)kHistMaxKeys / 64 * 4, stream);
    return e;
}

hipError_t launch_cluster_hist(int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                               int32_t marker, int32_t tie_desc,
                               int32_t* order, int32_t* cl_first, int32_t* cl_count,
                               uint8_t* used, int32_t* out_counts,
                               void* scratch, size_t scratch_bytes, hipStream_t stream)
{
    HistScratch hs;
    if (hist_carve(&hs, scratch, n_slots) > scratch_bytes) return hipErrorInvalidValue;
    const int g = grid_for(n_slots, 64);
    hipLaunchKernelGGL(hist_insert_kernel, dim3(g), dim3(64), 0, stream, n_slots, cls, b1, b2, hs, used);
    hipLaunchKernelGGL(hist_rank_kernel, dim3(1024), dim3(64), 0, stream, hs, marker, cl_first, cl_count, out_counts);
    hipLaunchKernelGGL(hist_place_kernel, dim3(g), dim3(64), 0, stream, n_slots, hs, order, used);
    hipLaunchKernelGGL(hist_finish_kernel, dim3(512), dim3(64), 0, stream, hs, cl_first, cl_count, tie_desc, order, out_counts);
    return hipGetLastError();
}

hipError_t launch_cluster_small(int32_t n_slots, const int32_t* n_slots_dev,
                                const int32_t* cls, const int32_t* b1, const int32_t* b2,
                                int32_t marker, int32_t tie_desc,
                                int32_t* order, int32_t* cl_first, int32_t* cl_count,
                                uint8_t* used, int32_t* out_counts, hipStream_t stream)
{
    // the attribute belongs to the (kernel, device) pair: set once per device, whichever thread / context gets there first
    static std::once_flag once[64];
    static hipError_t once_rc[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    dev &= 63;
    std::call_once(once[dev], [dev] {
        once_rc[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(cluster_small_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SmallLds));
    });
    if (once_rc[dev] != hipSuccess) return once_rc[dev];
    hipLaunchKernelGGL(cluster_small_kernel, dim3(1), dim3(kSmallThreads), sizeof(SmallLds), stream,
                       n_slots, n_slots_dev, cls, b1, b2, marker, tie_desc, order, cl_first, cl_count, used, out_counts);
    return hipGetLastError();
}

size_t gather_scratch_bytes(int32_t n) { return align_up((size_t)(n > 0 ? n : 1) * 4, 256) * 2 + 256; }

hipError_t launch_gather_evidence(const im_read_result* res, int32_t n,
                                  int32_t* cls, int32_t* b1, int32_t* b2, int32_t* src,
                                  int32_t cap, int32_t* n_out, void* scratch, size_t scratch_bytes,
                                  hipStream_t stream)
{
    if (n <= 0) return hipMemsetAsync(n_out, 0, sizeof(int32_t), stream);
    if (gather_scratch_bytes(n) > scratch_bytes) return hipErrorInvalidValue;
    const size_t stride = align_up((size_t)n * 4, 256);
    uint32_t* cnt = (uint32_t*)scratch;
    uint32_t* offs = (uint32_t*)((char*)scratch + stride);
    uint32_t* total = (uint32_t*)((char*)scratch + 2 * stride);
    const int g256 = grid_for(n, 256);
    hipLaunchKernelGGL(count_ev_kernel, dim3(g256), dim3(256), 0, stream, res, n, cnt);
    hipLaunchKernelGGL(scan_excl_kernel, dim3(1), dim3(kScanThreads), 0, stream, cnt, offs, (int64_t)n, (const int32_t*)nullptr, total);
    hipLaunchKernelGGL(write_ev_kernel, dim3(g256), dim3(256), 0, stream, res, n, offs, total, cap, cls, b1, b2, src, n_out);
    return hipGetLastError();
}

}  // namespace im
