// im_flushwide.hip -- the READCHUNK flushes of a group of contigs AND the split-read group-by, chip-wide, in three launches.
//
// im_flush.hip walks the flush list with one workgroup because "what flush f consumes is not pending for f + 1".  It does
// not have to be walked.  find_marker (src/indelminer.c:211-233) is the minimum over a pair table whose entries only leave
// it or enter it at the current position of a coordinate-sorted walk, and a flush's marker is min(that, the current
// position) (622-623): within a contig the markers never decrease.  process_evidence (117-146) consumes the sorted prefix
// in front of the first entry with b2 >= marker, so everything a flush f' consumed had b2 < marker(f') <= marker(f) for
// every later flush f: it could not have been f's cutting entry anyway.  Hence, with NO history,
//
//     cut(f)      = min { (b1,b2)(e) : e arrived before f's bounds, b2(e) >= marker(f) }           over all entries of the contig
//     consumed(e) = the first flush f at or after e's arrival with (b1,b2)(e) < cut(f)
//
// and because the markers are monotone the flushes an entry is a cutting candidate of are a contiguous run
// [first flush after its arrival, first flush whose marker exceeds its b2) -- empty for nearly every entry.
//
//   launch 1  flushw_cut_kernel    every entry hands its key to the cuts of its run: a range-minimum update of a small
//                                  segment tree over the flush list (one or two 64-bit atomicMin for a short run, ~log F for a
//                                  run to the contig's end -- the case of markers pinned low by stale pair-table entries --
//                                  which a wave whose entries share the run settles with one update for all 64)
//   launch 2  flushw_mark_kernel   every entry finds its flush (usually the first it looks at), writes consumed[], and a
//                                  consumed split-read slot enters the cluster table keyed (flush, class, b1, b2): a 64-bit
//                                  CAS of (flush id | representative slot), a count and the smallest member slot per entry
//   launch 3  flushw_emit_kernel   the smallest member of each cluster takes the cluster's record and its run of order[] (one
//                                  64-bit atomicAdd per WAVE for all its clusters) and the wave collects the members in slot
//                                  order = arrival order: members of a cluster arrive within a few hundred slots
//                                  of each other, so that is a handful of coalesced 64-slot sweeps -- no placement pass, no
//                                  ordering pass, no offsets scan.  The table and the tree leave clean for the next call.
//
// HBM streaming: 16 B per candidate and array read twice, 32 B written, 4 B per slot read once more; the table is touched by
// consumed slots only.

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kWin = 64;                 // flush descriptors a workgroup keeps in LDS (a workgroup's candidates span 1-2 flushes)
constexpr int kWideBlock = 256;

__device__ __forceinline__ uint64_t cut_key(int32_t b1, int32_t b2) { return ((uint64_t)(uint32_t)b1 << 32) | (uint32_t)b2; }

struct WideScratch {
    unsigned long long* head;       // [H] (flush id << 32) | (representative slot + 1), 0 = empty
    uint32_t* cnt;                  // [H] members
    uint32_t* first_slot;           // [H] smallest member slot, ~0 between calls
    uint32_t* slot_h;               // [n_slots] table position of a consumed split-read slot, ~0 otherwise
    unsigned long long* tree;       // [2 * F2cap] range-minimum tree over the flush list (heap order, leaves at F2 + f), ~0 between calls
    uint32_t H, F2cap;
};

struct WideArgs {
    const im_flush_desc* desc; int32_t n_fl; uint32_t F2;
    const int32_t *cls, *b1, *b2; int32_t* consumed;
    const int32_t* cand_rec; const int32_t* n_cand; int32_t cand_cap;
    int32_t pe_base, pe_count;
    WideScratch s;
};

// the flush list through a workgroup's LDS window
struct FlushView {
    const im_flush_desc* g; int32_t n_fl;
    const im_flush_desc* w; int32_t lo, n;
    __device__ __forceinline__ bool in(int32_t f) const { return (uint32_t)(f - lo) < (uint32_t)n; }
    __device__ __forceinline__ int32_t rec1(int32_t f) const { return in(f) ? w[f - lo].rec1 : g[f].rec1; }
    __device__ __forceinline__ int32_t marker(int32_t f) const { return in(f) ? w[f - lo].marker : g[f].marker; }
    __device__ __forceinline__ int32_t id(int32_t f) const { return in(f) ? w[f - lo].id : g[f].id; }
    __device__ __forceinline__ int32_t last(int32_t f) const
    {
        const int32_t l = in(f) ? w[f - lo].last : g[f].last;
        return l < f ? f : (l >= n_fl ? n_fl - 1 : l);
    }
};

__device__ __forceinline__ int32_t first_flush_rec(const im_flush_desc* d, int32_t n, int32_t rec)     // first f with rec1 > rec
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (d[mid].rec1 <= rec) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int32_t first_flush_pe(const im_flush_desc* d, int32_t n, int32_t i)        // first f with pe1 > i
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (d[mid].pe1 <= i) lo = mid + 1; else hi = mid; }
    return lo;
}

// the flushes entry (arrival fa, b2) is a cutting candidate of: [fa, fh); fh = the first flush of the contig whose marker
// exceeds b2 (monotone markers: a binary search), fa itself when it is a candidate of none
__device__ __forceinline__ int32_t run_end(const FlushView& V, int32_t fa, int32_t last, int32_t vb2)
{
    if (V.marker(fa) > vb2) return fa;
    int32_t lo = fa + 1, hi = last + 1;
    while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (V.marker(mid) > vb2) hi = mid; else lo = mid + 1; }
    return lo;
}

__device__ __forceinline__ void tree_min(unsigned long long* T, uint32_t F2, int32_t a, int32_t h, uint64_t key)
{
    uint32_t l = (uint32_t)a + F2, r = (uint32_t)h + F2;
    while (l < r) {
        if (l & 1u) atomicMin(&T[l++], (unsigned long long)key);
        if (r & 1u) atomicMin(&T[--r], (unsigned long long)key);
        l >>= 1; r >>= 1;
    }
}
__device__ __forceinline__ uint64_t tree_get(const unsigned long long* T, uint32_t F2, int32_t f)
{
    uint64_t m = ~0ull;
    for (uint32_t i = (uint32_t)f + F2; i >= 1u; i >>= 1) { const uint64_t v = T[i]; if (v < m) m = v; }
    return m;
}

__device__ __forceinline__ uint64_t wave_min64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o);
        const uint64_t ov = ((uint64_t)hi << 32) | lo;
        if (ov < v) v = ov;
    }
    return v;
}

// one wave-wide step of launch 1: lanes with `live` give key to the cuts of [fa, fh).  A wave whose live lanes share the run
// (the rule when markers are pinned low: every entry is a candidate up to its contig's end) settles it with one update.
__device__ __forceinline__ void give_keys(const WideArgs& A, bool live, int32_t fa, int32_t fh, uint64_t key, int lane)
{
    const uint64_t lv = __ballot(live);
    if (lv == 0ull) return;
    const int first = __ffsll((unsigned long long)lv) - 1;
    const int32_t ufa = __shfl(fa, first), ufh = __shfl(fh, first);
    if (__all(!live || (fa == ufa && fh == ufh))) {
        const uint64_t k = wave_min64(live ? key : ~0ull);
        if (lane == first) tree_min(A.s.tree, A.F2, ufa, ufh, k);
    } else if (live) tree_min(A.s.tree, A.F2, fa, fh, key);
}

// the workgroup's share of the candidates [c0, c1) and its window of the flush list in LDS; returns the first flush of the window
__device__ __forceinline__ int32_t stage_window(const WideArgs& A, im_flush_desc* s_desc, int32_t* s_flo, int32_t& c0, int32_t& c1, int32_t& nst)
{
    const int t = threadIdx.x;
    const int32_t ncand = min(*A.n_cand, A.cand_cap);
    const int64_t per = ((int64_t)ncand + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per;
    c0 = (int32_t)(b0 < ncand ? b0 : ncand);
    c1 = (int32_t)(b0 + per < ncand ? b0 + per : ncand);
    if (t == 0) *s_flo = c0 < c1 ? first_flush_rec(A.desc, A.n_fl, A.cand_rec[c0]) : A.n_fl;
    __syncthreads();
    const int32_t flo = *s_flo;
    nst = A.n_fl - flo;
    if (nst > kWin) nst = kWin;
    if (nst < 0) nst = 0;
    if (t < 2 * nst) reinterpret_cast<int4*>(s_desc)[t] = reinterpret_cast<const int4*>(A.desc + flo)[t];
    __syncthreads();
    return flo;
}

__global__ __launch_bounds__(kWideBlock) void flushw_cut_kernel(WideArgs A, int32_t* __restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) im_flush_desc s_desc[kWin];
    __shared__ int32_t s_flo;
    const int t = threadIdx.x, lane = t & 63;
    if (blockIdx.x == 0 && t == 0) { counts[0] = 0; counts[1] = 0; }       // launch 3 counts clusters and nodes up from here
    int32_t c0, c1, nst;
    const int32_t flo = stage_window(A, s_desc, &s_flo, c0, c1, nst);
    const FlushView V = { A.desc, A.n_fl, s_desc, flo, nst };
    const int4* cls4 = reinterpret_cast<const int4*>(A.cls);
    const int4* b14 = reinterpret_cast<const int4*>(A.b1);
    const int4* b24 = reinterpret_cast<const int4*>(A.b2);
    for (int32_t base = c0; base < c1; base += kWideBlock) {
        const int32_t c = base + t;
        const bool have = c < c1;
        int32_t fa = A.n_fl, last = 0;
        int4 vc = make_int4(-1, -1, -1, -1), v1 = make_int4(0, 0, 0, 0), v2 = v1;
        if (have) {
            const int32_t rec = A.cand_rec[c];
            vc = cls4[c]; v1 = b14[c]; v2 = b24[c];
            fa = flo;
            while (fa < A.n_fl && V.rec1(fa) <= rec) fa++;
            if (fa < A.n_fl) last = V.last(fa);
        }
        const bool ok = have && fa < A.n_fl;
        const int32_t sc[4] = { vc.x, vc.y, vc.z, vc.w }, s1[4] = { v1.x, v1.y, v1.z, v1.w }, s2[4] = { v2.x, v2.y, v2.z, v2.w };
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bool live = ok && sc[j] >= 0;
            int32_t fh = fa;
            if (live) { fh = run_end(V, fa, last, s2[j]); live = fh > fa; }
            give_keys(A, live, fa, fh, cut_key(s1[j], s2[j]), lane);
        }
    }
    // the paired-read entries (a few per thousand reads): the same, each through its own look-up
    const int32_t npe_round = (A.pe_count + 63) & ~63;
    for (int32_t i = blockIdx.x * kWideBlock + t; i < npe_round; i += gridDim.x * kWideBlock) {
        bool live = false;
        int32_t fa = 0, fh = 0; uint64_t key = ~0ull;
        if (i < A.pe_count) {
            const int32_t sl = A.pe_base + i;
            fa = first_flush_pe(A.desc, A.n_fl, i);
            if (fa < A.n_fl && A.cls[sl] >= 0) {
                const int32_t vb1 = A.b1[sl], vb2 = A.b2[sl];
                const FlushView G = { A.desc, A.n_fl, s_desc, 0, 0 };
                fh = run_end(G, fa, G.last(fa), vb2);
                live = fh > fa; key = cut_key(vb1, vb2);
            }
        }
        give_keys(A, live, fa, fh, key, lane);
    }
}

__device__ __forceinline__ uint32_t mix32(uint32_t f, uint32_t c, uint32_t x1, uint32_t x2)
{
    uint64_t k = ((uint64_t)x1 << 32) | x2;
    k ^= ((uint64_t)f << 17) | c;
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

// the flush that consumes an entry: the first f of [fh, last] with key < cut(f); 0 when none does
template <class CutOf>
__device__ __forceinline__ int32_t consuming_flush(const FlushView& V, int32_t fh, int32_t last, uint64_t key, CutOf cut_of)
{
    for (int32_t f = fh; f <= last; f++) if (key < cut_of(f)) return V.id(f);
    return 0;
}

__global__ __launch_bounds__(kWideBlock) void flushw_mark_kernel(WideArgs A)
{
    __shared__ __attribute__((aligned(16))) im_flush_desc s_desc[kWin];
    __shared__ unsigned long long s_cut[kWin];
    __shared__ int32_t s_flo;
    const int t = threadIdx.x;
    int32_t c0, c1, nst;
    const int32_t flo = stage_window(A, s_desc, &s_flo, c0, c1, nst);
    if (t < nst) s_cut[t] = tree_get(A.s.tree, A.F2, flo + t);
    __syncthreads();
    const FlushView V = { A.desc, A.n_fl, s_desc, flo, nst };
    auto cut_of = [&](int32_t f) -> uint64_t { return V.in(f) ? (uint64_t)s_cut[f - flo] : tree_get(A.s.tree, A.F2, f); };
    const int4* cls4 = reinterpret_cast<const int4*>(A.cls);
    const int4* b14 = reinterpret_cast<const int4*>(A.b1);
    const int4* b24 = reinterpret_cast<const int4*>(A.b2);
    for (int32_t c = c0 + t; c < c1; c += kWideBlock) {
        const int32_t rec = A.cand_rec[c];
        const int4 vc = cls4[c], v1 = b14[c], v2 = b24[c];
        int32_t fa = flo;
        while (fa < A.n_fl && V.rec1(fa) <= rec) fa++;
        const bool ok = fa < A.n_fl;
        const int32_t last = ok ? V.last(fa) : 0;
        const int32_t sc[4] = { vc.x, vc.y, vc.z, vc.w }, s1[4] = { v1.x, v1.y, v1.z, v1.w }, s2[4] = { v2.x, v2.y, v2.z, v2.w };
        int32_t ids[4]; uint32_t hs[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ids[j] = 0; hs[j] = 0xFFFFFFFFu;
            if (!ok || sc[j] < 0) continue;
            const uint64_t key = cut_key(s1[j], s2[j]);
            const int32_t id = consuming_flush(V, run_end(V, fa, last, s2[j]), last, key, cut_of);
            ids[j] = id;
            if (id <= 0 || sc[j] >= 2) continue;
            // the cluster table: one entry per distinct (flush, class, b1, b2); the entry names a representative slot, whose
            // class / b1 / b2 are compared through the slot arrays, and carries the flush id itself (consumed[] of another
            // slot is being written by this very launch)
            const uint32_t slot = (uint32_t)c * IM_MAX_EV + (uint32_t)j;
            const unsigned long long mine = ((unsigned long long)(uint32_t)id << 32) | (slot + 1u);
            uint32_t h = mix32((uint32_t)id, (uint32_t)sc[j], (uint32_t)s1[j], (uint32_t)s2[j]) & (A.s.H - 1u);
            for (;;) {
                const unsigned long long old = atomicCAS(&A.s.head[h], 0ull, mine);
                if (old == 0ull) break;
                if ((uint32_t)(old >> 32) == (uint32_t)id) {
                    const uint32_t rep = (uint32_t)old - 1u;
                    if (A.cls[rep] == sc[j] && A.b1[rep] == s1[j] && A.b2[rep] == s2[j]) break;
                }
                h = (h + 1u) & (A.s.H - 1u);
            }
            atomicAdd(&A.s.cnt[h], 1u);
            atomicMin(&A.s.first_slot[h], slot);
            hs[j] = h;
        }
        reinterpret_cast<int4*>(A.consumed)[c] = make_int4(ids[0], ids[1], ids[2], ids[3]);
        reinterpret_cast<uint4*>(A.s.slot_h)[c] = make_uint4(hs[0], hs[1], hs[2], hs[3]);
    }
    for (int32_t i = blockIdx.x * kWideBlock + t; i < A.pe_count; i += gridDim.x * kWideBlock) {
        const int32_t sl = A.pe_base + i;
        int32_t id = 0;
        const int32_t fa = first_flush_pe(A.desc, A.n_fl, i);
        if (fa < A.n_fl && A.cls[sl] >= 0) {
            const FlushView G = { A.desc, A.n_fl, s_desc, 0, 0 };
            const int32_t vb1 = A.b1[sl], vb2 = A.b2[sl], last = G.last(fa);
            id = consuming_flush(G, run_end(G, fa, last, vb2), last, cut_key(vb1, vb2),
                                 [&](int32_t f) -> uint64_t { return tree_get(A.s.tree, A.F2, f); });
        }
        A.consumed[sl] = id;
    }
}

__global__ __launch_bounds__(kWideBlock) void flushw_emit_kernel(WideArgs A, int32_t tie_desc, int32_t* __restrict__ order, int32_t* __restrict__ cl_key,
                                                                int32_t* __restrict__ cl_first, int32_t* __restrict__ cl_count,
                                                                unsigned long long* __restrict__ counts64)
{
    const int t = threadIdx.x, lane = t & 63;
    const int64_t gtid = (int64_t)blockIdx.x * kWideBlock + t, gstride = (int64_t)gridDim.x * kWideBlock;
    // the cuts are done with: the tree is all ones again for the next call
    for (int64_t i = gtid; i < 2 * (int64_t)A.F2; i += gstride) A.s.tree[i] = ~0ull;
    const int64_t nc = min(*A.n_cand, A.cand_cap);
    const int64_t n_slots = nc * IM_MAX_EV;
    const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int64_t base = gtid - lane; base < n_slots; base += gstride) {         // wave-uniform
        const int64_t i = base + lane;
        const uint32_t h = i < n_slots ? A.s.slot_h[i] : 0xFFFFFFFFu;
        const bool isfirst = h != 0xFFFFFFFFu && A.s.first_slot[h] == (uint32_t)i;
        const uint64_t mask = __ballot(isfirst);
        if (mask == 0ull) continue;
        const uint32_t cnt = isfirst ? A.s.cnt[h] : 0u;
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += y; }
        const uint32_t tot = (uint32_t)__shfl((int)incl, 63);
        const uint32_t ncl = (uint32_t)__popcll((unsigned long long)mask);
        // the wave's clusters and their runs of order[] in one atomic: low word = clusters, high word = nodes (= counts[0], counts[1])
        unsigned long long got = 0ull;
        if (lane == 0) got = atomicAdd(counts64, ((unsigned long long)tot << 32) | ncl);
        const uint32_t u0 = (uint32_t)__shfl((int)(uint32_t)got, 0), n0 = (uint32_t)__shfl((int)(uint32_t)(got >> 32), 0);
        const uint32_t my_first = n0 + incl - cnt;
        if (isfirst) {
            const uint32_t u = u0 + (uint32_t)__popcll((unsigned long long)(mask & below));
            cl_first[u] = (int32_t)my_first; cl_count[u] = (int32_t)cnt;
            reinterpret_cast<int4*>(cl_key)[u] = make_int4((int32_t)(A.s.head[h] >> 32), A.cls[i], A.b1[i], A.b2[i]);
            // the table cleans itself: only this lane still looks at the entry
            A.s.head[h] = 0ull; A.s.cnt[h] = 0u; A.s.first_slot[h] = 0xFFFFFFFFu;
        }
        // the members of each of the wave's clusters, in slot order: 64-slot sweeps from the smallest member on
        uint64_t m = mask;
        while (m != 0ull) {
            const int L = __ffsll((unsigned long long)m) - 1;
            m &= m - 1ull;
            const uint32_t hL = (uint32_t)__shfl((int)h, L), cL = (uint32_t)__shfl((int)cnt, L), fL = (uint32_t)__shfl((int)my_first, L);
            uint32_t found = 0;
            for (int64_t j0 = base + L; found < cL && j0 < n_slots; j0 += 64) {
                const int64_t j = j0 + lane;
                const bool hit = j < n_slots && A.s.slot_h[j] == hL;
                const uint64_t bal = __ballot(hit);
                if (hit) {
                    const uint32_t k = found + (uint32_t)__popcll((unsigned long long)(bal & below));
                    order[fL + (tie_desc ? cL - 1u - k : k)] = (int32_t)j;
                }
                found += (uint32_t)__popcll((unsigned long long)bal);
            }
        }
    }
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
inline uint32_t pow2_at_least(uint64_t x) { uint32_t p = 1; while ((uint64_t)p < x) p <<= 1; return p; }

inline size_t wide_carve(WideScratch* g, void* base, int32_t n_slots, int32_t n_fl_cap)
{
    const size_t nn = (size_t)(n_slots > 0 ? n_slots : 1);
    const uint32_t H = pow2_at_least(2 * (uint64_t)nn < 1024 ? 1024 : 2 * (uint64_t)nn);
    const uint32_t F2 = pow2_at_least((uint64_t)(n_fl_cap > 0 ? n_fl_cap : 1));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    const size_t oHead = take((size_t)H * 8), oCnt = take((size_t)H * 4), oFirst = take((size_t)H * 4), oTree = take((size_t)F2 * 16), oSlot = take(nn * 4 + 64);
    if (g) {
        char* b = static_cast<char*>(base);
        g->head = (unsigned long long*)(b + oHead); g->cnt = (uint32_t*)(b + oCnt); g->first_slot = (uint32_t*)(b + oFirst);
        g->tree = (unsigned long long*)(b + oTree); g->slot_h = (uint32_t*)(b + oSlot);
        g->H = H; g->F2cap = F2;
    }
    return off;
}

inline int grid_of(int64_t n, int threads, int cap)
{
    int64_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

size_t flushgroup_scratch_bytes(int32_t n_slots, int32_t n_fl_cap) { return wide_carve(nullptr, nullptr, n_slots, n_fl_cap); }

hipError_t launch_flushgroup_init(int32_t n_slots, int32_t n_fl_cap, void* scratch, hipStream_t stream)
{
    WideScratch g;
    wide_carve(&g, scratch, n_slots, n_fl_cap);
    hipError_t e = hipMemsetAsync(g.head, 0, (size_t)((char*)g.first_slot - (char*)g.head), stream);                      // head, cnt
    if (e == hipSuccess) e = hipMemsetAsync(g.first_slot, 0xFF, (size_t)((char*)g.slot_h - (char*)g.first_slot), stream);   // first_slot, tree
    return e;
}

hipError_t launch_flush_groupby(int32_t n_slots_layout, int32_t n_fl_layout, const im_flush_desc* desc, int32_t n_fl,
                                const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                                const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count,
                                int32_t tie_desc, int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                                void* scratch, hipStream_t stream)
{
    WideArgs A;
    wide_carve(&A.s, scratch, n_slots_layout, n_fl_layout);
    A.desc = desc; A.n_fl = n_fl; A.F2 = pow2_at_least((uint64_t)(n_fl > 0 ? n_fl : 1));
    A.cls = cls; A.b1 = b1; A.b2 = b2; A.consumed = consumed;
    A.cand_rec = cand_rec; A.n_cand = n_cand_dev; A.cand_cap = cand_cap; A.pe_base = pe_base; A.pe_count = pe_count;
    const int64_t work = (int64_t)cand_cap > pe_count ? cand_cap : pe_count;
    const int g = grid_of(work, kWideBlock, 2048);
    hipLaunchKernelGGL(flushw_cut_kernel, dim3(g), dim3(kWideBlock), 0, stream, A, counts);
    hipLaunchKernelGGL(flushw_mark_kernel, dim3(g), dim3(kWideBlock), 0, stream, A);
    hipLaunchKernelGGL(flushw_emit_kernel, dim3(grid_of((int64_t)cand_cap * IM_MAX_EV, kWideBlock, 4096)), dim3(kWideBlock), 0, stream, A, tie_desc,
                       order, cl_key, cl_first, cl_count, reinterpret_cast<unsigned long long*>(counts));
    return hipGetLastError();
}

}  // namespace im
