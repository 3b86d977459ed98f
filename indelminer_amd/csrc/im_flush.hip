// im_flush.hip -- the READCHUNK flushes of process_evidence on gfx950, streaming form.
//
// process_evidence (src/indelminer.c:117-209) sorts the pending evidence by (b1,b2), makes a graph node
// of every entry in front of the first one whose b2 >= marker (140-142, `break`), links split-read nodes
// with identical (class, b1, b2) (add_node, src/graph.c:122-127) and turns components into variants.
// Here that is two steps over evidence SLOT arrays that never leave the device:
//
//   flush cut   (once per flush, two small launches)  the cutting entry X = smallest (b1,b2) among the
//               pending entries with b2 >= marker; every pending entry with (b1,b2) < X is consumed by this
//               flush: consumed[slot] = flush id.  No sort: "in front of the first one" is a comparison
//               with a minimum.  Paired-read evidence takes part (a second slot range the host fills) but
//               is clustered on the host.
//   group-by    (once per batch of contigs, any number of flushes)  one cluster per distinct
//               (flush id, class, b1, b2) over all consumed split-read slots: open-addressing table keyed
//               by a REPRESENTATIVE SLOT (32-bit CAS; keys are compared through the slot arrays, so no
//               128-bit atomics), counts, one-workgroup offset scan over the distinct clusters, placement,
//               and a per-cluster ordering of the members by slot index (= arrival order).
//
// Both are HBM streaming over 16 B per slot (class, b1, b2, consumed); the group-by adds 4 B per slot for the
// table position and 4 B for order[].  The host receives cluster records, not evidence, and orders the
// few clusters of a flush itself (sort_variants, src/variant.c:40-44, is host code in the reference too).

#include "im_device.hpp"

namespace im {
namespace {

__device__ __forceinline__ uint64_t cut_key(int32_t b1, int32_t b2) { return ((uint64_t)(uint32_t)b1 << 32) | (uint32_t)b2; }

struct Ranges {
    int32_t a0, na, b0, nb;
    // range A given as RECORD bounds [a0, a0 + na) of a candidate list that lives on the device: the slots of
    // the candidates whose record index falls inside (null: a0 / na are slot bounds)
    const int32_t* cand_rec; const int32_t* n_cand; int32_t cand_cap;
};
__device__ __forceinline__ int32_t lower_bound_dev(const int32_t* a, int32_t n, int32_t v)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { const int32_t mid = lo + ((hi - lo) >> 1); if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ Ranges resolve(Ranges R)
{
    if (R.cand_rec) {
        const int32_t nc = min(*R.n_cand, R.cand_cap);
        const int32_t lo = lower_bound_dev(R.cand_rec, nc, R.a0), hi = lower_bound_dev(R.cand_rec, nc, R.a0 + R.na);
        R.a0 = lo * IM_MAX_EV; R.na = (hi - lo) * IM_MAX_EV;
    }
    return R;
}
__device__ __forceinline__ int32_t slot_of(const Ranges& R, int64_t i) { return i < R.na ? R.a0 + (int32_t)i : R.b0 + (int32_t)(i - R.na); }

__global__ __launch_bounds__(256) void flush_min_kernel(Ranges R0, const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                       const int32_t* __restrict__ b2, const int32_t* __restrict__ consumed,
                                                       int32_t marker, unsigned long long* __restrict__ cut_word)
{
    const Ranges R = resolve(R0);
    const int64_t n = (int64_t)R.na + R.nb;
    uint64_t best = ~0ull;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = slot_of(R, i);
        if (cls[s] < 0 || consumed[s] != 0) continue;
        const int32_t vb2 = b2[s];
        if (vb2 >= marker) { const uint64_t k = cut_key(b1[s], vb2); if (k < best) best = k; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), o);
        const uint64_t ob = ((uint64_t)hi << 32) | lo;
        if (ob < best) best = ob;
    }
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(cut_word, (unsigned long long)best);
}

__global__ __launch_bounds__(256) void flush_mark_kernel(Ranges R0, const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                        const int32_t* __restrict__ b2, int32_t* __restrict__ consumed,
                                                        int32_t flush_id, const unsigned long long* __restrict__ cut_word)
{
    const Ranges R = resolve(R0);
    const int64_t n = (int64_t)R.na + R.nb;
    const uint64_t X = *cut_word;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = slot_of(R, i);
        if (cls[s] < 0 || consumed[s] != 0) continue;
        if (cut_key(b1[s], b2[s]) < X) consumed[s] = flush_id;
    }
}

// All flushes of a batch in ONE launch: a flush touches only the slots that arrived since the contig began (a few
// thousand at the reference's chunk size), so a single workgroup walks the flush list in order -- the sequence is
// inherently serial (what flush f consumes is not pending for f + 1), and one launch replaces two per flush.
__global__ __launch_bounds__(1024) void flush_seq_kernel(const im_flush_desc* __restrict__ desc, int32_t n_fl,
                                                        const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                        const int32_t* __restrict__ b2, int32_t* __restrict__ consumed,
                                                        const int32_t* __restrict__ cand_rec, const int32_t* __restrict__ n_cand, int32_t cand_cap,
                                                        int32_t pe_base, int32_t pe_count)
{
    __shared__ unsigned long long s_best[16];
    __shared__ int32_t s_first[16];
    __shared__ unsigned long long s_cut;
    __shared__ int32_t s_lo, s_lo_rec0;           // first split-read slot of the current contig that may still be pending
    __shared__ int32_t s_plo, s_plo_pe0, s_pfirst[16];   // the same for the contig's paired-read entries
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // the paired-read entries are pending for the batch's first flush: their marks are cleared here (the split-read
    // marks are cleared where the candidates are written, im_dev_cands.consumed)
    for (int32_t i = t; i < pe_count; i += 1024) consumed[pe_base + i] = 0;
    if (t == 0) { s_lo = 0; s_lo_rec0 = -1; s_plo = 0; s_plo_pe0 = -1; }
    __syncthreads();
    __shared__ int32_t s_bound[1024];             // candidate bounds of up to 512 flushes: [2k] = first, [2k + 1] = end
    __shared__ im_flush_desc s_desc[512];         // ... and their descriptors: a flush starts without a trip to memory
    const int32_t ncand = min(*n_cand, cand_cap);
    for (int32_t f = 0; f < n_fl; f++) {
        if ((f & 511) == 0) {
            // the record bounds of the next 512 flushes -> candidate bounds, one binary search per thread, all at once
            // (a search is a chain of dependent loads: done flush by flush by one lane it is most of the kernel's time)
            __syncthreads();
            const int32_t ff = f + (t >> 1);
            if (ff < n_fl) s_bound[t] = lower_bound_dev(cand_rec, ncand, (t & 1) ? desc[ff].rec1 : desc[ff].rec0);
            if (t < 512 && f + t < n_fl) s_desc[t] = desc[f + t];
            __syncthreads();
        }
        const im_flush_desc D = s_desc[f & 511];
        // every thread works the flush's ranges out for itself from what the last flush left in LDS (no hand-over, no barrier)
        Ranges R;
        {
            const int32_t lo = s_bound[2 * (f & 511)], hi = max(lo, s_bound[2 * (f & 511) + 1]);
            R.a0 = lo * IM_MAX_EV; R.na = (hi - lo) * IM_MAX_EV; R.b0 = pe_base + D.pe0; R.nb = D.pe1 - D.pe0;
            R.cand_rec = nullptr; R.n_cand = nullptr; R.cand_cap = 0;
            if (R.nb < 0) R.nb = 0;
            // Everything in front of s_lo was consumed by earlier flushes of this contig (a flush consumes a prefix of
            // what is pending, in slot order nearly all of it): start there, not at the contig's first slot.
            const int32_t cur_lo = D.rec0 != s_lo_rec0 ? R.a0 : s_lo;
            const int32_t end = R.a0 + R.na;
            if (cur_lo > R.a0) { R.a0 = cur_lo < end ? cur_lo : end; R.na = end - R.a0; }
            // the paired-read entries of the contig likewise: in arrival order nearly all of a flush's are consumed by it
            const int32_t cur_plo = D.pe0 != s_plo_pe0 ? R.b0 : s_plo;
            const int32_t pend = R.b0 + R.nb;
            if (cur_plo > R.b0) { R.b0 = cur_plo < pend ? cur_plo : pend; R.nb = pend - R.b0; }
        }
        const int32_t marker = D.marker, id = D.id;
        // The split-read part by CANDIDATE: a candidate's IM_MAX_EV = 4 slots are 16 contiguous bytes in each of the four
        // arrays, so one thread takes whole candidates with 16-byte loads (a quarter of the load instructions, and the
        // three empty slots of a typical candidate cost nothing more).  One workgroup cannot hide memory latency with
        // occupancy: every thread keeps kC x 4 loads in flight, and when the flush's candidates fit one such sweep (the
        // usual case: ~4 000 candidates per READCHUNK flush) what it loaded stays in registers for the marking pass.
        const int32_t ca = R.a0 >> 2, cb = (R.a0 + R.na + 3) >> 2, nsr = cb - ca;
        const int4* cls4 = reinterpret_cast<const int4*>(cls) + ca;
        const int4* con4 = reinterpret_cast<const int4*>(consumed) + ca;
        const int4* b14 = reinterpret_cast<const int4*>(b1) + ca;
        const int4* b24 = reinterpret_cast<const int4*>(b2) + ca;
        constexpr int kC = 4;
        const bool cached = nsr <= 1024 * kC;
        int4 kc[kC], ku[kC], k1[kC], k2[kC];
#pragma unroll
        for (int e = 0; e < kC; e++) { kc[e] = make_int4(-1, -1, -1, -1); ku[e] = make_int4(1, 1, 1, 1); k1[e] = k2[e] = make_int4(0, 0, 0, 0); }
        uint64_t best = ~0ull;
        auto scan4 = [&](const int4& c, const int4& u, const int4& v1, const int4& v2) {
            if (c.x >= 0 && u.x == 0 && v2.x >= marker) { const uint64_t k = cut_key(v1.x, v2.x); if (k < best) best = k; }
            if (c.y >= 0 && u.y == 0 && v2.y >= marker) { const uint64_t k = cut_key(v1.y, v2.y); if (k < best) best = k; }
            if (c.z >= 0 && u.z == 0 && v2.z >= marker) { const uint64_t k = cut_key(v1.z, v2.z); if (k < best) best = k; }
            if (c.w >= 0 && u.w == 0 && v2.w >= marker) { const uint64_t k = cut_key(v1.w, v2.w); if (k < best) best = k; }
        };
        for (int32_t c0 = t; c0 < nsr; c0 += 1024 * kC) {
#pragma unroll
            for (int e = 0; e < kC; e++) {
                const int32_t c = c0 + 1024 * e;
                const bool in = c < nsr;
                kc[e] = in ? cls4[c] : make_int4(-1, -1, -1, -1);
                ku[e] = in ? con4[c] : make_int4(1, 1, 1, 1);
                k1[e] = in ? b14[c] : make_int4(0, 0, 0, 0);
                k2[e] = in ? b24[c] : make_int4(0, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < kC; e++) scan4(kc[e], ku[e], k1[e], k2[e]);
        }
        constexpr int kP = 4;
        for (int32_t i0 = t; i0 < R.nb; i0 += 1024 * kP) {     // the paired-read entries of the flush: independent loads, kP per array in flight
            int32_t pc[kP], pu[kP], p1[kP], p2[kP];
#pragma unroll
            for (int e = 0; e < kP; e++) {
                const int32_t i = i0 + 1024 * e;
                const bool in = i < R.nb;
                const int32_t sl = R.b0 + (in ? i : 0);
                pc[e] = in ? cls[sl] : -1; pu[e] = in ? consumed[sl] : 1; p2[e] = in ? b2[sl] : 0; p1[e] = in ? b1[sl] : 0;
            }
#pragma unroll
            for (int e = 0; e < kP; e++)
                if (pc[e] >= 0 && pu[e] == 0 && p2[e] >= marker) { const uint64_t k = cut_key(p1[e], p2[e]); if (k < best) best = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)best, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(best >> 32), o);
            const uint64_t ob = ((uint64_t)hi << 32) | lo;
            if (ob < best) best = ob;
        }
        if (lane == 0) s_best[wave] = best;
        __syncthreads();
        if (t == 0) { unsigned long long m = ~0ull; for (int w = 0; w < 16; w++) if (s_best[w] < m) m = s_best[w]; s_cut = m; }
        __syncthreads();
        const uint64_t X = s_cut;
        int32_t first = 0x7fffffff;                 // smallest split-read slot of the range that stays pending
        auto mark4 = [&](int32_t cand, const int4& c, const int4& u, const int4& v1, const int4& v2) {
            const int32_t s0 = (ca + cand) * 4;
            if (c.x >= 0 && u.x == 0) { if (cut_key(v1.x, v2.x) < X) consumed[s0] = id; else if (s0 < first) first = s0; }
            if (c.y >= 0 && u.y == 0) { if (cut_key(v1.y, v2.y) < X) consumed[s0 + 1] = id; else if (s0 + 1 < first) first = s0 + 1; }
            if (c.z >= 0 && u.z == 0) { if (cut_key(v1.z, v2.z) < X) consumed[s0 + 2] = id; else if (s0 + 2 < first) first = s0 + 2; }
            if (c.w >= 0 && u.w == 0) { if (cut_key(v1.w, v2.w) < X) consumed[s0 + 3] = id; else if (s0 + 3 < first) first = s0 + 3; }
        };
        if (cached) {
#pragma unroll
            for (int e = 0; e < kC; e++) mark4(t + 1024 * e, kc[e], ku[e], k1[e], k2[e]);      // out-of-range candidates carry cls = -1
        } else {
            for (int32_t c0 = t; c0 < nsr; c0 += 1024 * kC) {
#pragma unroll
                for (int e = 0; e < kC; e++) {
                    const int32_t c = c0 + 1024 * e;
                    const bool in = c < nsr;
                    kc[e] = in ? cls4[c] : make_int4(-1, -1, -1, -1);
                    ku[e] = in ? con4[c] : make_int4(1, 1, 1, 1);
                    k1[e] = in ? b14[c] : make_int4(0, 0, 0, 0);
                    k2[e] = in ? b24[c] : make_int4(0, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < kC; e++) mark4(c0 + 1024 * e, kc[e], ku[e], k1[e], k2[e]);
            }
        }
        int32_t pfirst = 0x7fffffff;                // smallest paired-read entry of the range that stays pending
        for (int32_t i0 = t; i0 < R.nb; i0 += 1024 * kP) {
            int32_t pc[kP], pu[kP], p1[kP], p2[kP];
#pragma unroll
            for (int e = 0; e < kP; e++) {
                const int32_t i = i0 + 1024 * e;
                const bool in = i < R.nb;
                const int32_t sl = R.b0 + (in ? i : 0);
                pc[e] = in ? cls[sl] : -1; pu[e] = in ? consumed[sl] : 1; p2[e] = in ? b2[sl] : 0; p1[e] = in ? b1[sl] : 0;
            }
#pragma unroll
            for (int e = 0; e < kP; e++) {
                if (pc[e] < 0 || pu[e] != 0) continue;
                const int32_t sl = R.b0 + i0 + 1024 * e;
                if (cut_key(p1[e], p2[e]) < X) consumed[sl] = id;
                else if (sl < pfirst) pfirst = sl;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int32_t of = __shfl_xor(pfirst, o); if (of < pfirst) pfirst = of; }
        if (lane == 0) s_pfirst[wave] = pfirst;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int32_t of = __shfl_xor(first, o); if (of < first) first = of; }
        if (lane == 0) s_first[wave] = first;
        __syncthreads();        // one workgroup = one CU = one vector L1: the marks are visible to the next flush's reads
        if (t == 0) {
            int32_t m = 0x7fffffff;
            for (int w = 0; w < 16; w++) if (s_first[w] < m) m = s_first[w];
            s_lo = m == 0x7fffffff ? R.a0 + R.na : m; s_lo_rec0 = D.rec0;
            int32_t pm = 0x7fffffff;
            for (int w = 0; w < 16; w++) if (s_pfirst[w] < pm) pm = s_pfirst[w];
            s_plo = pm == 0x7fffffff ? R.b0 + R.nb : pm; s_plo_pe0 = D.pe0;
        }
        __syncthreads();
    }
}

// ---- group-by ---------------------------------------------------------------------------------

struct GroupScratch {
    uint32_t* head;         // [H] representative slot + 1, 0 = empty
    uint32_t* cnt;          // [H]
    uint32_t* uid;          // [H] cluster id of the entry
    uint32_t* slot_h;       // [n_slots] table position of a consumed split-read slot, ~0 otherwise
    uint32_t* uniq;         // [n_slots] table positions in first-touch order
    uint32_t* cursor;       // [n_slots] per cluster
    uint32_t* misc;         // [0] distinct clusters while they are being counted (zero between calls), [3] the count for place / finish
    uint32_t  H;
};

__device__ __forceinline__ uint32_t mix32(uint32_t f, uint32_t c, uint32_t x1, uint32_t x2)
{
    uint64_t k = ((uint64_t)x1 << 32) | x2;
    k ^= ((uint64_t)f << 17) | c;
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k;
}

__global__ __launch_bounds__(256) void group_insert_kernel(int32_t n_cap, const int32_t* __restrict__ n_cand, const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                          const int32_t* __restrict__ b2, const int32_t* __restrict__ consumed, GroupScratch s)
{
    const int32_t n_slots = n_cand ? min(n_cap, *n_cand * IM_MAX_EV) : n_cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t c = cls[i], f = consumed[i];
        if (c < 0 || c >= 2 || f <= 0) { s.slot_h[i] = 0xFFFFFFFFu; continue; }
        const int32_t v1 = b1[i], v2 = b2[i];
        uint32_t h = mix32((uint32_t)f, (uint32_t)c, (uint32_t)v1, (uint32_t)v2) & (s.H - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&s.head[h], 0u, (uint32_t)i + 1u);
            if (old == 0u) {
                const uint32_t u = atomicAdd(&s.misc[0], 1u);
                s.uniq[u] = h; s.uid[h] = u;
                break;
            }
            const uint32_t rep = old - 1u;
            if (consumed[rep] == f && cls[rep] == c && b1[rep] == v1 && b2[rep] == v2) break;
            h = (h + 1u) & (s.H - 1);
        }
        atomicAdd(&s.cnt[h], 1u);
        s.slot_h[i] = h;
    }
}

// one workgroup: exclusive scan of the cluster sizes in first-touch order; writes the cluster records
__global__ __launch_bounds__(1024) void group_offsets_kernel(GroupScratch s, const int32_t* __restrict__ cls, const int32_t* __restrict__ b1,
                                                            const int32_t* __restrict__ b2, const int32_t* __restrict__ consumed,
                                                            int32_t* __restrict__ cl_key, int32_t* __restrict__ cl_first, int32_t* __restrict__ cl_count,
                                                            int32_t* __restrict__ counts)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t nu = s.misc[0];
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nu; base += 1024u) {
        const uint32_t u = base + (uint32_t)t;
        uint32_t h = 0, c = 0;
        if (u < nu) { h = s.uniq[u]; c = s.cnt[h]; }
        uint32_t x = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const uint32_t carry = carry_s;
        if (u < nu) {
            const uint32_t first = carry + woff + x - c;
            const uint32_t rep = s.head[h] - 1u;
            cl_first[u] = (int32_t)first; cl_count[u] = (int32_t)c;
            cl_key[4 * (size_t)u + 0] = consumed[rep]; cl_key[4 * (size_t)u + 1] = cls[rep];
            cl_key[4 * (size_t)u + 2] = b1[rep]; cl_key[4 * (size_t)u + 3] = b2[rep];
        }
        __syncthreads();
        if (t == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    // the running count is handed on and cleared for the next call here, by the one workgroup that runs between the
    // insert and the place kernels: no end-of-kernel counter, which would be thousands of atomics on one address
    if (t == 0) { counts[0] = (int32_t)nu; counts[1] = (int32_t)carry_s; s.misc[3] = nu; s.misc[0] = 0u; }
}

__global__ __launch_bounds__(256) void group_place_kernel(int32_t n_cap, const int32_t* __restrict__ n_cand, GroupScratch s, const int32_t* __restrict__ cl_first, int32_t* __restrict__ order)
{
    const int32_t n_slots = n_cand ? min(n_cap, *n_cand * IM_MAX_EV) : n_cap;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_slots; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = s.slot_h[i];
        if (h == 0xFFFFFFFFu) continue;
        const uint32_t u = s.uid[h];
        const uint32_t p = atomicAdd(&s.cursor[u], 1u);
        order[(uint32_t)cl_first[u] + p] = (int32_t)i;
    }
}

// one wave per cluster: members ascending by slot index (descending with tie_desc)
constexpr int kGroupLds = 1024;
__global__ __launch_bounds__(64) void group_finish_kernel(GroupScratch s, const int32_t* __restrict__ cl_first, const int32_t* __restrict__ cl_count,
                                                         int32_t tie_desc, int32_t* __restrict__ order, int32_t* __restrict__ tmp)
{
    __shared__ int32_t st[kGroupLds];
    const int lane = threadIdx.x;
    const uint32_t nu = s.misc[3];
    for (uint32_t u = blockIdx.x; u < nu; u += gridDim.x) {
        const int32_t f = cl_first[u], cnt = cl_count[u];
        // the table cleans itself: the entry, its count and the cluster's cursor are zero again for the next call
        if (lane == 0) { const uint32_t h = s.uniq[u]; s.head[h] = 0u; s.cnt[h] = 0u; s.cursor[u] = 0u; }
        if (cnt <= 1) continue;
        if (cnt <= kGroupLds) {
            for (int32_t t = lane; t < cnt; t += 64) st[t] = order[f + t];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int32_t t = lane; t < cnt; t += 64) {
                const int32_t v = st[t];
                int32_t r = 0;
                for (int32_t j = 0; j < cnt; j++) r += (st[j] < v) ? 1 : 0;
                order[f + (tie_desc ? (cnt - 1 - r) : r)] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        } else {
            // rare (a breakpoint with more than 1024 supporting reads): rank by counting through global memory
            for (int32_t t = lane; t < cnt; t += 64) tmp[f + t] = order[f + t];
            __threadfence();
            __builtin_amdgcn_wave_barrier();
            for (int32_t t = lane; t < cnt; t += 64) {
                const int32_t v = tmp[f + t];
                int32_t r = 0;
                for (int32_t j = 0; j < cnt; j++) r += (tmp[f + j] < v) ? 1 : 0;
                order[f + (tie_desc ? (cnt - 1 - r) : r)] = v;
            }
        }
    }
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }
inline uint32_t group_table_size(int32_t n_slots)
{
    uint32_t H = 1024;
    while (H < 2u * (uint32_t)(n_slots > 0 ? n_slots : 1)) H <<= 1;
    return H;
}
inline size_t group_carve(GroupScratch* g, int32_t** tmp, void* base, int32_t n_slots)
{
    const uint32_t H = group_table_size(n_slots);
    const size_t nn = (size_t)(n_slots > 0 ? n_slots : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = up256(off + bytes); return o; };
    // head, cnt, cursor, misc are contiguous: one memset clears them
    const size_t oHead = take((size_t)H * 4), oCnt = take((size_t)H * 4), oCur = take(nn * 4), oMisc = take(64);
    const size_t oUid = take((size_t)H * 4), oSlot = take(nn * 4), oUniq = take(nn * 4), oTmp = take(nn * 4);
    if (g) {
        char* b = static_cast<char*>(base);
        g->head = (uint32_t*)(b + oHead); g->cnt = (uint32_t*)(b + oCnt); g->cursor = (uint32_t*)(b + oCur); g->misc = (uint32_t*)(b + oMisc);
        g->uid = (uint32_t*)(b + oUid); g->slot_h = (uint32_t*)(b + oSlot); g->uniq = (uint32_t*)(b + oUniq);
        g->H = H;
        *tmp = (int32_t*)(b + oTmp);
    }
    (void)oMisc;
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

hipError_t launch_flush_cut(const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                            int32_t a0, int32_t a1, int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id,
                            uint64_t* cut_word, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, hipStream_t stream)
{
    Ranges R;
    R.a0 = a0; R.na = a1 > a0 ? a1 - a0 : 0; R.b0 = b0; R.nb = b1_end > b0 ? b1_end - b0 : 0;
    R.cand_rec = cand_rec; R.n_cand = n_cand_dev; R.cand_cap = cand_cap;
    // with record bounds the slot count is only known on the device: size the launch for the most it can be
    int64_t n = (int64_t)R.na + R.nb;
    if (cand_rec) { const int64_t most = (int64_t)R.na < cand_cap ? R.na : cand_cap; n = most * IM_MAX_EV + R.nb; }
    if (n <= 0) return hipSuccess;
    const int g = grid_of(n, 256, 2048);
    hipLaunchKernelGGL(flush_min_kernel, dim3(g), dim3(256), 0, stream, R, cls, b1, b2, consumed, marker,
                       reinterpret_cast<unsigned long long*>(cut_word));
    hipLaunchKernelGGL(flush_mark_kernel, dim3(g), dim3(256), 0, stream, R, cls, b1, b2, consumed, flush_id,
                       reinterpret_cast<const unsigned long long*>(cut_word));
    return hipGetLastError();
}

hipError_t launch_flush_seq(const im_flush_desc* desc, int32_t n_fl, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                            int32_t* consumed, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base,
                            int32_t pe_count, hipStream_t stream)
{
    if (n_fl <= 0) return hipSuccess;
    hipLaunchKernelGGL(flush_seq_kernel, dim3(1), dim3(1024), 0, stream, desc, n_fl, cls, b1, b2, consumed, cand_rec, n_cand_dev, cand_cap, pe_base, pe_count);
    return hipGetLastError();
}

size_t groupby_scratch_bytes(int32_t n_slots) { return group_carve(nullptr, nullptr, nullptr, n_slots); }

hipError_t launch_groupby_init(int32_t n_slots, void* scratch, hipStream_t stream)
{
    GroupScratch g; int32_t* tmp = nullptr;
    group_carve(&g, &tmp, scratch, n_slots);
    return hipMemsetAsync(g.head, 0, (size_t)((char*)g.misc - (char*)g.head) + 64, stream);
}

hipError_t launch_groupby(int32_t n_layout, int32_t n_slots, const int32_t* n_cand_dev, const int32_t* cls, const int32_t* b1, const int32_t* b2, const int32_t* consumed,
                          int32_t tie_desc, int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count,
                          int32_t* counts, void* scratch, hipStream_t stream)
{
    GroupScratch g; int32_t* tmp = nullptr;
    // The scratch is carved for the slot count it was INITIALISED for (n_layout), whatever this call's n_slots: the table is
    // zeroed once by launch_groupby_init and every call leaves its own entries clean, which only holds while every call
    // sees the same arrays at the same places (a layout per call put one call's counters inside another's slot lists:
    // phantom clusters, out-of-range representatives -- found on the 24-contig input, groups of very different sizes).
    group_carve(&g, &tmp, scratch, n_layout);
    const int gi = grid_of(n_slots, 256, 4096);
    hipLaunchKernelGGL(group_insert_kernel, dim3(gi), dim3(256), 0, stream, n_slots, n_cand_dev, cls, b1, b2, consumed, g);
    hipLaunchKernelGGL(group_offsets_kernel, dim3(1), dim3(1024), 0, stream, g, cls, b1, b2, consumed, cl_key, cl_first, cl_count, counts);
    hipLaunchKernelGGL(group_place_kernel, dim3(gi), dim3(256), 0, stream, n_slots, n_cand_dev, g, cl_first, order);
    hipLaunchKernelGGL(group_finish_kernel, dim3(grid_of(n_slots, 16, 4096)), dim3(64), 0, stream, g, cl_first, cl_count, tie_desc, order, tmp);
    return hipGetLastError();
}

}  // namespace im
