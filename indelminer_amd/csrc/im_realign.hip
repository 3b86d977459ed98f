// im_realign.hip -- split-read realignment on gfx950: one wavefront per read.
//
// Replaces attempt_pe_alignment (src/alignment.c:764-799) and everything below
// it for numgaps == 0 (the reference default, src/indelminer.c:934), where the
// banded Smith-Waterman degenerates to a scan of one diagonal:
//
//   K1  find_best_band   src/alignment.c:393-447   k-mer vote over diagonals
//   K2  local_align      src/localalign.c:15-196   max-scoring segment on the band
//   K3  ALIGN/fetch_cigar src/globalalign.c:333-401,507-604  all-REP script -> =/X runs
//   K4  find_best_del_candidate / count_matches  src/alignment.c:219-339
//   a10 update_readsegs  src/readaln.c:348-458     final segment list
//   a11 new_evidence     src/evidence.c:4-34       evidence record
//
// Work shape.  A workgroup is ONE wavefront (64 lanes) and owns one read at a
// time; the grid is a few workgroups per CU that stride over the batch, with
// the blockIdx -> read mapping arranged so that the workgroups of one XCD
// (blockIdx % 8) walk neighbouring reads and share reference windows in that
// XCD's L2.  Per-wave LDS: a 4 KiB read k-mer table, a 4 KiB packed-byte
// diagonal histogram, the read and two match-flag strips.  No MFMA: this is
// integer scan / histogram work bound by LDS atomics and latency.
//
// Lane layout for everything positional: lane l owns read positions
// 4l..4l+3 (hence IM_MAX_READ = 255); prefix sums / minima run as lane-local
// 4-step chains plus one 64-lane scan.

#include "im_device.hpp"

// Diagnostic build only (-DIM_STAMPS, `python indelminer_amd/build.py --stamps`): per-phase
// shader-cycle sums over all waves, read back through im_debug_stamps_read.  The product
// library is built without it and contains no stamp.
#ifdef IM_STAMPS
__device__ unsigned long long im_stamp_acc[32];
#define IM_STAMP_DECL unsigned long long stamp_prev_ = __builtin_readcyclecounter();
#define IM_STAMP(id) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
                          if (threadIdx.x == 0) atomicAdd(&im_stamp_acc[id], t_ - stamp_prev_); \
                          stamp_prev_ = __builtin_readcyclecounter(); } while (0)
#define IM_STAMP_ARG , unsigned long long& stamp_prev_, int stamp_base_
#define IM_STAMP_PASS(base) , stamp_prev_, base
#define IM_STAMP_B(id) IM_STAMP(stamp_base_ + (id))
#else
#define IM_STAMP_DECL
#define IM_STAMP(id)
#define IM_STAMP_ARG
#define IM_STAMP_PASS(base)
#define IM_STAMP_B(id)
#endif

namespace im {
namespace {

constexpr int kDiagChunk = 4096;            // diagonals per histogram pass (1 byte each)
constexpr int kTblBytes  = 4096;            // 4^6 direct table, or 512-slot hash (keys+vals)
constexpr int kHashSlots = 512;
constexpr int kDirectMaxK = 6;

constexpr int kScoreMatch = 1;              // src/localalign.c:10-13
constexpr int kScoreMismatch = -10;

struct WaveLds {
    uint32_t diag[kDiagChunk / 4 + 16];     // packed byte counters (+ slack for band sums)
    uint32_t tbl[kTblBytes / 4];
    uint32_t rd[(256 + 16) / 4];            // read bases, read coordinates
    uint32_t eq[2][256 / 4];                // per band alignment: match flag per read position
};

// ---- wave helpers (64 lanes) ------------------------------------------------
//
// Scans and reductions run on the DPP cross-lane path (row_shr / row_bcast), six
// VALU steps and no LDS round trip, instead of ds_bpermute shuffles.  Reductions
// end in v_readlane, so their results live in SGPRs and the control flow that
// hangs off them is scalar.

constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// A workgroup is one wavefront, so no s_barrier is ever needed: LDS executes one wave's
// instructions in order, a later read sees an earlier write of any lane.  What is needed is
// only that the compiler keeps the order -- and, unlike __syncthreads(), does NOT drain the
// vector-memory queue (s_waitcnt vmcnt(0)), which is what lets reference-window loads issued
// before a phase overlap that phase.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ int wave_scan_add(int v, int /*lane*/)
{
    v += dpp_mov<kDppRowShr1>(0, v);
    v += dpp_mov<kDppRowShr2>(0, v);
    v += dpp_mov<kDppRowShr4>(0, v);
    v += dpp_mov<kDppRowShr8>(0, v);
    v += dpp_mov<kDppBcast15, 0xa>(0, v);
    v += dpp_mov<kDppBcast31, 0xc>(0, v);
    return v;                                // inclusive
}
__device__ __forceinline__ int wave_scan_min_incl(int v)
{
    v = min(v, dpp_mov<kDppRowShr1>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr2>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr4>(INT_MAX, v));
    v = min(v, dpp_mov<kDppRowShr8>(INT_MAX, v));
    v = min(v, dpp_mov<kDppBcast15, 0xa>(INT_MAX, v));
    v = min(v, dpp_mov<kDppBcast31, 0xc>(INT_MAX, v));
    return v;
}
__device__ __forceinline__ int wave_scan_max_incl(int v)
{
    v = max(v, dpp_mov<kDppRowShr1>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr2>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr4>(INT_MIN, v));
    v = max(v, dpp_mov<kDppRowShr8>(INT_MIN, v));
    v = max(v, dpp_mov<kDppBcast15, 0xa>(INT_MIN, v));
    v = max(v, dpp_mov<kDppBcast31, 0xc>(INT_MIN, v));
    return v;
}
__device__ __forceinline__ int wave_scan_min_excl(int v, int /*lane*/)
{
    // exclusive running minimum, identity INT_MAX
    return dpp_mov<kDppWaveShr1>(INT_MAX, wave_scan_min_incl(v));
}
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_scan_max_incl(v), 63); }
__device__ __forceinline__ int wave_min(int v) { return __builtin_amdgcn_readlane(wave_scan_min_incl(v), 63); }
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_add(v, 0), 63); }

// base2bits, src/alignment.c:11-24
__device__ __forceinline__ uint32_t code2(uint32_t c)
{
    const uint32_t u = (c | 0x20u) - 'a';
    uint32_t x = (c >> 1) & 3u;
    x ^= x >> 1;
    const bool valid = (u < 26u) && ((0x80045u >> u) & 1u);
    return valid ? x : 0u;
}

__device__ __forceinline__ uint32_t lds_byte(const uint32_t* base, uint32_t i)
{
    return reinterpret_cast<const uint8_t*>(base)[i];
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// ---- K1: band search ---------------------------------------------------------

struct Band {
    int st;             // 0 ok, IM_ST_ABORT
    int low;            // diagonal handed to the scan
    int votes;
    int win, piece;
};

// the read k-mer table: value 0 = k-mer absent from the read piece, 0xFF = occurs
// more than once (bin_diagonals only lets read-unique k-mers vote, 97-98), else
// 1 + offset of the k-mer in the piece.
template <int KT, bool DIRECT>
__device__ __forceinline__ void table_build(WaveLds& s, uint32_t p0, uint32_t nq, uint32_t k, int lane, uint32_t read_pk8)
{
    if constexpr (KT == 6) {
        // k = 6 (the reference default): the read's 2-bit codes travel as one packed byte per lane
        // (bases 4l..4l+3); the two following lanes' bytes come over DPP, and the lane's four
        // 6-mers are bit fields of that 24-bit window.  The table is kept clean by un-doing the
        // entries after the vote (table_undo), so no 4 KiB clear per band search.
        uint8_t* t8 = reinterpret_cast<uint8_t*>(s.tbl);
        const uint32_t n1 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)read_pk8);
        const uint32_t n2 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n1);
        const uint32_t w24 = (read_pk8 << 16) | (n1 << 8) | n2;
        uint32_t code[4]; bool have[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = 4u * lane + j;
            have[j] = x >= p0 && x < p0 + nq;
            code[j] = (w24 >> (12 - 2 * j)) & 0xFFFu;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) if (have[j]) t8[code[j]] = (uint8_t)(4u * lane + j - p0 + 1u);
        wave_lds_sync();
        bool lost[4];
#pragma unroll
        for (int j = 0; j < 4; j++) lost[j] = have[j] && (t8[code[j]] != (uint8_t)(4u * lane + j - p0 + 1u));
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) if (lost[j]) t8[code[j]] = 0xFFu;
        wave_lds_sync();
        return;
    }
    uint8_t* t8 = reinterpret_cast<uint8_t*>(s.tbl);
    // clear
    if (DIRECT) {
#pragma unroll
        for (int i = 0; i < kTblBytes / 4 / 64; i++) s.tbl[lane + 64 * i] = 0u;
    } else {
#pragma unroll
        for (int i = 0; i < kHashSlots / 64; i++) {
            s.tbl[lane + 64 * i] = 0xFFFFFFFFu;                 // keys
            s.tbl[kHashSlots + lane + 64 * i] = 0u;             // vals
        }
    }
    wave_lds_sync();
    const uint32_t mask = (1u << (2 * k)) - 1u;                  // k <= 15
    uint32_t code[4];
    bool have[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t q = 4u * lane + j;
        have[j] = q < nq;
        uint32_t c = 0;
        if (have[j])
            for (uint32_t u = 0; u < k; u++) c = (c << 2) | code2(lds_byte(s.rd, p0 + q + u));
        code[j] = c & mask;
    }
    if (DIRECT) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (have[j]) t8[code[j]] = (uint8_t)(4u * lane + j + 1u);
        wave_lds_sync();
        bool lost[4];
#pragma unroll
        for (int j = 0; j < 4; j++) lost[j] = have[j] && (t8[code[j]] != (uint8_t)(4u * lane + j + 1u));
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) if (lost[j]) t8[code[j]] = 0xFFu;
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!have[j]) continue;
            uint32_t h = (code[j] * 2654435761u) >> 23;          // 9 bits
            for (int probe = 0; probe < kHashSlots; probe++) {
                const uint32_t old = atomicCAS(&s.tbl[h], 0xFFFFFFFFu, code[j]);
                if (old == 0xFFFFFFFFu) { atomicMax(&s.tbl[kHashSlots + h], 4u * lane + j + 1u); break; }
                if (old == code[j])     { atomicMax(&s.tbl[kHashSlots + h], 0xFFu); break; }
                h = (h + 1) & (kHashSlots - 1);
            }
        }
    }
    wave_lds_sync();
}

// un-does table_build<6>: every lane zeroes the entries of its own k-mers
__device__ __forceinline__ void table_undo6(WaveLds& s, uint32_t p0, uint32_t nq, int lane, uint32_t read_pk8)
{
    uint8_t* t8 = reinterpret_cast<uint8_t*>(s.tbl);
    const uint32_t n1 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)read_pk8);
    const uint32_t n2 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n1);
    const uint32_t w24 = (read_pk8 << 16) | (n1 << 8) | n2;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t x = 4u * lane + j;
        if (x >= p0 && x < p0 + nq) t8[(w24 >> (12 - 2 * j)) & 0xFFFu] = 0;
    }
}

template <bool DIRECT>
__device__ __forceinline__ uint32_t table_lookup(const WaveLds& s, uint32_t code)
{
    if (DIRECT) return lds_byte(s.tbl, code);
    uint32_t h = (code * 2654435761u) >> 23;
    for (int probe = 0; probe < kHashSlots; probe++) {
        const uint32_t key = s.tbl[h];
        if (key == code) return s.tbl[kHashSlots + h];
        if (key == 0xFFFFFFFFu) return 0u;
        h = (h + 1) & (kHashSlots - 1);
    }
    return 0u;
}

// find_best_band (src/alignment.c:393-447): read_seeds x2 (29-68), bin_diagonals
// (70-128), bin_bands (130-140), select_band (142-181).  Window = contig[w0,w1),
// read piece = read[p0,p1), anchor in contig coordinates.
template <int KT, bool DIRECT>
__device__ __forceinline__ Band band_search(WaveLds& s, const uint64_t* __restrict__ pk,
                            uint32_t w0, uint32_t w1, uint32_t anchor,
                            uint32_t p0, uint32_t p1, uint32_t k, uint32_t g, int lane, uint32_t read_pk8 IM_STAMP_ARG)
{
    Band b;
    const uint32_t W = w1 - w0, Lp = p1 - p0;
    const uint32_t numdiag = (W - (k - 1)) + (Lp - (k - 1));     // unsigned, as written (403-404)
    b.win = (int)W; b.piece = (int)Lp; b.votes = 0; b.low = 0; b.st = 0;
    if (!(numdiag > g) || p1 < p0) { b.st = IM_ST_ABORT; return b; }   // forceasserts 405, 407
    if (Lp < k) { b.low = (int)(numdiag - 1); return b; }         // 408-412
    if ((int32_t)numdiag <= 0) { b.st = IM_ST_ABORT; return b; }  // reference would run off its arrays

    const uint32_t nq = Lp - k + 1;                               // k-mers in the read piece
    const uint32_t npos = (W >= k) ? (W - k + 1) : 0u;            // k-mer starts in the window
    const uint32_t kmask = (1u << (2 * k)) - 1u;
    const int anchor_rel = (int)(anchor - w0);                    // select_band gets it as int (431,146)
    const uint32_t step = kDiagChunk - g;

    // Window words of one histogram chunk: every lane takes 16 positions per sweep of 1024.
    // All sweeps of a chunk are requested up front (kPre of them into registers) so that the
    // chunk pays one memory round trip, and chunk 0 is requested BEFORE the read table is
    // built so that the trip overlaps that work (kPre = 3 sweeps live in registers).
    constexpr int kPre = 3;
    int bc = 0, bd = INT_MAX, bi = 0;                             // select_band's max, dist, indx

    for (uint32_t c0 = 0; c0 < numdiag; c0 += step) {
        const int p_lo = max(0, (int)c0 - (int)nq);
        const int p_hi = min((int)npos - 1, (int)(c0 + kDiagChunk) - 2);
        const uint32_t g0 = (w0 + (uint32_t)(p_lo <= p_hi ? p_lo : 0)) & ~15u;
        const uint32_t a_hi = w0 + (uint32_t)(p_lo <= p_hi ? p_hi : 0);
        uint64_t whi0 = 0, wlo0 = 0, whi1 = 0, wlo1 = 0, whi2 = 0, wlo2 = 0;
        if (p_lo <= p_hi) {
            const uint32_t A0 = g0 + 16u * lane, A1 = A0 + 1024u, A2 = A0 + 2048u;
            if (A0 <= a_hi) { whi0 = pk[A0 >> 5]; wlo0 = pk[(A0 >> 5) + 1]; }
            if (A1 <= a_hi) { whi1 = pk[A1 >> 5]; wlo1 = pk[(A1 >> 5) + 1]; }
            if (A2 <= a_hi) { whi2 = pk[A2 >> 5]; wlo2 = pk[(A2 >> 5) + 1]; }
        }
        if (c0 == 0) {
            table_build<KT, DIRECT>(s, p0, nq, k, lane, read_pk8);
            IM_STAMP_B(0);
        }
        // clear the part of the histogram this chunk can touch
        {
            const uint32_t nbytes = min(numdiag - c0, (uint32_t)kDiagChunk) + (g ? 64u : 0u);
            for (uint32_t w = lane; 4u * w < nbytes; w += 64) s.diag[w] = 0u;
        }
        wave_lds_sync();
        IM_STAMP_B(1);

        // vote: window k-mer at p and read-unique k-mer at q land on diagonal p - q + nq (102-105)
        if (p_lo <= p_hi) {
            int it = 0;
            for (uint32_t A = g0 + 16u * lane; A <= a_hi; A += 1024u, it++) {
                uint64_t hi, lo;
                if (it == 0) { hi = whi0; lo = wlo0; }
                else if (it == 1) { hi = whi1; lo = wlo1; }
                else if (it == 2) { hi = whi2; lo = wlo2; }
                else { hi = pk[A >> 5]; lo = pk[(A >> 5) + 1]; }
                const uint64_t pkd = (A & 16u) ? ((hi << 32) | (lo >> 32)) : hi;  // bases A..A+31, first base on top
                if (DIRECT) {
                    // all sixteen table reads first, then the (rare) hits
                    uint32_t v[16];
                    if constexpr (KT == 6) {
                        const uint32_t u = (uint32_t)(pkd >> 32), l = (uint32_t)pkd;   // bases 0-15 | 16-31
#pragma unroll
                        for (int j = 0; j < 16; j++) {
                            const uint32_t code = (j <= 10) ? ((u >> (20 - 2 * j)) & 0xFFFu)
                                                            : (__builtin_amdgcn_alignbit(u, l, 52 - 2 * j) & 0xFFFu);
                            v[j] = lds_byte(s.tbl, code);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; j++)
                            v[j] = lds_byte(s.tbl, (uint32_t)(pkd >> (2 * (32 - j - (int)k))) & kmask);
                    }
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int p = (int)(A + j - w0);
                        if (v[j] - 1u >= 0xFEu || p < p_lo || p > p_hi) continue;
                        const uint32_t off = (uint32_t)p - (v[j] - 1u) + nq - c0;    // diagonal index - c0
                        if (off < (uint32_t)kDiagChunk)
                            atomicAdd(&s.diag[off >> 2], 1u << ((off & 3u) * 8u));
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const int p = (int)(A + j - w0);
                        if (p < p_lo || p > p_hi) continue;
                        const uint32_t code = (uint32_t)(pkd >> (2 * (32 - j - (int)k))) & kmask;
                        const uint32_t v = table_lookup<DIRECT>(s, code);
                        if (v == 0u || v == 0xFFu) continue;
                        const uint32_t off = (uint32_t)p - (v - 1u) + nq - c0;       // diagonal index - c0
                        if (off < (uint32_t)kDiagChunk)
                            atomicAdd(&s.diag[off >> 2], 1u << ((off & 3u) * 8u));
                    }
                }
            }
        }
        wave_lds_sync();
        IM_STAMP_B(2);

        // bin_bands + select_band over i in [c0, iend)
        const uint32_t iend = min(c0 + step, numdiag);
        const uint32_t nband = numdiag - g;                       // bands[i] == 0 for i >= nband (135)
        for (uint32_t dw = lane; 4u * dw < iend - c0; dw += 64) {
            const uint32_t v = s.diag[dw];
            if (g == 0 && v == 0u && bc > 0) continue;            // four empty diagonals cannot beat a vote
#pragma unroll
            for (int bb = 0; bb < 4; bb++) {
                const uint32_t i = c0 + 4u * dw + bb;
                if (i >= iend) break;
                int cnt = (int)((v >> (8 * bb)) & 255u);
                if (g > 0) {                                      // bin_bands: sum of g+1 neighbouring diagonals
                    if (i < nband) for (uint32_t e = 1; e <= g; e++) cnt += (int)lds_byte(s.diag, i - c0 + e);
                    else cnt = 0;
                }
                const int d = abs(anchor_rel - (int)i);
                if (cnt > bc || (cnt == bc && (d < bd || (d == bd && (int)i < bi)))) { bc = cnt; bd = d; bi = (int)i; }
            }
        }
        wave_lds_sync();
        IM_STAMP_B(3);
    }
    if constexpr (KT == 6) table_undo6(s, p0, nq, lane, read_pk8);
    // wave argmax with select_band's order: most votes, then nearest the anchor, then smallest index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(bc, o), od = __shfl_xor(bd, o), oi = __shfl_xor(bi, o);
        if (oc > bc || (oc == bc && (od < bd || (od == bd && oi < bi)))) { bc = oc; bd = od; bi = oi; }
    }
    b.votes = bc;
    b.low = bi - (int)nq;                                          // 438
    IM_STAMP_B(4);
    return b;
}

// ---- K2/K3: the diagonal scan ------------------------------------------------

struct Aln {
    int st;                 // 0 ok, IM_ST_ABORT
    int q1, q2, r1, r2;     // 0-based half-open read / contig coordinates; q1 == q2: no alignment
    int f, l;               // leading / trailing '=' run (src/alignment.c:585-599)
};

// local_align + ALIGN + fetch_cigar for low == up == d (src/localalign.c:100-176
// with band == 1; closed form validated in SURVEY.md A.5a):
//   forward : c_t = max(0, c_{t-1} + w_t); end = first t where c_t is the strict maximum
//   reverse : start = largest s <= end with sum_{s..end} w == best
// Match flags of the aligned positions go to eqdst[] in read coordinates.
__device__ __forceinline__ Aln diag_scan(WaveLds& s, const uint8_t* __restrict__ contig,
                         uint32_t w0, uint32_t w1, uint32_t p0, uint32_t p1, int d,
                         uint32_t* eqdst, int lane)
{
    Aln a;
    a.st = 0; a.q1 = a.q2 = a.r1 = a.r2 = 0; a.f = a.l = 0;
    const int M = (int)(p1 - p0), N = (int)(w1 - w0);
    if (M <= 0 || N <= 0 || d < -M || d > N) { a.st = IM_ST_ABORT; return a; }   // src/localalign.c:31-32,70-77
    const int t_lo = max(0, -d), t_hi = min(M, N - d);

    eqdst[lane] = 0u;
    const int t0 = 4 * lane;
    uint32_t rdw = 0, rfw = 0;
    if (t0 < t_hi && t0 + 3 >= t_lo) {
        const uint8_t* rb = reinterpret_cast<const uint8_t*>(s.rd) + p0 + t0;
        rdw = (uint32_t)rb[0] | ((uint32_t)rb[1] << 8) | ((uint32_t)rb[2] << 16) | ((uint32_t)rb[3] << 24);
        rfw = load_u32_unaligned(contig + ((int64_t)w0 + d + t0));
    }
    int w[4]; bool eq[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = t0 + j;
        const bool valid = t >= t_lo && t < t_hi;
        eq[j] = valid && (((rdw >> (8 * j)) & 255u) == ((rfw >> (8 * j)) & 255u));
        w[j] = valid ? (eq[j] ? kScoreMatch : kScoreMismatch) : 0;
    }
    // inclusive prefix sums S_t
    int S[4];
    S[0] = w[0]; S[1] = S[0] + w[1]; S[2] = S[1] + w[2]; S[3] = S[2] + w[3];
    const int incl = wave_scan_add(S[3], lane);
    const int excl = incl - S[3];
#pragma unroll
    for (int j = 0; j < 4; j++) S[j] += excl;
    // running minimum of S including the empty prefix (0)
    int m[4];
    m[0] = S[0]; m[1] = min(m[0], S[1]); m[2] = min(m[1], S[2]); m[3] = min(m[2], S[3]);
    const int pm = min(0, wave_scan_min_excl(m[3], lane));
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = S[j] - min(pm, m[j]);
    const int best = wave_max(max(max(c[0], c[1]), max(c[2], c[3])));
    if (best <= 0) return a;                                       // score <= 0 (src/alignment.c:365-372)
    int e_loc = INT_MAX;
#pragma unroll
    for (int j = 3; j >= 0; j--) if (c[j] == best) e_loc = t0 + j;
    const int end = wave_min(e_loc);
    const int el = end >> 2, ej = end & 3;
    int s_sel = (ej == 0) ? S[0] : (ej == 1) ? S[1] : (ej == 2) ? S[2] : S[3];
    const int Send = __builtin_amdgcn_readlane(s_sel, el);
    const int target = Send - best;
    int st_loc = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = t0 + j;
        const int sprev = (j == 0) ? excl : S[j - 1];
        if (t >= t_lo && t <= end && sprev == target) st_loc = t;
    }
    const int start = wave_max(st_loc);
    if (start < 0 || end == start) return a;                       // single cell: score 0 (src/localalign.c:191-193)

    // '='/X flags of the aligned span, and the leading/trailing '=' runs
    uint32_t flags = 0;
    int fm = INT_MAX, lm = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int t = t0 + j;
        if (t >= start && t <= end) {
            if (eq[j]) flags |= 1u << (8 * j);
            else { fm = min(fm, t); lm = max(lm, t); }
        }
    }
    fm = wave_min(fm); lm = wave_max(lm);
    a.f = (fm == INT_MAX ? end + 1 : fm) - start;
    a.l = end - (lm < 0 ? start - 1 : lm);
    // scatter flags to read coordinates p0 + t
    {
        uint8_t* e8 = reinterpret_cast<uint8_t*>(eqdst);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = (int)p0 + t0 + j;
            if (x < 256 && ((flags >> (8 * j)) & 1u)) e8[x] = 1;
        }
        wave_lds_sync();
    }
    a.q1 = (int)p0 + start;                                         // src/alignment.c:385-388
    a.q2 = (int)p0 + end + 1;
    a.r1 = (int)w0 + d + start;
    a.r2 = (int)w0 + d + end + 1;
    return a;
}

// ---- the per-read driver -----------------------------------------------------

__device__ __forceinline__ void store_band(im_read_result* out, int which, const Band& b, const Aln& a, int lane)
{
    if (lane == 0) {
        im_band_aln* o = &out->band[which];
        o->r1 = a.r1; o->r2 = a.r2; o->q1 = a.q1; o->q2 = a.q2;
        o->low = b.low; o->votes = b.votes; o->win_bytes = b.win; o->piece_bytes = b.piece;
    }
}

__device__ __forceinline__ void finish(im_read_result* out, int status, int n_band, int lane)
{
    if (lane == 0) { out->status = status; out->n_band = n_band; if (status != IM_ST_EVIDENCE) { out->n_ev = 0; out->n_ops = 0; out->ref_start = 0; } }
}

// evidence slots of read c (see im_dev_batch): slot k < n_ev live, the rest empty
__device__ __forceinline__ void write_slots(const RealignArgs& A, int c, int n_ev, int cls0, int b1, int b2, int lane)
{
    if (A.batch.ev_cls && lane < IM_MAX_EV) {
        const int64_t sl = (int64_t)c * IM_MAX_EV + lane;
        const bool live = lane < n_ev;
        A.batch.ev_cls[sl] = live ? cls0 : -1;
        A.batch.ev_b1[sl] = live ? b1 : 0;
        A.batch.ev_b2[sl] = live ? b2 : 0;
    }
}

// attempt_pe_alignment -> attempt_diagonal_alignments (src/alignment.c:539-799)
template <int KT, bool DIRECT>
__device__ __forceinline__ void realign_one(WaveLds& s, const RealignArgs& A, int c, int lane)
{
    IM_STAMP_DECL
    im_read_result* out = &A.batch.out[c];
    // Every per-read scalar is the same in all 64 lanes.  Saying so (v_readfirstlane) puts the
    // values in SGPRs and turns the control flow below into scalar branches instead of exec-mask
    // bookkeeping -- the compiler cannot prove uniformity of loaded values on its own.
    const int64_t off = uni64(A.batch.base_off[c]);
    const int64_t Lraw = uni(A.batch.read_len[c]);
    const int tid = uni(A.batch.tid[c]);
    const int anchor = uni(A.batch.anchor[c]);
    const int R = uni(A.batch.range_max[c]);
    const uint32_t k = KT ? (uint32_t)KT : A.P.klength, g = KT ? 0u : A.P.numgaps, eth = A.P.ethreshold;

    if (lane < 16) reinterpret_cast<uint32_t*>(&out->band[0])[lane] = 0u;
    if (lane < 7) out->reserved[lane] = 0;
    write_slots(A, c, 0, -1, 0, 0, lane);
    if (Lraw <= 0 || Lraw > IM_MAX_READ || tid < 0 || tid >= A.ref.n_contigs || (off & 3)) {
        finish(out, (Lraw > IM_MAX_READ || (off & 3)) ? IM_ST_UNSUPPORTED : IM_ST_ABORT, 0, lane);
        return;
    }
    const int L = (int)Lraw;
    const uint8_t* contig = A.ref.ascii + uni64(A.ref.asc_off[tid]);
    const uint64_t* pk = A.ref.pk + uni64(A.ref.pk_off[tid]);
    const int clen = uni(A.ref.len[tid]);

    // stage the read
    uint32_t read_pk8 = 0;      // 2-bit codes of this lane's four bases, first base on top
    {
        uint32_t v = 0;
        if (4 * lane < L) v = *reinterpret_cast<const uint32_t*>(A.batch.bases + off + 4 * lane);
        const int rem = L - 4 * lane;                              // zero the bytes past the read
        if (rem < 4) v &= (rem <= 0) ? 0u : ((1u << (8 * rem)) - 1u);
        s.rd[lane] = v;
        if (lane < 4) s.rd[64 + lane] = 0u;
        read_pk8 = (code2(v & 255u) << 6) | (code2((v >> 8) & 255u) << 4) | (code2((v >> 16) & 255u) << 2) | code2(v >> 24);
    }
    wave_lds_sync();

    // window geometry (src/alignment.c:774-783)
    int distance = R;
    const int left1  = anchor >= distance ? anchor - distance : 0;
    const int right1 = clen < (anchor + distance) ? clen : anchor + distance;
    distance = R + (int)A.P.maxdelsize;
    const int left2  = anchor >= distance ? anchor - distance : 0;
    const int right2 = clen < (anchor + distance) ? clen : anchor + distance;
    if (!(anchor >= left1 && anchor >= left2 && anchor <= right1 && anchor <= right2 &&
          left2 >= 0 && right2 > 0)) { finish(out, IM_ST_ABORT, 0, lane); return; }      // 548-553

    // piece 1: the whole read in [left1,right1) (557-566)
    IM_STAMP(0);
    const Band b1 = band_search<KT, DIRECT>(s, pk, (uint32_t)left1, (uint32_t)right1, (uint32_t)anchor, 0u, (uint32_t)L, k, g, lane, read_pk8 IM_STAMP_PASS(1));
    if (b1.st) { finish(out, b1.st, 1, lane); return; }
    const Aln a1 = diag_scan(s, contig, (uint32_t)left1, (uint32_t)right1, 0u, (uint32_t)L, b1.low, s.eq[0], lane);
    store_band(out, 0, b1, a1, lane);
    IM_STAMP(6);
    if (a1.st) { finish(out, a1.st, 1, lane); return; }
    const int r1 = a1.r1, r2 = a1.r2, q1 = a1.q1, q2 = a1.q2;
    if (q1 == q2) { finish(out, IM_ST_NONE, 1, lane); return; }                          // 568-572
    if (q1 == 0 && q2 == L) { finish(out, IM_ST_NONE, 1, lane); return; }                // 575-582: no I/D op at g = 0

    // piece 2: the rest of the read in the extended window, four cases (605-717)
    const uint32_t uL = (uint32_t)L, f = (uint32_t)a1.f, l = (uint32_t)a1.l;
    uint32_t w0, w1, anc, p0, p1; bool want_tail;
    if (r1 > anchor) {
        if (q1 == 0) {
            if (!(uL > f)) { finish(out, IM_ST_ABORT, 1, lane); return; }
            if ((uL - f) < eth || ((uint32_t)right2 - (uint32_t)r1 - f) < eth) { finish(out, IM_ST_NONE, 1, lane); return; }
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)right2; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
        } else if (q2 == L) {
            if (!(uL > l)) { finish(out, IM_ST_ABORT, 1, lane); return; }
            if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)anchor) < eth) { finish(out, IM_ST_NONE, 1, lane); return; }
            w0 = (uint32_t)anchor; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l; want_tail = false;
        } else { finish(out, IM_ST_NONE, 1, lane); return; }
    } else if (r1 < anchor) {
        if (r2 >= anchor) { finish(out, IM_ST_NONE, 1, lane); return; }
        if (q1 == 0) {
            if (!(uL > f)) { finish(out, IM_ST_ABORT, 1, lane); return; }
            if ((uL - f) < eth || ((uint32_t)anchor - (uint32_t)r1 - f) < eth) { finish(out, IM_ST_NONE, 1, lane); return; }
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)anchor; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
        } else if (q2 == L) {
            if (!(uL > l)) { finish(out, IM_ST_ABORT, 1, lane); return; }
            if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)left2) < eth) { finish(out, IM_ST_NONE, 1, lane); return; }
            w0 = (uint32_t)left2; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l; want_tail = false;
        } else { finish(out, IM_ST_NONE, 1, lane); return; }
    } else { finish(out, IM_ST_NONE, 1, lane); return; }                                  // r1 == anchor (712-717)
    if ((int32_t)(w1 - w0) <= 0) { finish(out, IM_ST_ABORT, 1, lane); return; }

    IM_STAMP(7);
    const Band b2 = band_search<KT, DIRECT>(s, pk, w0, w1, anc, p0, p1, k, g, lane, read_pk8 IM_STAMP_PASS(8));
    if (b2.st) { finish(out, b2.st, 2, lane); return; }
    const Aln a2 = diag_scan(s, contig, w0, w1, p0, p1, b2.low, s.eq[1], lane);
    store_band(out, 1, b2, a2, lane);
    IM_STAMP(13);
    if (a2.st) { finish(out, a2.st, 2, lane); return; }
    const int r3 = a2.r1, r4 = a2.r2, q3 = a2.q1, q4 = a2.q2;
    if (want_tail) { if (q4 != L || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 623-627, 679-683
    else           { if (q3 != 0 || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 645-649, 701-705
    if (!(q1 < q2 && q3 < q4)) { finish(out, IM_ST_ABORT, 2, lane); return; }             // 720-721

    // combine (723-754).  "A" = the piece that starts at read offset 0, "B" = the one that ends at L.
    const uint32_t *eqA, *eqB;
    int qa2, rA, qb1, rB;               // A = read[0,qa2) at contig rA.. ; B = read[qb1,L) at contig rB..
    bool split;                         // true: overlapping pieces, choose the split point (K4)
    if (q1 > q3 && q1 <= q4)        { eqA = s.eq[1]; qa2 = q4; rA = r3; eqB = s.eq[0]; qb1 = q1; rB = r1; split = true;  }
    else if (q3 > q1 && q3 <= q2)   { eqA = s.eq[0]; qa2 = q2; rA = r1; eqB = s.eq[1]; qb1 = q3; rB = r3; split = true;  }
    else if (q1 > q4 && r1 == r4)   { eqA = s.eq[1]; qa2 = q4; rA = r3; eqB = s.eq[0]; qb1 = q1; rB = r1; split = false; }
    else if (q3 > q2 && r2 == r3)   { eqA = s.eq[0]; qa2 = q2; rA = r1; eqB = s.eq[1]; qb1 = q3; rB = r3; split = false; }
    else { finish(out, IM_ST_NONE, 2, lane); return; }
    // find_best_del_candidate asserts its first piece starts at read offset 0 (314-315)
    // (holds by the accept conditions above: the A piece has q == 0)

    // per-position match flags of A on [0,qa2) and B on [qb1,L)
    const int x0 = 4 * lane;
    const uint32_t wa = eqA[lane], wb = eqB[lane];
    int fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        fa[j] = (x < qa2) ? (int)((wa >> (8 * j)) & 1u) : 0;
        fb[j] = (x >= qb1 && x < L) ? (int)((wb >> (8 * j)) & 1u) : 0;
    }
    const int ta = fa[0] + fa[1] + fa[2] + fa[3], tb = fb[0] + fb[1] + fb[2] + fb[3];
    const int ia = wave_scan_add(ta, lane), ib = wave_scan_add(tb, lane);
    const int totA = __builtin_amdgcn_readlane(ia, 63), totB = __builtin_amdgcn_readlane(ib, 63);
    int pa[4], pb[4];                   // exclusive prefix counts at x
    pa[0] = ia - ta; pb[0] = ib - tb;
#pragma unroll
    for (int j = 1; j < 4; j++) { pa[j] = pa[j - 1] + fa[j - 1]; pb[j] = pb[j - 1] + fb[j - 1]; }

    int index, nextindex, matches;
    if (split) {
        // count_matches(i) = '=' of A in read[0,i) + '=' of B in read[i,L); X counts are
        // L - that, so "max matches, then min mismatches, first wins" is the first maximum.
        int bs = -1, bx = INT_MAX;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            if (x >= qb1 && x <= qa2) {
                const int sc = pa[j] + (totB - pb[j]);
                if (sc > bs) { bs = sc; bx = x; }
            }
        }
        const int best = wave_max(bs);
        index = wave_min(bs == best ? bx : INT_MAX);
        if (best < 0 || index == INT_MAX) { finish(out, IM_ST_ABORT, 2, lane); return; }   // forceassert(index != -1)
        nextindex = index;
        matches = best;
    } else {
        index = qa2; nextindex = qb1;
        matches = totA + totB;
    }

    // update_readsegs (src/readaln.c:348-458) in closed form: A's runs over [0,index),
    // an I of nextindex-index bases if the pieces leave read bases uncovered, a D if
    // the reference positions leave a gap, then B's runs over [nextindex,L).
    const int refindx = rA + index;
    const int rindex  = rB + (nextindex - qb1);
    const bool hasI = nextindex > index;
    const bool hasD = refindx < rindex;
    if (!hasI && !hasD) { finish(out, IM_ST_NONE, 2, lane); return; }                      // no D/I segment -> NULL

    // run-length encode the final per-position classes
    int cls[4]; bool bnd[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        cls[j] = (x >= L) ? -1 : (x < index) ? (fa[j] ? IM_OP_EQ : IM_OP_X)
                 : (x < nextindex) ? IM_OP_I : (fb[j] ? IM_OP_EQ : IM_OP_X);
    }
    const int prevc = dpp_mov<kDppWaveShr1>(-2, cls[3]);      // lane 0 keeps -2
    int nb = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        const int pc = (j == 0) ? prevc : cls[j - 1];
        bnd[j] = (x < L) && (x == 0 || x == index || x == nextindex || cls[j] != pc);
        nb += bnd[j] ? 1 : 0;
    }
    const int inb = wave_scan_add(nb, lane);
    const int total_b = __builtin_amdgcn_readlane(inb, 63);
    const int n_ops = total_b + (hasD ? 1 : 0);
    if (n_ops > IM_MAX_OPS) { finish(out, IM_ST_OVERFLOW, 2, lane); return; }
    // run length = distance to the next boundary: boundary k leaves its position in LDS
    // (the vote histogram is idle by now), run k ends where boundary k+1 starts
    int32_t* bpos = reinterpret_cast<int32_t*>(s.diag);
    {
        int k = inb - nb;
#pragma unroll
        for (int j = 0; j < 4; j++) if (bnd[j]) bpos[k++] = x0 + j;
        if (lane == 0) bpos[total_b] = L;
    }
    wave_lds_sync();
    int slot = inb - nb;                 // boundaries before this lane
    int seg_indel = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        if (bnd[j]) {
            const int sl = slot + ((hasD && x >= nextindex) ? 1 : 0);
            out->ops[sl] = ((uint32_t)(bpos[slot + 1] - x) << 4) | (uint32_t)cls[j];
            if (x == index) seg_indel = slot;       // the I run itself, or the run the D op goes in front of
            slot++;
        }
    }
    seg_indel = wave_max(seg_indel);     // only one lane set it (others 0); slot >= 1 there
    if (lane == 0) {
        if (hasD) out->ops[seg_indel] = ((uint32_t)(rindex - refindx) << 4) | IM_OP_D;
        im_evidence* e = &out->ev[0];
        if (hasD) {
            e->cls = IM_CLS_DELETION; e->b1 = refindx; e->b2 = rindex;
            e->lflank = index; e->rflank = L - nextindex;
        } else {
            e->cls = IM_CLS_INSERTION; e->b1 = refindx; e->b2 = refindx;
            e->lflank = index; e->rflank = L - nextindex;
        }
        e->seg = seg_indel;
        e->read_off = index;
        // X bases left in aln1 + aln3: aligned bases minus '=' bases
        const int aligned = index + (L - nextindex);
        e->nd_print = aligned - matches;
        e->nd_filter = aligned - matches;
        out->ref_start = rA;
        out->n_ops = n_ops;
        out->n_ev = 1;
        out->status = IM_ST_EVIDENCE;
        out->n_band = 2;
    }
    write_slots(A, c, 1, hasD ? IM_CLS_DELETION : IM_CLS_INSERTION, refindx, hasD ? rindex : refindx, lane);
    IM_STAMP(14);
}

template <int KT, bool DIRECT>
__global__ __launch_bounds__(64) void realign_kernel(RealignArgs A)
{
    __shared__ WaveLds s;
    const int lane = threadIdx.x;
    const int G = gridDim.x;                    // multiple of 8
    const int per = G >> 3;
    // blocks with equal blockIdx % 8 share an XCD (observed round-robin placement,
    // speed only): give each XCD a contiguous run of `per` reads per sweep.
    const int mine = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if constexpr (KT == 6) {        // table_build<6> / table_undo6 keep the table clean from here on
#pragma unroll
        for (int i = 0; i < kTblBytes / 4 / 64; i++) s.tbl[lane + 64 * i] = 0u;
        wave_lds_sync();
    }
    for (int base = 0; base < A.batch.n; base += G) {
        const int c = base + mine;
        if (c < A.batch.n) realign_one<KT, DIRECT>(s, A, c, lane);
        wave_lds_sync();
    }
}

// 2-bit packing of the reference, 32 bases per word, first base in the top bits
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t* __restrict__ ascii, uint64_t* __restrict__ pk, int64_t n_words)
{
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
        const uint4* src = reinterpret_cast<const uint4*>(ascii + w * 32);
        const uint4 a = src[0], b = src[1];
        const uint32_t d[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
        uint64_t v = 0;
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) v = (v << 2) | code2((d[i] >> (8 * j)) & 255u);
        pk[w] = v;
    }
}

}  // namespace

#ifdef IM_STAMPS
extern "C" int im_debug_stamps_read(unsigned long long* out32, int reset)
{
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(im_stamp_acc), 32 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[32] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(im_stamp_acc), z, sizeof z);
    }
    return e == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_pack_reference(const uint8_t* ascii, uint64_t* pk, int64_t n_bases_padded, hipStream_t stream)
{
    const int64_t n_words = n_bases_padded / 32;
    if (n_words <= 0) return hipSuccess;
    int blocks = (int)((n_words + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(pack_kernel, dim3(blocks), dim3(256), 0, stream, ascii, pk, n_words);
    return hipGetLastError();
}

hipError_t launch_realign(const RealignArgs& a, int n_cu, hipStream_t stream)
{
    if (a.batch.n <= 0) return hipSuccess;
    // ~9 KiB LDS per one-wave workgroup -> up to 17 per CU; ask for 16 per CU.
    int64_t want = (int64_t)n_cu * 16;
    int64_t need = ((int64_t)a.batch.n + 7) / 8 * 8;
    int grid = (int)(need < want ? need : want);
    grid = (grid + 7) / 8 * 8;
    if (a.P.klength == 6 && a.P.numgaps == 0)
        hipLaunchKernelGGL((realign_kernel<6, true>), dim3(grid), dim3(64), 0, stream, a);     // reference defaults
    else if (a.P.klength <= (uint32_t)kDirectMaxK)
        hipLaunchKernelGGL((realign_kernel<0, true>), dim3(grid), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((realign_kernel<0, false>), dim3(grid), dim3(64), 0, stream, a);
    return hipGetLastError();
}

}  // namespace im
