// im_realign_long.hip -- split-read realignment of reads of 256 .. IM_MAX_READ bases (numgaps == 0).
//
// The same reference path as im_realign.hip (attempt_pe_alignment, src/alignment.c:764-799, with the band one
// diagonal wide) in a second lane layout: lane l owns the SIXTEEN read positions 16l .. 16l+15, the k-mer table
// stores 16-bit read offsets, the diagonal histogram counts in 16 bits (a 300-base read puts up to 295 votes on
// its diagonal), and the per-position match flags of the two band alignments live in LDS instead of a register.
// One wavefront per read, 23.3 KiB of LDS per wave.  The launch follows realign_kernel on the same stream when the
// context was told that such reads occur (im_expect_read_length); reads of up to 255 bases are left to that kernel,
// whose four-positions-per-lane layout is the fast one for them.
//
//   K1  find_best_band   src/alignment.c:393-447   band_search_long
//   K2  local_align      src/localalign.c:15-196   diag_scan_long (closed form of the one-diagonal band)
//   K3  ALIGN/fetch_cigar src/globalalign.c:333-401,507-604   = / X flags per read position
//   K4  find_best_del_candidate / count_matches  src/alignment.c:219-339
//   a10 update_readsegs  src/readaln.c:348-458
//   a11 new_evidence     src/evidence.c:4-34

#include "im_device.hpp"
#include "im_wave.hpp"

namespace im {
namespace {

constexpr int kLB = 16;                         // read positions per lane
static_assert(IM_MAX_READ + 4 <= 64 * kLB, "a lane owns 16 read positions");
constexpr int kLDiag = 2048;                    // diagonals per histogram pass (16 bits each)
constexpr int kLHash = 2048;                    // hash slots, k > 6 (at most IM_MAX_READ - 6 read k-mers)
constexpr int kLRead = 64 * kLB;                // 1024
constexpr uint32_t kRepeat = 0xFFFFu;           // hash table: the k-mer occurs more than once in the read piece

// DIRECT (k <= 6, the default): the table is the 8 KiB of 4^6 16-bit entries -- 15.4 KiB per wave, ten waves on a CU; the hash form
// (k > 6) holds kLHash keys and kLHash values, 16 KiB.
template <bool DIRECT>
struct LongLdsT {
    alignas(16) uint32_t diag[kLDiag / 2 + 8];
    alignas(16) uint32_t tbl[DIRECT ? kLHash : 2 * kLHash];
    alignas(16) uint32_t rd[(kLRead + 16) / 4]; // read bases, read coordinates
    alignas(16) uint8_t  eq[2][kLRead + 16];    // per band alignment: 1 where the read position is an aligned '='
    int32_t bpos[IM_MAX_OPS + 2];
};

struct LBand { int st, low, votes, win, piece; };

template <class LongLds>
__device__ __forceinline__ uint32_t read_kmer(const LongLds& s, uint32_t at, uint32_t k)
{
    uint32_t c = 0;
    for (uint32_t u = 0; u < k; u++) c |= code2(lds_byte(s.rd, at + u)) << (2u * u);
    return c;
}

__device__ __forceinline__ uint32_t hash_of(uint32_t code) { return (code * 2654435761u) >> 21; }      // 11 bits

// find_best_band (src/alignment.c:393-447) for numgaps == 0: read_seeds x2 (29-68), bin_diagonals (70-128),
// bin_bands (130-140: the band is the diagonal), select_band (142-181).
template <bool DIRECT, class LongLds>
__device__ LBand band_search_long(LongLds& s, const uint8_t* __restrict__ pk, uint32_t w0, uint32_t w1, uint32_t anchor,
                                  uint32_t p0, uint32_t p1, uint32_t k, int lane)
{
    LBand b;
    const uint32_t W = w1 - w0, Lp = p1 - p0;
    const uint32_t numdiag = (W - (k - 1)) + (Lp - (k - 1));     // unsigned, as written (403-404)
    b.win = (int)W; b.piece = (int)Lp; b.votes = 0; b.low = 0; b.st = 0;
    if (!(numdiag > 0u) || p1 < p0) { b.st = IM_ST_ABORT; return b; }   // forceasserts 405, 407
    if (Lp < k) { b.low = (int)(numdiag - 1); return b; }         // 408-412
    if ((int32_t)numdiag <= 0) { b.st = IM_ST_ABORT; return b; }  // the reference would run off its arrays

    const uint32_t nq = Lp - k + 1;                               // k-mers in the read piece
    const uint32_t npos = (W >= k) ? (W - k + 1) : 0u;            // k-mer starts in the window
    const uint32_t kmask = (1u << (2 * k)) - 1u;                  // k <= 15
    const int anchor_rel = (int)(anchor - w0);                    // select_band receives it as int (431, 146)
    uint16_t* t16 = reinterpret_cast<uint16_t*>(s.tbl);

    // ---- the read piece's k-mers; only those that occur once vote (bin_diagonals, 97-98) ----
    uint32_t code[kLB];
#pragma unroll
    for (int r = 0; r < kLB; r++) {
        const uint32_t q = (uint32_t)lane + 64u * r;
        code[r] = q < nq ? read_kmer(s, p0 + q, k) & kmask : 0u;
    }
    if (DIRECT) {
        // the table is all zero between two searches (the entries are taken back below)
#pragma unroll
        for (int r = 0; r < kLB; r++) { const uint32_t q = (uint32_t)lane + 64u * r; if (q < nq) t16[code[r]] = (uint16_t)(q + 1u); }
        wave_lds_sync();
        uint32_t lost = 0;
#pragma unroll
        for (int r = 0; r < kLB; r++) { const uint32_t q = (uint32_t)lane + 64u * r; if (q < nq && t16[code[r]] != (uint16_t)(q + 1u)) lost |= 1u << r; }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < kLB; r++) if ((lost >> r) & 1u) t16[code[r]] = 0;
    } else {
        for (int i = lane; i < kLHash; i += 64) { s.tbl[i] = 0xFFFFFFFFu; s.tbl[kLHash + i] = 0u; }
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < kLB; r++) {
            const uint32_t q = (uint32_t)lane + 64u * r;
            if (q >= nq) continue;
            uint32_t h = hash_of(code[r]);
            for (int probe = 0; probe < kLHash; probe++) {
                const uint32_t old = atomicCAS(&s.tbl[h], 0xFFFFFFFFu, code[r]);
                if (old == 0xFFFFFFFFu) { atomicMax(&s.tbl[kLHash + h], q + 1u); break; }
                if (old == code[r])     { atomicMax(&s.tbl[kLHash + h], kRepeat); break; }
                h = (h + 1) & (kLHash - 1);
            }
        }
    }
    wave_lds_sync();

    int bc = 0, bd = INT_MAX, bi = 0;                             // select_band's max, dist, indx
    for (uint32_t c0 = 0; c0 < numdiag; c0 += kLDiag) {
        for (int i = lane; i < kLDiag / 2 + 8; i += 64) s.diag[i] = 0u;
        wave_lds_sync();
        // window k-mer at p and read-unique k-mer at q land on diagonal p - q + nq (102-105): diagonals of this pass
        // come from p in [c0 - nq, c0 + kLDiag - 2]
        const int p_lo = max(0, (int)c0 - (int)nq);
        const int p_hi = min((int)npos - 1, (int)(c0 + kLDiag) - 2);
        for (int p = p_lo + lane; p <= p_hi; p += 64) {
            const uint32_t P = w0 + (uint32_t)p;
            uint64_t dd;
            __builtin_memcpy(&dd, pk + (P >> 2), 8);
            const uint32_t c = (uint32_t)(dd >> (2u * (P & 3u))) & kmask;
            uint32_t v;
            if (DIRECT) v = t16[c];
            else {
                v = 0u;
                uint32_t h = hash_of(c);
                for (int probe = 0; probe < kLHash; probe++) {
                    const uint32_t key = s.tbl[h];
                    if (key == c) { v = s.tbl[kLHash + h]; break; }
                    if (key == 0xFFFFFFFFu) break;
                    h = (h + 1) & (kLHash - 1);
                }
                if (v == kRepeat) v = 0u;
            }
            if (v == 0u) continue;
            const uint32_t off = (uint32_t)p - (v - 1u) + nq - c0;
            if (off < (uint32_t)kLDiag) atomicAdd(&s.diag[off >> 1], 1u << (16u * (off & 1u)));
        }
        wave_lds_sync();
        // select_band over the pass: most votes, then nearest the anchor, then the smallest index
        const uint32_t iend = min(c0 + (uint32_t)kLDiag, numdiag);
        int mc = -1, md = INT_MAX, mi = INT_MAX;
        for (uint32_t i = c0 + (uint32_t)lane; i < iend; i += 64) {
            const uint32_t o = i - c0;
            const int cnt = (int)((s.diag[o >> 1] >> (16u * (o & 1u))) & 0xFFFFu);
            const int d = abs((int)((uint32_t)anchor_rel - i));
            if (cnt > mc || (cnt == mc && d < md)) { mc = cnt; md = d; mi = (int)i; }     // i ascends: the first of equals stays
        }
        const int M = wave_max(mc);
        const int D = wave_min(mc == M ? md : INT_MAX);
        const int I = wave_min((mc == M && md == D) ? mi : INT_MAX);
        if (M > bc || (M == bc && D < bd)) { bc = M; bd = D; bi = I; }
        wave_lds_sync();
    }
    if (DIRECT) {
#pragma unroll
        for (int r = 0; r < kLB; r++) { const uint32_t q = (uint32_t)lane + 64u * r; if (q < nq) t16[code[r]] = 0; }
        wave_lds_sync();
    }
    b.votes = bc;
    b.low = bi - (int)nq;                                          // 438
    return b;
}

struct LAln {
    int st;                 // 0 ok, IM_ST_ABORT
    int q1, q2, r1, r2;     // 0-based half-open read / contig coordinates; q1 == q2: no alignment
    int f, l;               // leading / trailing '=' run (src/alignment.c:585-599)
};

// local_align + ALIGN + fetch_cigar for low == up == d (src/localalign.c:100-176 with band == 1):
//   forward : c_t = max(0, c_{t-1} + w_t); end = first t where c_t is the strict maximum
//   reverse : start = largest s <= end with sum_{s..end} w == best
// The '=' flags of the aligned positions land in s.eq[which], in read coordinates.
template <class LongLds>
__device__ LAln diag_scan_long(LongLds& s, int which, const uint8_t* __restrict__ contig, uint32_t w0, uint32_t w1, uint32_t p0, uint32_t p1,
                               int d, int lane)
{
    LAln a;
    a.st = 0; a.q1 = a.q2 = a.r1 = a.r2 = 0; a.f = a.l = 0;
    {
        uint4* e4 = reinterpret_cast<uint4*>(s.eq[which]);
        if (lane < (kLRead + 16) / 16) e4[lane] = make_uint4(0u, 0u, 0u, 0u);
        if (lane == 0) e4[64] = make_uint4(0u, 0u, 0u, 0u);
    }
    wave_lds_sync();
    const int M = (int)(p1 - p0), N = (int)(w1 - w0);
    if (M <= 0 || N <= 0 || d < -M || d > N) { a.st = IM_ST_ABORT; return a; }   // src/localalign.c:31-32,70-77
    const int t_lo = max(0, -d), t_hi = min(M, N - d);

    const int t0 = kLB * lane;
    uint8_t rf[kLB];
#pragma unroll
    for (int j = 0; j < kLB; j++) rf[j] = 0;
    const bool mine = t0 < t_hi && t0 + kLB - 1 >= t_lo;
    if (mine) __builtin_memcpy(rf, contig + ((int64_t)w0 + d + t0), kLB);          // >= 64 zero bytes around every contig
    int w[kLB]; uint32_t eqm = 0;
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int t = t0 + j;
        const bool valid = t >= t_lo && t < t_hi;
        const bool e = valid && lds_byte(s.rd, p0 + (uint32_t)t) == (uint32_t)rf[j];
        eqm |= e ? 1u << j : 0u;
        w[j] = valid ? (e ? kScoreMatch : kScoreMismatch) : 0;
    }
    // inclusive prefix sums S_t
    int S[kLB];
    S[0] = w[0];
#pragma unroll
    for (int j = 1; j < kLB; j++) S[j] = S[j - 1] + w[j];
    const int incl = wave_scan_add(S[kLB - 1], lane);
    const int excl = incl - S[kLB - 1];
#pragma unroll
    for (int j = 0; j < kLB; j++) S[j] += excl;
    // running minimum of S including the empty prefix (0)
    int m[kLB];
    m[0] = S[0];
#pragma unroll
    for (int j = 1; j < kLB; j++) m[j] = min(m[j - 1], S[j]);
    const int pm = min(0, wave_scan_min_excl(m[kLB - 1], lane));
    int best_l = INT_MIN;
#pragma unroll
    for (int j = 0; j < kLB; j++) best_l = max(best_l, S[j] - min(pm, m[j]));
    const int best = wave_max(best_l);
    if (best <= 0) return a;                                       // score <= 0 (src/alignment.c:365-372)
    int e_loc = INT_MAX;
#pragma unroll
    for (int j = kLB - 1; j >= 0; j--) if (S[j] - min(pm, m[j]) == best) e_loc = t0 + j;
    const int end = wave_min(e_loc);
    int s_sel = INT_MIN;
#pragma unroll
    for (int j = 0; j < kLB; j++) if (t0 + j == end) s_sel = S[j];
    const int Send = wave_max(s_sel);
    const int target = Send - best;
    int st_loc = -1;
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int t = t0 + j;
        const int sprev = (j == 0) ? excl : S[j - 1];
        if (t >= t_lo && t <= end && sprev == target) st_loc = t;
    }
    const int start = wave_max(st_loc);
    if (start < 0 || end == start) return a;                       // single cell: score 0 (src/localalign.c:191-193)

    int fm = INT_MAX, lm = -1;
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int t = t0 + j;
        if (t >= start && t <= end) {
            if ((eqm >> j) & 1u) s.eq[which][p0 + (uint32_t)t] = 1;
            else { fm = min(fm, t); lm = max(lm, t); }
        }
    }
    fm = wave_min(fm); lm = wave_max(lm);
    a.f = (fm == INT_MAX ? end + 1 : fm) - start;
    a.l = end - (lm < 0 ? start - 1 : lm);
    wave_lds_sync();
    a.q1 = (int)p0 + start;                                         // src/alignment.c:385-388
    a.q2 = (int)p0 + end + 1;
    a.r1 = (int)w0 + d + start;
    a.r2 = (int)w0 + d + end + 1;
    return a;
}

__device__ __forceinline__ void store_band_long(im_read_result* out, int which, const LBand& b, const LAln& a, int lane)
{
    if (lane == 0) {
        im_band_aln* o = &out->band[which];
        o->r1 = a.r1; o->r2 = a.r2; o->q1 = a.q1; o->q2 = a.q2;
        o->low = b.low; o->votes = b.votes; o->win_bytes = b.win; o->piece_bytes = b.piece;
    }
}

// attempt_pe_alignment -> attempt_diagonal_alignments (src/alignment.c:539-799)
template <bool DIRECT, class LongLds>
__device__ void realign_long_one(LongLds& s, const RealignArgs& A, int c, int L, int lane)
{
    im_read_result* out = &A.batch.out[c];
    const int64_t off = sload(A.batch.base_off + c);
    const int tid = sload(A.batch.tid + c);
    const int anchor = sload(A.batch.anchor + c);
    const int R = sload(A.batch.range_max + c);
    const uint32_t k = A.P.klength, eth = A.P.ethreshold;

    if (lane < 16) reinterpret_cast<uint32_t*>(&out->band[0])[lane] = 0u;
    if (lane < 7) out->reserved[lane] = 0;
    if (!A.keep_slots) write_slots(A, c, 0, -1, 0, 0, lane);
    if (tid < 0 || tid >= A.ref.n_contigs || (off & 3)) { finish(out, (off & 3) ? IM_ST_UNSUPPORTED : IM_ST_ABORT, 0, lane); return; }
    const uint8_t* contig = A.ref.ascii + sload(A.ref.asc_off + tid);
    const uint8_t* pk = A.ref.pk + sload(A.ref.pk_off + tid);
    const int clen = sload(A.ref.len + tid);

    // stage the read, zeroes behind it
    for (int i = lane; i < (kLRead + 16) / 4; i += 64) {
        uint32_t v = 0;
        if (4 * i < L) v = *reinterpret_cast<const uint32_t*>(A.batch.bases + off + 4 * i);
        const int rem = L - 4 * i;
        if (rem < 4) v &= (rem <= 0) ? 0u : ((1u << (8 * rem)) - 1u);
        s.rd[i] = v;
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
    const LBand b1 = band_search_long<DIRECT>(s, pk, (uint32_t)left1, (uint32_t)right1, (uint32_t)anchor, 0u, (uint32_t)L, k, lane);
    if (b1.st) { finish(out, b1.st, 1, lane); return; }
    const LAln a1 = diag_scan_long(s, 0, contig, (uint32_t)left1, (uint32_t)right1, 0u, (uint32_t)L, b1.low, lane);
    store_band_long(out, 0, b1, a1, lane);
    if (a1.st) { finish(out, a1.st, 1, lane); return; }
    const int r1 = a1.r1, r2 = a1.r2, q1 = a1.q1, q2 = a1.q2;
    if (q1 == q2) { finish(out, IM_ST_NONE, 1, lane); return; }                          // 568-572
    if (q1 == 0 && q2 == L) { finish(out, IM_ST_NONE, 1, lane); return; }                // 575-582: no I/D op without gaps

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

    const LBand b2 = band_search_long<DIRECT>(s, pk, w0, w1, anc, p0, p1, k, lane);
    if (b2.st) { finish(out, b2.st, 2, lane); return; }
    const LAln a2 = diag_scan_long(s, 1, contig, w0, w1, p0, p1, b2.low, lane);
    store_band_long(out, 1, b2, a2, lane);
    if (a2.st) { finish(out, a2.st, 2, lane); return; }
    const int r3 = a2.r1, r4 = a2.r2, q3 = a2.q1, q4 = a2.q2;
    if (want_tail) { if (q4 != L || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 623-627, 679-683
    else           { if (q3 != 0 || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 645-649, 701-705
    if (!(q1 < q2 && q3 < q4)) { finish(out, IM_ST_ABORT, 2, lane); return; }             // 720-721

    // combine (723-754).  "A" = the piece that starts at read offset 0, "B" = the one that ends at L.
    int wa, wb;                         // which s.eq[] holds the A / B piece
    int qa2, rA, qb1, rB;               // A = read[0,qa2) at contig rA.. ; B = read[qb1,L) at contig rB..
    bool split;                         // true: overlapping pieces, choose the split point (K4)
    if (q1 > q3 && q1 <= q4)        { wa = 1; qa2 = q4; rA = r3; wb = 0; qb1 = q1; rB = r1; split = true;  }
    else if (q3 > q1 && q3 <= q2)   { wa = 0; qa2 = q2; rA = r1; wb = 1; qb1 = q3; rB = r3; split = true;  }
    else if (q1 > q4 && r1 == r4)   { wa = 1; qa2 = q4; rA = r3; wb = 0; qb1 = q1; rB = r1; split = false; }
    else if (q3 > q2 && r2 == r3)   { wa = 0; qa2 = q2; rA = r1; wb = 1; qb1 = q3; rB = r3; split = false; }
    else { finish(out, IM_ST_NONE, 2, lane); return; }

    // per-position match flags of A on [0,qa2) and B on [qb1,L), as bit masks of this lane's sixteen positions
    const int x0 = kLB * lane;
    uint32_t fa = 0, fb = 0;
    {
        uint8_t ea[kLB], eb[kLB];
        __builtin_memcpy(ea, &s.eq[wa][x0], kLB);
        __builtin_memcpy(eb, &s.eq[wb][x0], kLB);
#pragma unroll
        for (int j = 0; j < kLB; j++) {
            const int x = x0 + j;
            fa |= (x < qa2 && ea[j]) ? 1u << j : 0u;
            fb |= (x >= qb1 && x < L && eb[j]) ? 1u << j : 0u;
        }
    }
    const int ta = __popc(fa), tb = __popc(fb);
    const int ia = wave_scan_add(ta, lane), ib = wave_scan_add(tb, lane);
    const int totA = __builtin_amdgcn_readlane(ia, 63), totB = __builtin_amdgcn_readlane(ib, 63);
    const int ea0 = ia - ta, eb0 = ib - tb;             // '=' of A / B in front of x0

    int index, nextindex, matches;
    if (split) {
        // count_matches(i) = '=' of A in read[0,i) + '=' of B in read[i,L); X counts are L - that, so "max matches,
        // then min mismatches, first wins" is the first maximum.
        int bs = -1, bx = INT_MAX;
#pragma unroll
        for (int j = 0; j < kLB; j++) {
            const int x = x0 + j;
            if (x >= qb1 && x <= qa2) {
                const uint32_t below = (1u << j) - 1u;
                const int sc = ea0 + __popc(fa & below) + (totB - eb0 - __popc(fb & below));
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

    // update_readsegs (src/readaln.c:348-458) in closed form: A's runs over [0,index), an I of nextindex-index bases if
    // the pieces leave read bases uncovered, a D if the reference positions leave a gap, then B's runs over [nextindex,L).
    const int refindx = rA + index;
    const int rindex  = rB + (nextindex - qb1);
    const bool hasI = nextindex > index;
    const bool hasD = refindx < rindex;
    if (!hasI && !hasD) { finish(out, IM_ST_NONE, 2, lane); return; }                      // no D/I segment -> NULL

    // run-length encode the final per-position classes
    int cls[kLB];
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int x = x0 + j;
        cls[j] = (x >= L) ? -1 : (x < index) ? (((fa >> j) & 1u) ? IM_OP_EQ : IM_OP_X)
                 : (x < nextindex) ? IM_OP_I : (((fb >> j) & 1u) ? IM_OP_EQ : IM_OP_X);
    }
    const int prevc = dpp_mov<kDppWaveShr1>(-2, cls[kLB - 1]);      // lane 0 keeps -2
    uint32_t bnd = 0;
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int x = x0 + j;
        const int pc = (j == 0) ? prevc : cls[j - 1];
        if ((x < L) && (x == 0 || x == index || x == nextindex || cls[j] != pc)) bnd |= 1u << j;
    }
    const int nb = __popc(bnd);
    const int inb = wave_scan_add(nb, lane);
    const int total_b = __builtin_amdgcn_readlane(inb, 63);
    const int n_ops = total_b + (hasD ? 1 : 0);
    if (n_ops > IM_MAX_OPS) { finish(out, IM_ST_OVERFLOW, 2, lane); return; }
    // run length = distance to the next boundary
    {
        int kk = inb - nb;
#pragma unroll
        for (int j = 0; j < kLB; j++) if ((bnd >> j) & 1u) s.bpos[kk++] = x0 + j;
        if (lane == 0) s.bpos[total_b] = L;
    }
    wave_lds_sync();
    int slot = inb - nb;                 // boundaries before this lane
    int seg_indel = 0;
#pragma unroll
    for (int j = 0; j < kLB; j++) {
        const int x = x0 + j;
        if ((bnd >> j) & 1u) {
            const int sl = slot + ((hasD && x >= nextindex) ? 1 : 0);
            out->ops[sl] = ((uint32_t)(s.bpos[slot + 1] - x) << 4) | (uint32_t)cls[j];
            if (x == index) seg_indel = slot;       // the I run itself, or the run the D op goes in front of
            slot++;
        }
    }
    seg_indel = wave_max(seg_indel);     // only one lane set it (others 0); slot >= 1 there
    if (lane == 0) {
        if (hasD) out->ops[seg_indel] = ((uint32_t)(rindex - refindx) << 4) | IM_OP_D;
        im_evidence* e = &out->ev[0];
        e->cls = hasD ? IM_CLS_DELETION : IM_CLS_INSERTION;
        e->b1 = refindx; e->b2 = hasD ? rindex : refindx;
        e->lflank = index; e->rflank = L - nextindex;
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
    wave_lds_sync();
}

template <bool DIRECT>
__global__ __launch_bounds__(64, (DIRECT ? 3 : 2)) void realign_long_kernel(RealignArgs A)
{
    __shared__ LongLdsT<DIRECT> s;
    const int lane = threadIdx.x;
    // ONE read per wave, as in realign_kernel: a 2 x 300 library is all long reads, and a wave per 64 consecutive reads (the first
    // form of this kernel: long reads were the exception then) left the chip a wave per CU.  The launch covers every read of the
    // slice; the waves of reads the four-positions-per-lane kernel has served leave at once.
    const int n = A.n_dev ? min(sload(A.n_dev), A.batch.n) : A.batch.n;
    const int left = n - A.first;
    const int G = min((int)gridDim.x, (left + 7) / 8 * 8);
    if ((int)blockIdx.x >= G) return;
    const int c = A.first + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);      // blocks of one XCD take neighbouring reads
    if (c >= n) return;
    const int L = sload(A.batch.read_len + c);
    if (!(L > kShortRead && L <= IM_MAX_READ)) return;
    if (DIRECT) {
        uint4* t4 = reinterpret_cast<uint4*>(s.tbl);
        for (int i = lane; i < (int)(sizeof(s.tbl) / 16); i += 64) t4[i] = make_uint4(0u, 0u, 0u, 0u);
        wave_lds_sync();
    }
    realign_long_one<DIRECT>(s, A, c, L, lane);
}

}  // namespace

hipError_t launch_realign_long(const RealignArgs& a, int n_cu, hipStream_t stream)
{
    if (a.batch.n <= 0 || a.P.numgaps != 0) return hipSuccess;
    const int64_t cap = (int64_t)n_cu * 4096 / 8 * 8;       // one read per wave: a batch beyond the largest grid goes out in slices
    for (int64_t first = 0; first < (int64_t)a.batch.n; first += cap) {
        RealignArgs b = a;
        b.first = (int32_t)first;
        const int64_t rest = ((int64_t)a.batch.n - first + 7) / 8 * 8;
        const int g = (int)(rest < cap ? rest : cap);
        if (a.P.klength <= 6)
            hipLaunchKernelGGL((realign_long_kernel<true>), dim3(g), dim3(64), 0, stream, b);
        else
            hipLaunchKernelGGL((realign_long_kernel<false>), dim3(g), dim3(64), 0, stream, b);
    }
    return hipGetLastError();
}

}  // namespace im
