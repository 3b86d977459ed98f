// im_realign_any.hip -- split-read realignment WITHOUT the bounds of the laid-out kernels.
//
// The reference realigns reads of any length with a band of any width (src/readaln.c:242-267, src/indelminer.c:934,948:
// numgaps is an unbounded unsigned).  The kernels of im_realign.hip / im_realign_long.hip are laid out for what sequencers
// deliver -- reads of up to 255 / 1020 bases at numgaps == 0, up to 255 bases and bands of up to 61 diagonals otherwise (one
// lane per diagonal) -- and leave every other read IM_ST_UNSUPPORTED.  This file takes those: ONE LANE PER READ for the
// dynamic programs and the merge, their state in a per-lane arena of device memory whose words are interleaved over the 64
// lanes of a wave that hold a read (word i of lane l lives at arena[R i + l], so lanes that walk their arrays in step touch a
// few cache lines per access, not one each; R = 64 for large batches, fewer when the batch would leave SIMDs without a wave); the k-mer band searches in between by the WHOLE WAVE on one read after the other.  It follows
// the reference statement by statement where order matters (the strict comparisons of the three dynamic programs decide
// between co-optimal alignments) and restates what is order-free (the k-mer tables as one open-addressing
// table per read piece, the band sums as a running sum):
//
//   K1  find_best_band           src/alignment.c:393-447 (read_seeds 29-68, bin_diagonals 70-128, bin_bands 130-140, select_band 142-181)
//   K2  local_align              src/localalign.c:15-196
//   K3  ALIGN / align            src/globalalign.c:66-401, the recursion as frames; the script kept per read position
//       fetch_cigar              src/globalalign.c:507-604
//   K4  find_best_del_candidate  src/alignment.c:219-339
//   a8  attempt_diagonal_alignments  src/alignment.c:539-759
//   a10 update_readsegs          src/readaln.c:348-458
//   a11 new_evidence             src/evidence.c:4-34 (+ the reductions of src/variant.c:217-290,704-775)
//
// Two launches: any_pick_kernel lists the reads the other kernels left (or, for numgaps > 60, every read) with the largest
// read and window among them; the host sizes the arena from those and realign_any_kernel works the list off, every lane
// of a wave claiming a read per round.  The result record's own bounds (IM_MAX_OPS segments, IM_MAX_EV indels per read) stay.

#include "im_device.hpp"
#include "im_wave.hpp"

namespace im {
namespace {

constexpr int kNegInf = -9999999;            // MININT, src/localalign.c:3
constexpr int kOpen = 10, kExt = 10;         // src/localalign.c:10-13
constexpr int kFrameWords = 16, kFrames = 48;
constexpr int kCoopMaxBand = 32;             // bands up to this wide: the wave searches together and steps together (realign_any_kernel)
constexpr uint32_t kEmpty = 0xFFFFFFFFu;
enum : int { kKindEq = 0, kKindX = 1, kKindI = 2, kKindNone = 3 };

// arena layout in words per lane (the same function sizes the arena on the host)
struct AnyLayout {
    int32_t maxM, maxW, maxB, hslots;
    int32_t lane_shift;         // log2 of the reads a wave takes per round (R = 1 << lane_shift lanes hold a read each; the arena is
                                // interleaved over R lanes): 64 for large batches, fewer when the batch would leave SIMDs without a wave
    int32_t rows_in_lds;        // 1: o_cc .. o_dp index the wave's LDS block (from 0), not the arena
    int32_t l_rows_words;       // LDS words of the four rows (0 when they live in the arena)
    int32_t l_tab_slots;        // slots of the wave's k-mer table in LDS (key, offset, count: 3 words each); 0: every lane searches alone
    int32_t l_diag_words;       // LDS words of the wave's diagonal histogram (longer ones go to the arena)
    int32_t lds_bytes;
    int32_t o_hkey, o_hpos, o_hcnt, o_diag, o_cc, o_dd, o_cp, o_dp, o_mp0, o_mp1, o_mp2, o_fp, o_pos0, o_pos1, o_ops0, o_ops1, o_fin, o_stk;
    int32_t words;
};

__host__ __device__ inline AnyLayout make_layout(int32_t maxM, int32_t maxW, int32_t maxB, int32_t lane_shift)
{
    AnyLayout y;
    y.maxM = maxM; y.maxW = maxW; y.maxB = maxB; y.lane_shift = lane_shift;
    const int32_t R = 1 << lane_shift;
    int32_t h = 16; while (h < 2 * maxM) h <<= 1;
    y.hslots = h;
    int32_t at = 0;
    auto take = [&](int32_t n) { const int32_t o = at; at += n; return o; };
    y.o_hkey = take(h); y.o_hpos = take(h); y.o_hcnt = take(h);
    y.o_diag = take(maxW + maxM + 8);
    // the four rows: 4 * (maxB + 4) words per lane, 256 bytes per word over the wave; in LDS up to 64 KiB of it
    y.rows_in_lds = (size_t)4 * (size_t)(maxB + 4) * 4u * (size_t)R <= ((size_t)64 << 10) ? 1 : 0;
    if (y.rows_in_lds) { y.o_cc = 0; y.o_dd = maxB + 4; y.o_cp = 2 * (maxB + 4); y.o_dp = 3 * (maxB + 4); }
    else { y.o_cc = take(maxB + 4); y.o_dd = take(maxB + 4); y.o_cp = take(maxB + 4); y.o_dp = take(maxB + 4); }
    y.o_mp0 = take(maxM + 2); y.o_mp1 = take(maxM + 2); y.o_mp2 = take(maxM + 2); y.o_fp = take(maxM + 2);
    y.o_pos0 = take(maxM + 2); y.o_pos1 = take(maxM + 2);       // per read position: kind | reference bases skipped in front << 2
    y.o_ops0 = take(IM_MAX_OPS + 4); y.o_ops1 = take(IM_MAX_OPS + 4); y.o_fin = take(IM_MAX_OPS + 4);
    y.o_stk = take(kFrames * kFrameWords);
    y.words = at;
    // the wave's LDS: the rows (64 lanes' worth), then the shared table and histogram of the cooperative band search, 144 KiB at
    // most (a workgroup may take all of a CU's 160 KiB): tables of reads of up to 4096 bases
    const int32_t budget = (144 << 10) / 4;
    y.l_rows_words = y.rows_in_lds ? 4 * (maxB + 4) * R : 0;
    y.l_tab_slots = (y.l_rows_words + 3 * h + 1024 <= budget) ? h : 0;
    int32_t left = budget - y.l_rows_words - 3 * y.l_tab_slots;
    if (left > maxW + maxM + 8) left = maxW + maxM + 8;
    y.l_diag_words = y.l_tab_slots ? left : 0;
    y.lds_bytes = 4 * (y.l_rows_words + 3 * y.l_tab_slots + y.l_diag_words);
    return y;
}

// one lane's view of its arena.  The four rows of the banded passes (CC / DD / CP / DP: the arrays every cell of the dynamic
// programs reads and writes) live in LDS when the band is narrow enough for 64 lanes' rows to fit -- the launch decides --
// and in the arena otherwise: q is that home, lane-interleaved like the arena (offsets o_cc .. then count from 0).
#define IM_GLOBAL_AS __attribute__((address_space(1)))
typedef IM_GLOBAL_AS const uint8_t* gbytes;        // the contig's and the read's bytes: device memory, and said to be
#define IM_LDS_AS __attribute__((address_space(3)))
template <bool LDSROWS> struct RowPtr { typedef IM_GLOBAL_AS int32_t* T; };
template <> struct RowPtr<true> { typedef IM_LDS_AS int32_t* T; };
// The pointers carry their address space (global arena, LDS rows): through the calls below a plain pointer would be a generic one and
// every access a flat instruction (1 070 of them in the first build) -- slower than a global or LDS access and counted on both wait counters.
template <bool LDSROWS>
struct ArT {
    IM_GLOBAL_AS int32_t* p;
    typename RowPtr<LDSROWS>::T q;
    int32_t sh;         // AnyLayout::lane_shift
    __device__ __forceinline__ auto& at(int32_t off, int32_t i) const { return p[(int64_t)(off + i) << sh]; }
    __device__ __forceinline__ auto& atu(int32_t off, int32_t i) const { return ((IM_GLOBAL_AS uint32_t*)p)[(int64_t)(off + i) << sh]; }
    __device__ __forceinline__ auto& row(int32_t off, int32_t i) const { return q[(int64_t)(off + i) << sh]; }
};

__device__ __forceinline__ int wsub(uint32_t a, uint32_t b) { return a == b ? kScoreMatch : kScoreMismatch; }   // W[i][j], src/localalign.c:61-67

struct AnyBand { int st, low, up, votes; };

// find_best_band.  ref / read are the contig and the read; the window is [z1, e1), the piece [z2, e2).
template <class AR>
__device__ AnyBand any_find_band(const AR& a, const AnyLayout& Y, gbytes ref, uint32_t z1, uint32_t e1, uint32_t anchor,
                                 gbytes read, uint32_t z2, uint32_t e2, uint32_t k, uint32_t g)
{
    AnyBand b; b.st = 0; b.low = b.up = 0; b.votes = 0;
    const uint32_t W = e1 - z1, L = e2 - z2;
    const uint32_t numdiag = (W - (k - 1)) + (L - (k - 1));                 // unsigned, as written (403-404)
    if (!(numdiag > g) || e2 < z2) { b.st = IM_ST_ABORT; return b; }       // forceasserts 405, 407
    if (L < k) { b.low = b.up = (int)(numdiag - 1); return b; }            // 408-412
    if ((int32_t)numdiag <= 0 || numdiag > (1u << 28)) { b.st = IM_ST_ABORT; return b; }
    if (numdiag > (uint32_t)(Y.maxW + Y.maxM + 8) || L > (uint32_t)Y.maxM) { b.st = IM_ST_OVERFLOW; return b; }
    const uint32_t mask = (1u << (2 * k)) - 1u, hm = (uint32_t)Y.hslots - 1u;
    for (int32_t i = 0; i < Y.hslots; i++) { a.atu(Y.o_hkey, i) = kEmpty; a.at(Y.o_hcnt, i) = 0; }
    // the read piece's k-mers: one table entry per distinct k-mer with its count and (for count == 1) its offset
    uint32_t code = 0;
    for (uint32_t i = 0; i < L; i++) {
        code = ((code << 2) | code2(read[z2 + i])) & mask;
        if (i + 1 < k) continue;
        uint32_t h = (code * 2654435761u) >> 7 & hm;
        for (;;) {
            const uint32_t key = a.atu(Y.o_hkey, (int32_t)h);
            if (key == kEmpty) { a.atu(Y.o_hkey, (int32_t)h) = code; a.at(Y.o_hpos, (int32_t)h) = (int32_t)(i + 1 - k); a.at(Y.o_hcnt, (int32_t)h) = 1; break; }
            if (key == code) { a.at(Y.o_hcnt, (int32_t)h) += 1; break; }
            h = (h + 1) & hm;
        }
    }
    for (uint32_t i = 0; i < numdiag; i++) a.at(Y.o_diag, (int32_t)i) = 0;
    // every window position whose k-mer occurs exactly once in the piece votes for its diagonal (97-112)
    if (W >= k) {
        code = 0;
        for (uint32_t i = 0; i < W; i++) {
            code = ((code << 2) | code2(ref[z1 + i])) & mask;
            if (i + 1 < k) continue;
            uint32_t h = (code * 2654435761u) >> 7 & hm;
            for (;;) {
                const uint32_t key = a.atu(Y.o_hkey, (int32_t)h);
                if (key == kEmpty) break;
                if (key == code) {
                    if (a.at(Y.o_hcnt, (int32_t)h) == 1) {
                        const uint32_t indx = (i + 1 - k) - (uint32_t)a.at(Y.o_hpos, (int32_t)h) + L - k + 1;     // 102-105
                        if (indx < numdiag) a.at(Y.o_diag, (int32_t)indx) += 1;
                    }
                    break;
                }
                h = (h + 1) & hm;
            }
        }
    }
    // bin_bands + select_band: bands[i] = diag[i] + .. + diag[i + g] for i < numdiag - g, else 0; the most votes, then the
    // band nearest the anchor, then the first
    const int anchor_rel = (int)(anchor - z1);
    int best = 0, dist = INT_MAX; uint32_t indx = 0;
    int run = 0;
    for (uint32_t j = 0; j <= g; j++) run += a.at(Y.o_diag, (int32_t)j);
    for (uint32_t i = 0; i < numdiag; i++) {
        const int bsum = i < numdiag - g ? run : 0;
        int d = (int)((uint32_t)anchor_rel - i); if (d < 0) d = -d;
        if (bsum > best) { best = bsum; indx = i; dist = d; }
        else if (bsum == best && d < dist) { indx = i; dist = d; }
        if (i + g + 1 < numdiag) run += a.at(Y.o_diag, (int32_t)(i + g + 1)) - a.at(Y.o_diag, (int32_t)i);
    }
    b.votes = best;
    b.low = (int)(indx - (L - k + 1));                                      // 438-439
    b.up = (int)(indx + g - (L - k + 1));
    return b;
}

// ---- ALIGN (src/globalalign.c:66-401) ----------------------------------------------------------------------------------

// The edit script per READ POSITION of the piece (pos = A0's base index): kind of the base (= / X / I) and the reference
// bases skipped in front of it -- what DEL / INS / REP (36-59) append, in a form fetch_cigar's run lengths fall out of.
struct Script { int32_t o_pos; int32_t pos0; int32_t ia, jb; };

template <class AR>
__device__ __forceinline__ void s_rep(const AR& a, Script& S, gbytes A0, gbytes B0)
{
    const int32_t p = S.pos0 + S.ia;
    const int32_t w = a.at(S.o_pos, p);
    a.at(S.o_pos, p) = (w & ~3) | (A0[S.ia + 1] == B0[S.jb + 1] ? kKindEq : kKindX);
    S.ia++; S.jb++;
}
template <class AR>
__device__ __forceinline__ void s_del(const AR& a, Script& S, int n)       // read bases without a partner
{
    for (int t = 0; t < n; t++) { const int32_t p = S.pos0 + S.ia + t; a.at(S.o_pos, p) = (a.at(S.o_pos, p) & ~3) | kKindI; }
    S.ia += n;
}
template <class AR>
__device__ __forceinline__ void s_ins(const AR& a, Script& S, int n)       // reference bases skipped
{
    a.at(S.o_pos, S.pos0 + S.ia) += n << 2;
    S.jb += n;
}

// crossing records of the middle diagonal: pointer + 1 in the upper bits, type in the lower two
__device__ __forceinline__ int32_t mpk(int ptr, int type) { return ((ptr + 1) << 2) | type; }

struct Fwd { int k, l, v, rmid; };

// the forward pass of align() (100-248) on A[1..M], B[1..N] inside [low, up]
template <class AR>
__device__ Fwd any_global_forward(const AR& a, const AnyLayout& Y, gbytes A, gbytes B, int M, int N, int low, int up, int tb, int te)
{
    const int g = kOpen, h = kExt, m = g + h;
    const int band = up - low + 1, midd = band / 2 + 1;
    Fwd o; o.rmid = low + midd - 1;
    int leftd = 1 - low, rightd = up - low + 1;
    int c = 0, d = 0, e = 0, IP = 0;
#define CC(i) a.row(Y.o_cc, (i))
#define DD(i) a.row(Y.o_dd, (i))
#define CP(i) a.row(Y.o_cp, (i))
#define DP(i) a.row(Y.o_dp, (i))
#define MP0(i) a.at(Y.o_mp0, (i))
#define MP1(i) a.at(Y.o_mp1, (i))
#define MP2(i) a.at(Y.o_mp2, (i))
    if (leftd < midd) {                                                     // 102-111
        for (int j = 0; j < midd; j++) CP(j) = DP(j) = -1;
        for (int j = midd; j <= rightd; j++) CP(j) = DP(j) = 0;
        MP0(0) = MP1(0) = MP2(0) = mpk(-1, 0);
    } else if (leftd > midd) {                                              // 112-121
        const int fr = leftd - midd;
        for (int j = 0; j <= midd; j++) CP(j) = DP(j) = fr;
        for (int j = midd + 1; j <= rightd; j++) CP(j) = DP(j) = -1;
        MP0(fr) = MP1(fr) = MP2(fr) = mpk(-1, 0);
    } else {                                                                // 122-133
        for (int j = 0; j <= rightd; j++) CP(j) = DP(j) = 0;
        MP0(0) = MP1(0) = MP2(0) = mpk(-1, 0);
    }
    CC(leftd) = 0;                                                          // 135-146
    int t = (tb == 2) ? 0 : -g;
    for (int j = leftd + 1; j <= rightd; j++) { CC(j) = t = t - h; DD(j) = t - g; }
    CC(rightd + 1) = kNegInf; DD(rightd + 1) = kNegInf;
    DD(leftd) = (tb == 1) ? 0 : -g;
    CC(leftd - 1) = kNegInf;
    for (int i = 1; i <= M; i++) {                                          // 147-234
        if (i > N - up) rightd--;
        if (leftd > 1) leftd--;
        const uint32_t ai = A[i];
        if ((c = CC(leftd + 1) - m) > (d = DD(leftd + 1) - h)) { d = c; DP(leftd) = CP(leftd + 1); }
        else DP(leftd) = DP(leftd + 1);
        const int ib = leftd + low - 1 + i;
        if (ib > 0) c = CC(leftd) + wsub(ai, B[ib]);
        if (d > c || ib <= 0) { c = d; CP(leftd) = DP(leftd); }
        e = c - g;
        DD(leftd) = d; CC(leftd) = c;
        IP = CP(leftd);
        if (leftd == midd) CP(leftd) = DP(leftd) = IP = i;
        for (int curd = leftd + 1; curd <= rightd; curd++) {
            if (curd != midd) {                                             // 166-188
                if ((c = c - m) > (e = e - h)) { e = c; IP = CP(curd - 1); }
                if ((c = CC(curd + 1) - m) > (d = DD(curd + 1) - h)) { d = c; DP(curd) = CP(curd + 1); }
                else DP(curd) = DP(curd + 1);
                c = CC(curd) + wsub(ai, B[curd + low - 1 + i]);
                if (c < d || c < e) {
                    if (e > d) { c = e; CP(curd) = IP; }
                    else       { c = d; CP(curd) = DP(curd); }
                }
                CC(curd) = c; DD(curd) = d;
            } else {                                                        // 189-232: on the middle diagonal
                int m1, m2, m0;
                if ((c = c - m) > (e = e - h)) { e = c; m1 = mpk(CP(curd - 1), 2); }
                else m1 = mpk(IP, 2);
                if ((c = CC(curd + 1) - m) > (d = DD(curd + 1) - h)) { d = c; m2 = mpk(CP(curd + 1), 1); }
                else m2 = mpk(DP(curd + 1), 1);
                c = CC(curd) + wsub(ai, B[curd + low - 1 + i]);
                if (c < d || c < e) {
                    if (e > d) { c = e; m0 = (m1 & ~3) | 2; }
                    else       { c = d; m0 = (m2 & ~3) | 1; }
                } else m0 = mpk(i - 1, 0);
                if (c - g > e) m1 = m0;
                if (c - g > d) m2 = m0;
                MP0(i) = m0; MP1(i) = m1; MP2(i) = m2;
                CP(curd) = DP(curd) = IP = i;
                CC(curd) = c; DD(curd) = d;
            }
        }
    }
    if (te == 1 && d + g > c)      { o.k = DP(rightd); o.l = 2; }           // 236-249
    else if (te == 2 && e + g > c) { o.k = IP;         o.l = 1; }
    else                           { o.k = CP(rightd); o.l = 0; }
    if (o.rmid > N - M) o.l = 2;
    else if (o.rmid < N - M) o.l = 1;
    o.v = c;
    return o;
}

// align()'s divide and conquer (66-307) as frames.  A0 / B0: 1-based views of the located sub-strings.  Returns 0 or a status.
template <class AR>
__device__ int any_global_align(const AR& a, const AnyLayout& Y, Script& S, gbytes A0, gbytes B0, int Ma, int Na, int low, int up, int* score_out)
{
    enum { F_AO, F_BO, F_M, F_N, F_LOW, F_UP, F_TB, F_TE, F_PHASE, F_K, F_L, F_KT, F_RMID };
    int sp = 0, top = 0, guard = 0; bool first = true;
#define FR(s, w) a.at(Y.o_stk, (s) * kFrameWords + (w))
#define FPW(i) a.at(Y.o_fp, (i))
    auto push = [&](int ao, int bo, int M, int N, int lo, int u, int tb, int te) {
        FR(sp, F_AO) = ao; FR(sp, F_BO) = bo; FR(sp, F_M) = M; FR(sp, F_N) = N; FR(sp, F_LOW) = lo; FR(sp, F_UP) = u;
        FR(sp, F_TB) = tb; FR(sp, F_TE) = te; FR(sp, F_PHASE) = 0;
        sp++;
    };
    push(0, 0, Ma, Na, low, up, 0, 0);
    while (sp > 0) {
        if (sp >= kFrames - 1) return IM_ST_OVERFLOW;
        if (++guard > (1 << 24)) return IM_ST_ABORT;                        // every frame makes progress; this only bounds a corrupted walk
        const int fs = sp - 1;
        const int ao = FR(fs, F_AO), bo = FR(fs, F_BO), M = FR(fs, F_M), N = FR(fs, F_N), lo = FR(fs, F_LOW), u = FR(fs, F_UP);
        const int tb = FR(fs, F_TB), te = FR(fs, F_TE), phase = FR(fs, F_PHASE);
        int k = FR(fs, F_K), l = FR(fs, F_L), kt = FR(fs, F_KT), rmid = FR(fs, F_RMID);
        gbytes A = A0 + ao; gbytes B = B0 + bo;
        int nphase = phase; bool pop = false, pushed = false;
        int c_ao = 0, c_bo = 0, c_M = 0, c_N = 0, c_lo = 0, c_u = 0, c_tb = 0, c_te = 0;
        auto child = [&](int xa, int xb, int xM, int xN, int xlo, int xu, int xtb, int xte) {
            pushed = true; c_ao = xa; c_bo = xb; c_M = xM; c_N = xN; c_lo = xlo; c_u = xu; c_tb = xtb; c_te = xte; };
        if (phase == 0) {
            if (N <= 0) { if (M > 0) s_del(a, S, M); pop = true; if (first) { top = -1; first = false; } }           // 82-85
            else if (M <= 0) { s_ins(a, S, N); pop = true; if (first) { top = -1; first = false; } }                 // 86-89
            else if (u - lo + 1 <= 1) { for (int i = 0; i < M; i++) s_rep(a, S, A0, B0); pop = true; if (first) { top = -1; first = false; } }   // 90-93
            else {
                if (u - lo + 1 > Y.maxB || M > Y.maxM) return IM_ST_OVERFLOW;
                const Fwd fo = any_global_forward(a, Y, A, B, M, N, lo, u, tb, te);
                if (first) { top = fo.v; first = false; }
                // the chain of crossing points, reversed into forward pointers (253-258)
                int kk = fo.k, ll = fo.l, r = -1;
                while (kk > -1) {
                    if (kk > M) return IM_ST_ABORT;
                    FPW(kk) = mpk(r, ll);
                    const int w = ll == 0 ? MP0(kk) : ll == 1 ? MP1(kk) : MP2(kk);
                    const int nk = (w >> 2) - 1, nl = w & 3;
                    if (nk >= kk) return IM_ST_ABORT;                       // crossing points strictly descend
                    r = kk; kk = nk; ll = nl;
                }
                if (r == -1) {                                              // never crossed the middle: same strings, half the band (260-262)
                    if (fo.rmid < 0) FR(fs, F_LOW) = fo.rmid + 1; else FR(fs, F_UP) = fo.rmid - 1;
                } else {
                    k = r; { const int w = FPW(k); l = (w >> 2) - 1; kt = w & 3; }
                    rmid = fo.rmid; FR(fs, F_RMID) = rmid;
                    if (rmid < 0) { nphase = 1; child(ao, bo, r - 1, r + rmid, rmid + 1, min(u, r + rmid), tb, 1); }             // 269-275
                    else if (rmid > 0) { nphase = 2; child(ao, bo, r, r + rmid - 1, max(-r, lo), rmid - 1, tb, 2); }
                    else nphase = 3;
                }
            }
        } else if (phase == 1) { s_del(a, S, 1); nphase = 3; }
        else if (phase == 2) { s_ins(a, S, 1); nphase = 3; }
        else if (phase == 4) { s_del(a, S, 1); k = l; { const int w = FPW(k); l = (w >> 2) - 1; kt = w & 3; } nphase = 3; }
        else if (phase == 5) { s_ins(a, S, 1); k = l; { const int w = FPW(k); l = (w >> 2) - 1; kt = w & 3; } nphase = 3; }
        else if (phase == 6) pop = true;
        if (nphase == 3 && !pushed && !pop) {
            const int t2 = u - rmid - 1, t3 = lo - rmid + 1;               // intermediate blocks (278-293)
            while (l > -1 && kt == 0) { s_rep(a, S, A0, B0); k = l; const int w = FPW(k); l = (w >> 2) - 1; kt = w & 3; }
            if (l > -1) {
                const int t1 = l - k - 1;
                if (kt == 1) { s_ins(a, S, 1); nphase = 4; child(ao + k, bo + k + rmid + 1, t1, t1, 0, min(t1, t2), 2, 1); }
                else         { s_del(a, S, 1); nphase = 5; child(ao + k + 1, bo + k + rmid, t1, t1, max(-t1, t3), 0, 1, 2); }
            } else {                                                        // last block (296-304)
                if (N - M > rmid) { s_ins(a, S, 1); const int t1 = k + rmid + 1; nphase = 6; child(ao + k, bo + t1, M - k, N - t1, 0, min(N - t1, t2), 2, te); }
                else if (N - M < rmid) { s_del(a, S, 1); const int t1 = M - (k + 1); nphase = 6; child(ao + k + 1, bo + k + rmid, t1, N - (k + rmid), max(-t1, t3), 0, 1, te); }
                else pop = true;
            }
        }
        FR(fs, F_PHASE) = nphase; FR(fs, F_K) = k; FR(fs, F_L) = l; FR(fs, F_KT) = kt;
        if (pop) sp--;
        if (pushed) push(c_ao, c_bo, c_M, c_N, c_lo, c_u, c_tb, c_te);
    }
    *score_out = top;
    return 0;
}

struct AnyAln { int st, r1, r2, q1, q2, n_ops; };

// attempt_band_alignment = local_align + ALIGN + fetch_cigar (src/alignment.c:343-391).  Window = contig[w0, w0 + N),
// piece = read[p0, p0 + M); the CIGAR lands at o_ops.
template <class AR>
__device__ AnyAln any_band_alignment(const AR& a, const AnyLayout& Y, gbytes contig, gbytes read, int p0, int M, int w0, int N,
                                     int low_in, int up_in, int32_t o_pos, int32_t o_ops)
{
    AnyAln r; r.st = 0; r.r1 = r.r2 = r.q1 = r.q2 = 0; r.n_ops = 0;
    if (low_in > up_in || M <= 0 || N <= 0) { r.st = IM_ST_ABORT; return r; }        // forceassert 359; strlen > 0, src/localalign.c:31-32
    gbytes A = read + p0 - 1;                                                // 1-based views (42-43)
    gbytes B = contig + w0 - 1;
    const int g = kOpen, h = kExt, m = g + h;
    const int low = max(-M, low_in), up = min(N, up_in);                              // 70-71
    const int band = up - low + 1;
    if (band < 1) { r.st = IM_ST_ABORT; return r; }                                   // the reference prints and exits (74-77)
    if (band > Y.maxB || M > Y.maxM) { r.st = IM_ST_OVERFLOW; return r; }
    int leftd, rightd, c, d, e = 0, ib;
    if (low > 0) leftd = 1; else if (up < 0) leftd = band; else leftd = 1 - low;      // 82-99
    rightd = band;
    const int si = max(0, -up), ei = min(M, N - low);
    CC(leftd) = 0;
    for (int j = leftd + 1; j <= rightd; j++) { CC(j) = 0; DD(j) = -g; }
    CC(rightd + 1) = kNegInf; DD(rightd + 1) = kNegInf;
    int best = 0, endi = si, endj = si + low, starti = 0, startj = 0;
    CC(leftd - 1) = kNegInf;
    DD(leftd) = -g;
    for (int i = si + 1; i <= ei; i++) {                                              // 100-131
        if (i > N - up) rightd--;
        if (leftd > 1) leftd--;
        const uint32_t ai = A[i];
        if ((c = CC(leftd + 1) - m) > (d = DD(leftd + 1) - h)) d = c;
        if ((ib = leftd + low - 1 + i) > 0) c = CC(leftd) + wsub(ai, B[ib]);
        if (d > c) c = d;
        if (c < 0) c = 0;
        e = c - g;
        DD(leftd) = d; CC(leftd) = c;
        if (c > best) { best = c; endi = i; endj = ib; }
        for (int curd = leftd + 1; curd <= rightd; curd++) {
            if ((c = c - m) > (e = e - h)) e = c;
            if ((c = CC(curd + 1) - m) > (d = DD(curd + 1) - h)) d = c;
            c = CC(curd) + wsub(ai, B[curd + low - 1 + i]);
            if (e > c) c = e;
            if (d > c) c = d;
            if (c < 0) c = 0;
            CC(curd) = c; DD(curd) = d;
            if (c > best) { best = c; endi = i; endj = curd + low - 1 + i; }
        }
    }
    leftd = max(1, -endi - low + 1);                                                  // 132-143
    rightd = band - (up - (endj - endi));
    CC(rightd) = 0;
    {
        int t = -g;
        for (int j = rightd - 1; j >= leftd; j--) { CC(j) = t = t - h; DD(j) = t - g; }
        for (int j = rightd + 1; j <= band; ++j) CC(j) = kNegInf;
    }
    CC(leftd - 1) = DD(leftd - 1) = kNegInf;
    DD(rightd) = -g;
    bool found = false;
    // The reverse pass has no `ib > 0` guard in the reference: a band that hangs off the window's left edge makes it compare read
    // bases with the bytes IN FRONT of the window -- the contig's own bytes when the window starts inside it (read as they are),
    // whatever lies in front of the contig otherwise (undefined there; here the zero padding every contig has in front, and zero
    // for anything further out: a read of 2600 bases against a window at the contig's start walks 2500 bytes out).
    auto before = [&](int j) -> uint32_t { const int64_t at = (int64_t)w0 + j - 1; return at >= -64 ? contig[at] : 0u; };
    for (int i = endi; i >= 1 && !found; i--) {                                       // 144-176
        if (i + low <= 0) leftd++;
        if (rightd < band) rightd++;
        const uint32_t ai = A[i];
        if ((c = CC(rightd - 1) - m) > (d = DD(rightd - 1) - h)) d = c;
        if ((ib = rightd + low - 1 + i) <= N) c = CC(rightd) + wsub(ai, before(ib));
        if (d > c) c = d;
        e = c - g;
        DD(rightd) = d; CC(rightd) = c;
        if (c == best) { starti = i; startj = ib; found = true; break; }
        for (int curd = rightd - 1; curd >= leftd; curd--) {
            if ((c = c - m) > (e = e - h)) e = c;
            if ((c = CC(curd - 1) - m) > (d = DD(curd - 1) - h)) d = c;
            c = CC(curd) + wsub(ai, before(curd + low - 1 + i));                      // no `ib > 0` guard here in the reference either
            if (e > c) c = e;
            if (d > c) c = d;
            CC(curd) = c; DD(curd) = d;
            if (c == best) { starti = i; startj = curd + low - 1 + i; found = true; break; }
        }
    }
    if (starti < 0 || starti > M || startj < 0 || startj > N) return r;              // 180-185
    if (endi - starti == 0 || endj - startj == 0) return r;                           // 191-193
    // ALIGN on the located sub-strings (333-401)
    const int Ma = endi - starti + 1, Na = endj - startj + 1;
    int lo2 = low - (startj - starti), up2 = up - (startj - starti);
    lo2 = min(max(-Ma, lo2), min(Na - Ma, 0));                                        // 347-348
    up2 = max(min(Na, up2), max(Na - Ma, 0));
    gbytes A0 = A + starti - 1;
    gbytes B0 = B + startj - 1;
    Script S; S.o_pos = o_pos; S.pos0 = starti - 1; S.ia = 0; S.jb = 0;
    for (int p = 0; p <= M; p++) a.at(o_pos, p) = kKindNone;
    int score = 0;
    if (up2 - lo2 + 1 <= 1) {                                                         // 358-365
        for (int i = 1; i <= Ma; i++) { score += wsub(A0[i], B0[i]); s_rep(a, S, A0, B0); }
    } else {
        // Equal lengths and at most three mismatches on the main diagonal: that alignment is the ONLY optimal one -- any other
        // path between the same corners holds at least one inserted and one deleted base (two gaps, 40 or more) and at most
        // Ma - 1 matches, so it scores below Ma - 41 < Ma - 11 * 3 -- and ALIGN, which returns an optimal alignment inside the
        // band (diagonal 0 always lies in it: lo2 <= 0 <= up2), can only come back with it: the divide and conquer is skipped.
        int mm = 4;
        if (Ma == Na) { mm = 0; for (int i = 1; i <= Ma && mm <= 3; i++) mm += A0[i] != B0[i] ? 1 : 0; }
        if (mm <= 3) {
            for (int i = 1; i <= Ma; i++) s_rep(a, S, A0, B0);
            score = Ma * kScoreMatch + mm * (kScoreMismatch - kScoreMatch);
        } else {
            const int st = any_global_align(a, Y, S, A0, B0, Ma, Na, lo2, up2, &score);
            if (st) { r.st = st; return r; }
        }
    }
    if (score <= 0) return r;                                                         // src/alignment.c:365-372
    // fetch_cigar (507-604): [AP S] runs [tail S]; deletion lengths count into the consumed total as there (541-595)
    {
        const int pos0 = starti - 1;
        int n = 0, numtotal = pos0, run = -1, numrun = 0;
        bool over = false;
        auto put = [&](int op, int len) { if (n >= IM_MAX_OPS) over = true; else a.atu(o_ops, n++) = ((uint32_t)len << 4) | (uint32_t)op; };
        auto flush = [&]() {
            if (numrun > 0) { put(run == kKindEq ? IM_OP_EQ : run == kKindX ? IM_OP_X : run == kKindI ? IM_OP_I : IM_OP_D, numrun); numtotal += numrun; }
            numrun = 0;
        };
        if (pos0 > 0) put(IM_OP_S, pos0);
        for (int p = 0; p <= Ma; p++) {
            const int w = a.at(o_pos, pos0 + p);
            const int dl = w >> 2, kd = w & 3;
            if (dl > 0) { if (run != 4) { flush(); run = 4; } numrun += dl; }
            if (p < Ma) { if (run != kd) { flush(); run = kd; } numrun++; }
        }
        flush();
        if (numtotal < M) put(IM_OP_S, M - numtotal);
        if (over) { r.st = IM_ST_OVERFLOW; return r; }
        r.n_ops = n;
    }
    r.r1 = startj + w0 - 1; r.r2 = endj + w0; r.q1 = starti + p0 - 1; r.q2 = endi + p0;    // 385-388
    return r;
}

#define OPS_OP(w)  ((int)((w) & 15u))
#define OPS_LEN(w) ((int)((w) >> 4))

// count_matches (src/alignment.c:219-303) for one candidate split
template <class AR>
__device__ int any_split_score(const AR& a, int32_t o1, int n1, int q2, int32_t o2, int n2, int q4, int* pmm)
{
    const int q3 = q2;
    int i, j, matches = 0, mm = 0;
    for (i = 0, j = 0; i < n1; i++) {
        const uint32_t w = a.atu(o1, i); const int len = OPS_LEN(w), op = OPS_OP(w);
        if (op != IM_OP_D) j += len;
        if (j < q2) { if (op == IM_OP_EQ) matches += len; else if (op == IM_OP_X) mm += len; }
        if (j >= q2) { if (op == IM_OP_EQ) matches += q2 - (j - len); else if (op == IM_OP_X) mm += q2 - (j - len); break; }
    }
    for (i = 0, j = 0; i < n2; i++) {
        const uint32_t w = a.atu(o2, i); const int len = OPS_LEN(w), op = OPS_OP(w);
        if (op != IM_OP_D) j += len;
        if (j >= q3) { if (op == IM_OP_EQ) matches += j - q3; else if (op == IM_OP_X) mm += j - q3; i += 1; break; }
    }
    for (; i < n2; i++) {
        const uint32_t w = a.atu(o2, i); const int len = OPS_LEN(w), op = OPS_OP(w);
        if (op != IM_OP_D) j += len;
        if (j < q4) { if (op == IM_OP_EQ) matches += len; else if (op == IM_OP_X) mm += len; }
        if (j >= q4) { if (op == IM_OP_EQ) matches += q4 - (j - len); else if (op == IM_OP_X) mm += q4 - (j - len); break; }
    }
    *pmm = mm;
    return matches;
}

// find_best_del_candidate (306-339)
template <class AR>
__device__ int any_best_split(const AR& a, int q1, int q2, int32_t o1, int n1, int q3, int q4, int32_t o2, int n2, int L, int* pindex)
{
    if (q1 != 0 || q3 > q2) return IM_ST_ABORT;
    int bestm = 0, bestmm = INT_MAX, index = -1;
    for (int i = q3; i <= q2; i++) {
        int mm; const int matches = any_split_score(a, o1, n1, i, o2, n2, q4, &mm);
        if (matches > L) return IM_ST_ABORT;
        if (matches > bestm || (matches == bestm && mm < bestmm)) { bestm = matches; bestmm = mm; index = i; }
        if (matches == L && mm == 0) break;
    }
    if (index == -1) return IM_ST_ABORT;
    *pindex = index;
    return 0;
}

// update_readsegs (src/readaln.c:348-458) into the arena's final list, then the evidence records and the result record
template <class AR>
__device__ int any_build_result(const AR& a, const AnyLayout& Y, im_read_result* out, const RealignArgs& R, int cidx,
                                int r1, int32_t o1, int n1, int index, int q2, int r2, int32_t o2, int n2)
{
    int n = 0, refindx = r1; bool over = false;
    auto emit = [&](int len, int op) {
        if (n >= IM_MAX_OPS) { over = true; return; }
        a.atu(Y.o_fin, n++) = ((uint32_t)len << 4) | (uint32_t)op;
        if (op == IM_OP_EQ || op == IM_OP_X || op == IM_OP_D || op == IM_OP_M) refindx += len;
    };
    int i, j;
    for (i = 0, j = 0; i < n1; i++) {                                                 // 362-385
        const uint32_t w = a.atu(o1, i); const int op = OPS_OP(w), len = OPS_LEN(w);
        if (len <= 0) return IM_ST_ABORT;
        if (op != IM_OP_D) j += len;
        if (j <= index) emit(len, op);
        if (j > index) { const int part = index - (j - len); if (part > 0) emit(part, op); break; }
    }
    int rindex = r2, nextindex = index;
    if (index >= q2) {                                                                // 389-412
        int offset = 0;
        for (i = 0, j = 0; i < n2; i++) {
            const uint32_t w = a.atu(o2, i); const int op = OPS_OP(w), len = OPS_LEN(w);
            if (op != IM_OP_D) j += len;
            if (j <= q2) { }
            else if (j > q2 && j <= index) { if (op != IM_OP_I) { offset += len; if ((j - len) <= q2) offset -= q2 - (j - len); } }
            else if (j > index) { if (op != IM_OP_I) { if ((j - len) <= index) offset += index - (j - len); } }
        }
        rindex = r2 + offset;
    } else { emit(q2 - index, IM_OP_I); nextindex += q2 - index; }                   // 413-421
    if (refindx < rindex) emit(rindex - refindx, IM_OP_D);                            // 424-430
    for (i = 0, j = 0; i < n2; i++) {                                                 // 432-446
        const uint32_t w = a.atu(o2, i); const int op = OPS_OP(w), len = OPS_LEN(w);
        if (op != IM_OP_D) j += len;
        if (j > nextindex) { emit(j - nextindex, op); i++; break; }
    }
    for (; i < n2; i++) { const uint32_t w = a.atu(o2, i); emit(OPS_LEN(w), OPS_OP(w)); }       // 448-453
    if (over) return IM_ST_OVERFLOW;
    // one evidence record per D / I segment (src/alignment.c:449-476, src/evidence.c:4-34)
    int ne = 0, refpos = r1, readpos = 0;
    for (int s = 0; s < n; s++) {
        const uint32_t w = a.atu(Y.o_fin, s); const int op = OPS_OP(w), len = OPS_LEN(w);
        out->ops[s] = w;
        if (op == IM_OP_D || op == IM_OP_I) {
            if (ne >= IM_MAX_EV) return IM_ST_OVERFLOW;
            im_evidence e;
            e.cls = op == IM_OP_D ? IM_CLS_DELETION : IM_CLS_INSERTION;
            e.b1 = refpos; e.b2 = op == IM_OP_D ? refpos + len : refpos;
            e.seg = s; e.read_off = readpos; e.lflank = e.rflank = e.nd_print = e.nd_filter = 0;
            for (int t = 0; t < n; t++) {
                if (t == s) continue;
                const uint32_t x = a.atu(Y.o_fin, t); const int o = OPS_OP(x), l = OPS_LEN(x);
                int32_t* flank = t < s ? &e.lflank : &e.rflank;
                if (o == IM_OP_EQ) *flank += l;
                else if (o == IM_OP_X || o == IM_OP_I) { *flank += l; e.nd_print += l; e.nd_filter += l; }
                else if (o == IM_OP_D) { e.nd_print += l; e.nd_filter += l; }
                else if (o == IM_OP_S) e.nd_filter += l;
                else return IM_ST_ABORT;
            }
            out->ev[ne++] = e;
        }
        if (op == IM_OP_EQ || op == IM_OP_X || op == IM_OP_D) refpos += len;
        if (op != IM_OP_D) readpos += len;
    }
    out->ref_start = r1; out->n_ops = n; out->n_ev = ne;
    if (ne > 0 && R.batch.ev_cls) {
        for (int k = 0; k < IM_MAX_EV; k++) {
            const int64_t sl = (int64_t)cidx * IM_MAX_EV + k;
            R.batch.ev_cls[sl] = k < ne ? out->ev[k].cls : -1;
            R.batch.ev_b1[sl] = k < ne ? out->ev[k].b1 : 0;
            R.batch.ev_b2[sl] = k < ne ? out->ev[k].b2 : 0;
        }
    }
    return ne > 0 ? IM_ST_EVIDENCE : IM_ST_NONE;
}

__device__ __forceinline__ void any_finish(im_read_result* out, int status, int n_band)
{
    out->status = status; out->n_band = n_band;
    if (status != IM_ST_EVIDENCE) { out->n_ev = 0; out->n_ops = 0; out->ref_start = 0; }
}

__device__ __forceinline__ void any_store_band(im_read_result* out, int which, const AnyBand& b, const AnyAln& x, int win, int piece)
{
    im_band_aln* o = &out->band[which];
    o->r1 = x.r1; o->r2 = x.r2; o->q1 = x.q1; o->q2 = x.q2; o->low = b.low; o->votes = b.votes; o->win_bytes = win; o->piece_bytes = piece;
}

// ---- the band search with the whole wave on ONE read ----------------------------------------------------------------
//
// One lane per read leaves the k-mer vote -- a probe of the read's table per window position, thousands per read -- a chain of
// dependent round trips to device memory with nothing to hide them (a batch of long reads is a few hundred waves: 7 of the
// 14 ms of a 2 x 300 library at -g 2, profiles/r04_f_any_probe.txt).  So the wave takes its lanes' searches one after the
// other, all 64 lanes on one read: the table (open addressing: key, offset, count per slot) and the diagonal histogram in
// LDS, lanes on contiguous stretches of the read's k-mers / the window / the diagonals.  Reads whose table does not fit the
// LDS share keep the search of their own lane (any_find_band).

struct CoopLds { IM_LDS_AS uint32_t* key; IM_LDS_AS int32_t* pos; IM_LDS_AS int32_t* cnt; IM_LDS_AS int32_t* diag; int32_t tab_slots, diag_words; };

__device__ __forceinline__ bool band_better(int c1, int d1, int i1, int c2, int d2, int i2)     // select_band's order (142-181)
{
    return c1 > c2 || (c1 == c2 && (d1 < d2 || (d1 == d2 && i1 < i2)));
}

// every argument wave-uniform; gdiag: the wave's own contiguous stretch of device memory for histograms beyond the LDS share
__device__ AnyBand coop_find_band(const CoopLds& T, IM_GLOBAL_AS int32_t* gdiag, gbytes ref, uint32_t z1, uint32_t e1, uint32_t anchor,
                                  gbytes read, uint32_t z2, uint32_t e2, uint32_t k, uint32_t g, int lane)
{
    AnyBand b; b.st = 0; b.low = b.up = 0; b.votes = 0;
    const uint32_t W = e1 - z1, L = e2 - z2;
    const uint32_t numdiag = (W - (k - 1)) + (L - (k - 1));                 // unsigned, as written (403-404)
    if (!(numdiag > g) || e2 < z2) { b.st = IM_ST_ABORT; return b; }       // forceasserts 405, 407
    if (L < k) { b.low = b.up = (int)(numdiag - 1); return b; }            // 408-412
    if ((int32_t)numdiag <= 0 || numdiag > (1u << 28)) { b.st = IM_ST_ABORT; return b; }
    const uint32_t mask = (1u << (2 * k)) - 1u, nread = L - k + 1;
    uint32_t H = 16; while (H < 2u * nread) H <<= 1;
    const uint32_t hm = H - 1u;
    const bool dlds = numdiag <= (uint32_t)T.diag_words;
    // the histogram's home: LDS, or the wave's stretch of the arena -- each with the accesses of its own address space
    auto dzero = [&](uint32_t i) { if (dlds) T.diag[i] = 0; else gdiag[i] = 0; };
    auto dvote = [&](uint32_t i) { if (dlds) __hip_atomic_fetch_add(&T.diag[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                   else __hip_atomic_fetch_add(&gdiag[i], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    for (uint32_t i = (uint32_t)lane; i < H; i += 64u) { T.key[i] = kEmpty; T.cnt[i] = 0; }
    for (uint32_t i = (uint32_t)lane; i < numdiag; i += 64u) dzero(i);
    if (!dlds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the zeros are in before any lane's vote
    wave_lds_sync();
    // the read piece's k-mers, a contiguous stretch per lane with a rolling code
    {
        const uint32_t per = (nread + 63u) / 64u, s0 = (uint32_t)lane * per, s1 = min(nread, s0 + per);
        uint32_t code = 0;
        if (s0 < s1) for (uint32_t u = 0; u + 1 < k; u++) code = (code << 2) | code2(read[z2 + s0 + u]);
        for (uint32_t q = s0; q < s1; q++) {
            code = ((code << 2) | code2(read[z2 + q + k - 1])) & mask;
            uint32_t h = (code * 2654435761u) >> 7 & hm;
            for (;;) {
                uint32_t old = kEmpty;
                __hip_atomic_compare_exchange_strong(&T.key[h], &old, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == kEmpty) { T.pos[h] = (int32_t)q; __hip_atomic_fetch_add(&T.cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                if (old == code) { __hip_atomic_fetch_add(&T.cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
                h = (h + 1) & hm;
            }
        }
    }
    wave_lds_sync();
    // every window position whose k-mer occurs exactly once in the piece votes for its diagonal (97-112)
    if (W >= k) {
        const uint32_t npos = W - k + 1, per = (npos + 63u) / 64u, s0 = (uint32_t)lane * per, s1 = min(npos, s0 + per);
        uint32_t code = 0;
        if (s0 < s1) for (uint32_t u = 0; u + 1 < k; u++) code = (code << 2) | code2(ref[z1 + s0 + u]);
        for (uint32_t p = s0; p < s1; p++) {
            code = ((code << 2) | code2(ref[z1 + p + k - 1])) & mask;
            uint32_t h = (code * 2654435761u) >> 7 & hm;
            for (;;) {
                const uint32_t key = T.key[h];
                if (key == kEmpty) break;
                if (key == code) {
                    if (T.cnt[h] == 1) {
                        const uint32_t indx = p - (uint32_t)T.pos[h] + L - k + 1;                              // 102-105
                        if (indx < numdiag) dvote(indx);
                    }
                    break;
                }
                h = (h + 1) & hm;
            }
        }
    }
    if (!dlds) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    wave_lds_sync();
    // bin_bands + select_band: a contiguous stretch of bands per lane with a running sum, then the wave's best
    const int anchor_rel = (int)(anchor - z1);
    int bc = -1, bd = INT_MAX, bi = INT_MAX;
    {
        auto dg = [&](uint32_t i) -> int { return dlds ? T.diag[i] : __hip_atomic_load(&gdiag[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        const uint32_t per = (numdiag + 63u) / 64u, i0 = (uint32_t)lane * per, i1 = min(numdiag, i0 + per);
        int run = 0;
        if (i0 < i1 && i0 < numdiag - g) for (uint32_t j = i0; j <= i0 + g; j++) run += dg(j);
        for (uint32_t i = i0; i < i1; i++) {
            const int bsum = i < numdiag - g ? run : 0;
            int d = (int)((uint32_t)anchor_rel - i); if (d < 0) d = -d;
            if (band_better(bsum, d, (int)i, bc, bd, bi)) { bc = bsum; bd = d; bi = (int)i; }
            if (i + g + 1 < numdiag) run += dg(i + g + 1) - dg(i);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int oc = __shfl_xor(bc, o), od = __shfl_xor(bd, o), oi = __shfl_xor(bi, o);
        if (band_better(oc, od, oi, bc, bd, bi)) { bc = oc; bd = od; bi = oi; }
    }
    wave_lds_sync();
    b.votes = bc;
    b.low = (int)((uint32_t)bi - (L - k + 1));                              // 438-439
    b.up = (int)((uint32_t)bi + g - (L - k + 1));
    return b;
}

// ---- attempt_pe_alignment (764-799) + attempt_diagonal_alignments (539-759) for one read, in three steps with the two band
// searches between them (the wave does those together where it can)

struct AnyRead {
    int c, L, tid, anchor;
    int left1, right1, left2, right2;
    int stage;                      // 0: done; 1: wants band search 1; 2: wants band search 2
    AnyBand b1, b2;
    int r1, r2, q1, q2, n1;
    uint32_t f, l, w0, w1, anc, p0, p1;
    bool want_tail;
};

__device__ void any_step_begin(const RealignArgs& R, int c, AnyRead& X)
{
    im_read_result* out = &R.batch.out[c];
    X.c = c; X.L = R.batch.read_len[c]; X.tid = R.batch.tid[c]; X.anchor = R.batch.anchor[c];
    const int range = R.batch.range_max[c], clen = R.ref.len[X.tid], anchor = X.anchor;
    int distance = range;                                                             // 774-783
    X.left1  = anchor >= distance ? anchor - distance : 0;
    X.right1 = clen < (anchor + distance) ? clen : anchor + distance;
    distance = range + (int)R.P.maxdelsize;
    X.left2  = anchor >= distance ? anchor - distance : 0;
    X.right2 = clen < (anchor + distance) ? clen : anchor + distance;
    X.stage = 1;
    if (!(anchor >= X.left1 && anchor >= X.left2 && anchor <= X.right1 && anchor <= X.right2 && X.left2 >= 0 && X.right2 > 0)) {
        any_finish(out, IM_ST_ABORT, 0); X.stage = 0;                                 // 548-553
    }
}

// behind band search 1: the first alignment, the geometric case, the second search's window and piece
template <class AR>
__device__ void any_step_middle(const AR& a, const AnyLayout& Y, const RealignArgs& R, AnyRead& X)
{
    im_read_result* out = &R.batch.out[X.c];
    const int L = X.L, anchor = X.anchor;
    const uint32_t eth = R.P.ethreshold;
    gbytes contig = (gbytes)(R.ref.ascii + R.ref.asc_off[X.tid]);
    gbytes read = (gbytes)(R.batch.bases + R.batch.base_off[X.c]);
    X.stage = 0;
    if (X.b1.st) { any_finish(out, X.b1.st, 0); return; }
    const AnyAln a1 = any_band_alignment(a, Y, contig, read, 0, L, X.left1, X.right1 - X.left1, X.b1.low, X.b1.up, Y.o_pos0, Y.o_ops0);
    any_store_band(out, 0, X.b1, a1, X.right1 - X.left1, L);
    if (a1.st) { any_finish(out, a1.st, 1); return; }
    const int r1 = a1.r1, r2 = a1.r2, q1 = a1.q1, q2 = a1.q2, n1 = a1.n_ops;
    X.r1 = r1; X.r2 = r2; X.q1 = q1; X.q2 = q2; X.n1 = n1;
    if (q1 == q2) { any_finish(out, IM_ST_NONE, 1); return; }                         // 568-572
    if (q1 == 0 && q2 == L) {                                                         // 575-582
        any_finish(out, any_build_result(a, Y, out, R, X.c, r1, Y.o_ops0, n1, L, 0, -1, Y.o_ops1, 0), 1);
        return;
    }
    // leading / trailing '=' runs of the first CIGAR (585-599)
    uint32_t f = 0, l = 0;
    {
        int j = 0;
        for (int i = 0; i < n1; i++) { const uint32_t w = a.atu(Y.o_ops0, i); if (i == 0 && OPS_OP(w) == IM_OP_S) continue; if (OPS_OP(w) != IM_OP_EQ) break; j += OPS_LEN(w); }
        f = (uint32_t)j;
        j = 0;
        for (int i = n1 - 1; i >= 0; i--) { const uint32_t w = a.atu(Y.o_ops0, i); if (i == n1 - 1 && OPS_OP(w) == IM_OP_S) continue; if (OPS_OP(w) != IM_OP_EQ) break; j += OPS_LEN(w); }
        l = (uint32_t)j;
    }
    X.f = f; X.l = l;
    // piece 2 by the four geometric cases; the guards in unsigned arithmetic as written (605-717)
    const uint32_t uL = (uint32_t)L;
    const int right2 = X.right2, left2 = X.left2;
    uint32_t w0, w1, anc, p0, p1; bool want_tail;
    if (r1 > anchor) {
        if (q1 == 0) {
            if (!(uL > f)) { any_finish(out, IM_ST_ABORT, 1); return; }
            if ((uL - f) < eth || ((uint32_t)right2 - (uint32_t)r1 - f) < eth) { any_finish(out, IM_ST_NONE, 1); return; }
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)right2; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
        } else if (q2 == L) {
            if (!(uL > l)) { any_finish(out, IM_ST_ABORT, 1); return; }
            if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)anchor) < eth) { any_finish(out, IM_ST_NONE, 1); return; }
            w0 = (uint32_t)anchor; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l; want_tail = false;
        } else { any_finish(out, IM_ST_NONE, 1); return; }
    } else if (r1 < anchor) {
        if (r2 >= anchor) { any_finish(out, IM_ST_NONE, 1); return; }
        if (q1 == 0) {
            if (!(uL > f)) { any_finish(out, IM_ST_ABORT, 1); return; }
            if ((uL - f) < eth || ((uint32_t)anchor - (uint32_t)r1 - f) < eth) { any_finish(out, IM_ST_NONE, 1); return; }
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)anchor; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
        } else if (q2 == L) {
            if (!(uL > l)) { any_finish(out, IM_ST_ABORT, 1); return; }
            if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)left2) < eth) { any_finish(out, IM_ST_NONE, 1); return; }
            w0 = (uint32_t)left2; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l; want_tail = false;
        } else { any_finish(out, IM_ST_NONE, 1); return; }
    } else { any_finish(out, IM_ST_NONE, 1); return; }                                // r1 == anchor (712-717)
    if ((int32_t)(w1 - w0) <= 0) { any_finish(out, IM_ST_ABORT, 1); return; }
    X.w0 = w0; X.w1 = w1; X.anc = anc; X.p0 = p0; X.p1 = p1; X.want_tail = want_tail;
    X.stage = 2;
}

// behind band search 2: the second alignment and the merge of the two pieces
template <class AR>
__device__ void any_step_end(const AR& a, const AnyLayout& Y, const RealignArgs& R, AnyRead& X)
{
    im_read_result* out = &R.batch.out[X.c];
    const int L = X.L, c = X.c;
    gbytes contig = (gbytes)(R.ref.ascii + R.ref.asc_off[X.tid]);
    gbytes read = (gbytes)(R.batch.bases + R.batch.base_off[c]);
    const int r1 = X.r1, r2 = X.r2, q1 = X.q1, q2 = X.q2, n1 = X.n1;
    const uint32_t f = X.f, l = X.l, w0 = X.w0, w1 = X.w1, p0 = X.p0, p1 = X.p1;
    const bool want_tail = X.want_tail;
    X.stage = 0;
    if (X.b2.st) { any_finish(out, X.b2.st, 1); return; }
    const AnyAln a2 = any_band_alignment(a, Y, contig, read, (int)p0, (int)(p1 - p0), (int)w0, (int)(w1 - w0), X.b2.low, X.b2.up, Y.o_pos1, Y.o_ops1);
    any_store_band(out, 1, X.b2, a2, (int)(w1 - w0), (int)(p1 - p0));
    if (a2.st) { any_finish(out, a2.st, 2); return; }
    const int r3 = a2.r1, r4 = a2.r2, q3 = a2.q1, q4 = a2.q2; int n2 = a2.n_ops;
    if (want_tail) { if (q4 != L || q3 == q4) { any_finish(out, IM_ST_NONE, 2); return; } }
    else           { if (q3 != 0 || q3 == q4) { any_finish(out, IM_ST_NONE, 2); return; } }
    // add_prefix_soft_clip / add_suffix_soft_clip (478-532)
    if (want_tail && f > 0) {
        const uint32_t w = a.atu(Y.o_ops1, 0);
        if (n2 > 0 && OPS_OP(w) == IM_OP_S) a.atu(Y.o_ops1, 0) = ((uint32_t)(OPS_LEN(w) + (int)f) << 4) | IM_OP_S;
        else if (n2 >= IM_MAX_OPS) { any_finish(out, IM_ST_OVERFLOW, 2); return; }
        else { for (int i = n2; i > 0; i--) a.atu(Y.o_ops1, i) = a.atu(Y.o_ops1, i - 1); a.atu(Y.o_ops1, 0) = (f << 4) | IM_OP_S; n2++; }
    } else if (!want_tail && l > 0) {
        if (n2 <= 0) { any_finish(out, IM_ST_ABORT, 2); return; }
        const uint32_t w = a.atu(Y.o_ops1, n2 - 1);
        if (OPS_OP(w) == IM_OP_S) a.atu(Y.o_ops1, n2 - 1) = ((uint32_t)(OPS_LEN(w) + (int)l) << 4) | IM_OP_S;
        else if (n2 >= IM_MAX_OPS) { any_finish(out, IM_ST_OVERFLOW, 2); return; }
        else { a.atu(Y.o_ops1, n2) = (l << 4) | IM_OP_S; n2++; }
    }
    if (!(q1 < q2 && q3 < q4)) { any_finish(out, IM_ST_ABORT, 2); return; }           // 720-721
    int st, index = -1;
    if (q1 > q3 && q1 <= q4) {                                                        // 724-731
        st = any_best_split(a, q3, q4, Y.o_ops1, n2, q1, q2, Y.o_ops0, n1, L, &index);
        if (st == 0) st = any_build_result(a, Y, out, R, c, r3, Y.o_ops1, n2, index, q1, r1, Y.o_ops0, n1);
    } else if (q3 > q1 && q3 <= q2) {                                                 // 732-739
        st = any_best_split(a, q1, q2, Y.o_ops0, n1, q3, q4, Y.o_ops1, n2, L, &index);
        if (st == 0) st = any_build_result(a, Y, out, R, c, r1, Y.o_ops0, n1, index, q3, r3, Y.o_ops1, n2);
    } else if (q1 > q4 && r1 == r4) st = any_build_result(a, Y, out, R, c, r3, Y.o_ops1, n2, q4, q1, r1, Y.o_ops0, n1);       // 740-744
    else if (q3 > q2 && r2 == r3) st = any_build_result(a, Y, out, R, c, r1, Y.o_ops0, n1, q2, q3, r3, Y.o_ops1, n2);         // 745-749
    else st = IM_ST_NONE;
    any_finish(out, st, 2);
}

// One band search for every lane whose read is at `stage`: the wave together, lane by lane, where the read's table fits the
// LDS share; the lane alone otherwise.
template <class AR>
__device__ void any_search_round(const AR& a, const AnyLayout& Y, const RealignArgs& R, const CoopLds& T, IM_GLOBAL_AS int32_t* gdiag, AnyRead& X, int stage, int lane)
{
    const uint32_t k = R.P.klength, g = R.P.numgaps;
    const bool mine = X.stage == stage;
    // the piece and the window of this lane's search
    const uint32_t z1 = stage == 1 ? (uint32_t)X.left1 : X.w0, e1 = stage == 1 ? (uint32_t)X.right1 : X.w1;
    const uint32_t anc = stage == 1 ? (uint32_t)X.anchor : X.anc;
    const uint32_t z2 = stage == 1 ? 0u : X.p0, e2 = stage == 1 ? (uint32_t)X.L : X.p1;
    const uint32_t piece = e2 - z2;
    const bool coop = mine && T.tab_slots > 0 && e2 >= z2 && piece >= k && 2ull * (piece - k + 1) <= (unsigned long long)T.tab_slots;
    uint64_t m = __ballot(coop);
    while (m) {
        const int r = (int)__builtin_ctzll(m);
        m &= m - 1ull;
        const int tid = __shfl(X.tid, r), c = __shfl(X.c, r);
        gbytes contig = (gbytes)(R.ref.ascii + R.ref.asc_off[tid]);
        gbytes read = (gbytes)(R.batch.bases + R.batch.base_off[c]);
        const AnyBand b = coop_find_band(T, gdiag, contig, (uint32_t)__shfl((int)z1, r), (uint32_t)__shfl((int)e1, r), (uint32_t)__shfl((int)anc, r),
                                         read, (uint32_t)__shfl((int)z2, r), (uint32_t)__shfl((int)e2, r), k, g, lane);
        if (lane == r) { if (stage == 1) X.b1 = b; else X.b2 = b; }
    }
    if (mine && !coop) {
        gbytes contig = (gbytes)(R.ref.ascii + R.ref.asc_off[X.tid]);
        gbytes read = (gbytes)(R.batch.bases + R.batch.base_off[X.c]);
        const AnyBand b = any_find_band(a, Y, contig, z1, e1, anc, read, z2, e2, k, g);
        if (stage == 1) X.b1 = b; else X.b2 = b;
    }
}

// counters: [0] reads listed, [1] longest read among them, [2] widest window, [3] next list entry to claim
__global__ __launch_bounds__(256) void any_pick_kernel(RealignArgs A, int all, int32_t* list, int32_t* counters)
{
    const int n = A.n_dev ? min(*A.n_dev, A.batch.n) : A.batch.n;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n; c += gridDim.x * blockDim.x) {
        im_read_result* out = &A.batch.out[c];
        const int64_t off = A.batch.base_off[c];
        const int L = A.batch.read_len[c], tid = A.batch.tid[c];
        if (all) {
            // no other kernel has seen the batch: what they leave in every record
            uint32_t* bw = reinterpret_cast<uint32_t*>(&out->band[0]);
            for (int i = 0; i < 16; i++) bw[i] = 0u;
            for (int i = 0; i < 7; i++) out->reserved[i] = 0;
            if (!A.keep_slots && A.batch.ev_cls)
                for (int k = 0; k < IM_MAX_EV; k++) { const int64_t sl = (int64_t)c * IM_MAX_EV + k; A.batch.ev_cls[sl] = -1; A.batch.ev_b1[sl] = 0; A.batch.ev_b2[sl] = 0; }
            const bool bad = L <= 0 || tid < 0 || tid >= A.ref.n_contigs;
            any_finish(out, (off & 3) ? IM_ST_UNSUPPORTED : bad ? IM_ST_ABORT : IM_ST_UNSUPPORTED, 0);
            if (bad) continue;
        }
        if (out->status != IM_ST_UNSUPPORTED || (off & 3) || L <= 0 || tid < 0 || tid >= A.ref.n_contigs) continue;
        const int clen = A.ref.len[tid], anchor = A.batch.anchor[c];
        const int64_t reach = (int64_t)A.batch.range_max[c] + (int64_t)A.P.maxdelsize;
        const int64_t lo = anchor >= reach ? anchor - reach : 0, hi = (int64_t)anchor + reach > clen ? clen : anchor + reach;
        const int W = (int)(hi > lo ? hi - lo : 0);
        list[atomicAdd(&counters[0], 1)] = c;
        atomicMax(&counters[1], L);
        atomicMax(&counters[2], W);
    }
}

// LDSROWS: the four rows of the banded passes live in LDS (the launch decides: make_layout's rows_in_lds)
template <bool LDSROWS>
__global__ __launch_bounds__(64) void realign_any_kernel(RealignArgs A, const int32_t* list, int32_t* counters, int32_t* arena, AnyLayout Y)
{
    extern __shared__ int32_t s_dyn[];
    const int lane = threadIdx.x;
    const int R = 1 << Y.lane_shift;                    // lanes that hold a read; the others only take part in the band searches
    IM_GLOBAL_AS int32_t* wave_arena = (IM_GLOBAL_AS int32_t*)arena + ((int64_t)blockIdx.x * Y.words << Y.lane_shift);
    IM_LDS_AS int32_t* lds = (IM_LDS_AS int32_t*)s_dyn;
    ArT<LDSROWS> a; a.sh = Y.lane_shift; a.p = wave_arena + (lane & (R - 1));
    if constexpr (LDSROWS) a.q = lds + (lane & (R - 1)); else a.q = a.p;
    CoopLds T;
    T.tab_slots = Y.l_tab_slots; T.diag_words = Y.l_diag_words;
    T.key = (IM_LDS_AS uint32_t*)(lds + Y.l_rows_words);
    T.pos = lds + Y.l_rows_words + Y.l_tab_slots;
    T.cnt = T.pos + Y.l_tab_slots;
    T.diag = T.cnt + Y.l_tab_slots;
    // the lanes' histogram words of the arena are one contiguous stretch: the wave's own when it votes together
    IM_GLOBAL_AS int32_t* gdiag = wave_arena + ((int64_t)Y.o_diag << Y.lane_shift);
    const int n = counters[0];
    if (Y.maxB > kCoopMaxBand) {
        // wide bands: the row-by-row dynamic programs are the pass and differ widely from read to read -- every lane on its own,
        // claiming its next read when it is done with one (waiting for each other at the steps cost 64 -> 76 ms at -g 61)
        const uint32_t k = A.P.klength, g = A.P.numgaps;
        if (lane >= R) return;
        for (;;) {
            const int at = atomicAdd(&counters[3], 1);
            if (at >= n) break;
            AnyRead X;
            any_step_begin(A, list[at], X);
            if (X.stage != 1) continue;
            gbytes contig = (gbytes)(A.ref.ascii + A.ref.asc_off[X.tid]);
            gbytes read = (gbytes)(A.batch.bases + A.batch.base_off[X.c]);
            X.b1 = any_find_band(a, Y, contig, (uint32_t)X.left1, (uint32_t)X.right1, (uint32_t)X.anchor, read, 0u, (uint32_t)X.L, k, g);
            any_step_middle(a, Y, A, X);
            if (X.stage != 2) continue;
            X.b2 = any_find_band(a, Y, contig, X.w0, X.w1, X.anc, read, X.p0, X.p1, k, g);
            any_step_end(a, Y, A, X);
        }
        return;
    }
    // narrow bands: the band searches are the pass.  A lane claims a read per round; the wave moves through the three steps
    // together so that the searches in between can be done by all lanes on one read
    for (;;) {
        const int at = lane < R ? atomicAdd(&counters[3], 1) : n;
        const bool have = at < n;
        if (__ballot(have) == 0ull) break;
        AnyRead X; X.stage = 0; X.c = 0; X.tid = 0; X.L = 0; X.anchor = 0; X.left1 = X.right1 = 0;
        X.w0 = X.w1 = X.anc = X.p0 = X.p1 = 0;
        if (have) any_step_begin(A, list[at], X);
        any_search_round(a, Y, A, T, gdiag, X, 1, lane);
#if defined(IM_ANY_STOP) && IM_ANY_STOP == 1
        continue;
#endif
        if (X.stage == 1) any_step_middle(a, Y, A, X);
#if defined(IM_ANY_STOP) && IM_ANY_STOP == 2
        continue;
#endif
        any_search_round(a, Y, A, T, gdiag, X, 2, lane);
#if defined(IM_ANY_STOP) && IM_ANY_STOP == 3
        continue;
#endif
        if (X.stage == 2) any_step_end(a, Y, A, X);
    }
}

}  // namespace

size_t realign_any_arena_bytes(int32_t max_read, int32_t max_window, uint32_t numgaps, int32_t lane_shift, int32_t n_waves)
{
    const AnyLayout y = make_layout(max_read, max_window, (int32_t)numgaps + 1, lane_shift);
    return ((size_t)y.words << lane_shift) * sizeof(int32_t) * (size_t)n_waves;
}

hipError_t launch_realign_any_pick(const RealignArgs& a, int all, int32_t* list, int32_t* counters, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), stream);
    if (e != hipSuccess) return e;
    if (a.batch.n <= 0) return hipSuccess;
    int blocks = (a.batch.n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(any_pick_kernel, dim3(blocks), dim3(256), 0, stream, a, all, list, counters);
    return hipGetLastError();
}

hipError_t launch_realign_any(const RealignArgs& a, const int32_t* list, int32_t* counters, int32_t* arena,
                              int32_t max_read, int32_t max_window, int32_t lane_shift, int32_t n_waves, hipStream_t stream)
{
    const AnyLayout y = make_layout(max_read, max_window, (int32_t)a.P.numgaps + 1, lane_shift);
    const size_t lds = (size_t)y.lds_bytes;
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(realign_any_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 << 10);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(realign_any_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 << 10);
        if (e != hipSuccess) return e;
        attr_bytes = (size_t)144 << 10;
    }
    if (y.rows_in_lds) hipLaunchKernelGGL(realign_any_kernel<true>, dim3(n_waves), dim3(64), lds, stream, a, list, counters, arena, y);
    else hipLaunchKernelGGL(realign_any_kernel<false>, dim3(n_waves), dim3(64), lds, stream, a, list, counters, arena, y);
    return hipGetLastError();
}

}  // namespace im
