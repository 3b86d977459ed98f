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
// XCD's L2.  Per-wave LDS: a 4 KiB read k-mer table, a 1.9 KiB packed-byte
// diagonal histogram and the read (6.2 KiB, 24 waves per CU).  No MFMA: this
// is integer scan / histogram work; measured, it is bound by VALU issue
// (profiles/README.md), so the design rule is fewest vector instructions per read.
//
// Lane layout for everything positional: lane l owns read positions
// 4l..4l+3 (reads of up to kShortRead = 255 bases; longer ones are im_realign_long.hip's); prefix sums / minima
// run as lane-local 4-step chains plus one 64-lane scan.

#include "im_device.hpp"
#include "im_wave.hpp"

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
#elif defined(IM_STOP_AFTER)
// Diagnostic build only (profiles/phase_counts.sh): every read stops after phase IM_STOP_AFTER, so the
// difference of SQ_INSTS_* between consecutive builds is that phase's dynamic instruction count.
#define IM_STAMP_DECL
#define IM_STAMP(id) do { if ((id) == IM_STOP_AFTER) { finish(out, IM_ST_NONE, 0, lane); return; } } while (0)
#define IM_STAMP_ARG , int stamp_base_
#define IM_STAMP_PASS(base) , base
#define IM_STAMP_B(id) do { if (stamp_base_ + (id) == IM_STOP_AFTER) { b.st = IM_ST_ABORT; return b; } } while (0)
#else
#define IM_STAMP_DECL
#define IM_STAMP(id)
#define IM_STAMP_ARG
#define IM_STAMP_PASS(base)
#define IM_STAMP_B(id)
#endif

namespace im {
namespace {

#ifndef IM_DIAG_CHUNK
#define IM_DIAG_CHUNK 1920
#endif
#ifndef IM_WAVES_PER_SIMD
#define IM_WAVES_PER_SIMD 6
#endif
#ifndef IM_BLOCKS_PER_CU
#define IM_BLOCKS_PER_CU 4096
#endif
constexpr int kDiagChunk = IM_DIAG_CHUNK;   // diagonals per histogram pass (1 byte each)
constexpr int kTblBytes  = 4096;            // 4^6 direct table, or 512-slot hash (keys+vals)
constexpr int kHashSlots = 512;
constexpr int kDirectMaxK = 6;

constexpr int kDiagWords = kDiagChunk / 4 + 16;     // packed byte counters (+ slack for band sums)
static_assert(kDiagWords % 4 == 0, "the histogram is cleared with 16-byte stores");

struct WaveLds {
    alignas(16) uint32_t diag[kDiagWords];
    alignas(16) uint32_t tbl[kTblBytes / 4];
    uint32_t rd[(256 + 16) / 4];            // read bases, read coordinates
#ifdef IM_LDS_PAD                           // diagnostic builds only (occupancy experiments)
    uint32_t pad[IM_LDS_PAD / 4];
#endif
};

// ---- K1: band search ---------------------------------------------------------

struct Band {
    int st;             // 0 ok, IM_ST_ABORT
    int low;            // diagonal handed to the scan
    int votes;
    int win, piece;
};

// the read k-mer table: 1 + offset of the k-mer in the piece, for k-mers that occur exactly once
// (bin_diagonals only lets read-unique k-mers vote, 97-98).  In the direct (k <= 6) table every
// other entry is 0, so "votes" is simply "non-zero"; the hash table (k > 6) marks repeats 0xFF.
template <int KT, bool DIRECT>
__device__ __forceinline__ void table_build(WaveLds& s, uint32_t p0, uint32_t nq, uint32_t k, int lane, uint32_t read_pk8, uint32_t* codes)
{
    if constexpr (KT == 6 || KT == -1) {
        // direct table (k = 6, the reference default, with its mask a constant; KT = -1: any k <= 6): cleared with four
        // 16-byte stores per lane -- cheaper in vector instructions than un-doing the entries after the vote, and vector
        // issue is what binds the kernel.  The read's 2-bit codes travel as one packed byte per lane
        // (bases 4l..4l+3); the two following lanes' bytes come over DPP, and the lane's four
        // 6-mers are bit fields of that 24-bit window.
        {
            uint4* t4 = reinterpret_cast<uint4*>(s.tbl);
#pragma unroll
            for (int i = 0; i < kTblBytes / 16 / 64; i++) t4[lane + 64 * i] = make_uint4(0u, 0u, 0u, 0u);
            wave_lds_sync();
        }
        uint8_t* t8 = reinterpret_cast<uint8_t*>(s.tbl);
        const uint32_t n1 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)read_pk8);
        const uint32_t n2 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n1);
        const uint32_t w24 = read_pk8 | (n1 << 8) | (n2 << 16);
        uint32_t code[4]; bool have[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = 4u * lane + j;
            have[j] = x >= p0 && x < p0 + nq;
            code[j] = (w24 >> (2 * j)) & (KT == 6 ? 0xFFFu : ((1u << (2 * k)) - 1u));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) if (have[j]) t8[code[j]] = (uint8_t)(4u * lane + j - p0 + 1u);
        wave_lds_sync();
        bool lost[4];
#pragma unroll
        for (int j = 0; j < 4; j++) lost[j] = have[j] && (t8[code[j]] != (uint8_t)(4u * lane + j - p0 + 1u));
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 4; j++) if (lost[j]) t8[code[j]] = 0u;      // repeated: nobody votes with it
        wave_lds_sync();
        return;
    }
    if constexpr (KT == 7) {
        // k = 7..13 ("prefix table"): the 4 KiB byte table goes by the k-mer's FIRST SIX bases and names the read position (+ 1);
        // the whole k-mer of every read position waits in codes[] for the vote to check.  Entries that meet an occupied slot take
        // the next one; a k-mer that meets ITSELF there occurs twice in the read: the entry is flagged, neither copy votes.
        uint8_t* t8 = reinterpret_cast<uint8_t*>(s.tbl);
        {
            uint4* t4 = reinterpret_cast<uint4*>(s.tbl);
#pragma unroll
            for (int i = 0; i < kTblBytes / 16 / 64; i++) t4[lane + 64 * i] = make_uint4(0u, 0u, 0u, 0u);
        }
        // bases 4l .. 4l+18 of the read as 2-bit codes: this lane's byte and the four lanes' behind it
        const uint32_t n1 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)read_pk8);
        const uint32_t n2 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n1);
        const uint32_t n3 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n2);
        const uint32_t n4 = (uint32_t)dpp_mov<kDppWaveShl1>(0, (int)n3);
        const uint32_t lo = read_pk8 | (n1 << 8) | (n2 << 16) | (n3 << 24);
        const uint32_t kmask = (1u << (2 * k)) - 1u;
        uint32_t code[4], slot[4]; bool pend[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = 4u * lane + j;
            pend[j] = x >= p0 && x < p0 + nq;
            code[j] = __builtin_amdgcn_alignbit(n4, lo, 2u * j) & kmask;          // k <= 13: 26 bits
            slot[j] = code[j] & 0xFFFu;
            if (pend[j]) codes[x - p0] = code[j];
        }
        wave_lds_sync();
        for (;;) {
            uint32_t e[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { e[j] = pend[j] ? lds_byte(s.tbl, slot[j]) : 1u; }
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 4; j++) if (pend[j] && e[j] == 0u) t8[slot[j]] = (uint8_t)(4u * lane + j - p0 + 1u);
            wave_lds_sync();
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!pend[j]) continue;
                const uint32_t me = 4u * lane + j - p0 + 1u;
                const uint32_t now = lds_byte(s.tbl, slot[j]);
                if (now == me) { pend[j] = false; continue; }                       // in
                const uint32_t other = codes[now - 1u];
                if ((other & 0x7FFFFFFFu) == code[j]) {                             // the same k-mer twice in the read: nobody votes with it
                    atomicOr(&codes[now - 1u], 0x80000000u);
                    pend[j] = false;
                    continue;
                }
                slot[j] = (slot[j] + 1u) & 0xFFFu;                                   // a neighbour's: the next slot
                any = true;
            }
            wave_lds_sync();
            if (!__builtin_amdgcn_ballot_w64(any)) break;
        }
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
            for (uint32_t u = 0; u < k; u++) c |= code2(lds_byte(s.rd, p0 + q + u)) << (2u * u);
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
        for (int j = 0; j < 4; j++) if (lost[j]) t8[code[j]] = 0u;
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

// One vote of the direct path: count the diagonal, and hand back select_band's order for it as ONE key, largest wins --
// this diagonal's count after the vote, then nearest the anchor, then smallest index.  The vote that lifts a diagonal to
// its final count records that diagonal's key, so the largest key any lane saw names the chunk's band: no pass over the
// histogram afterwards.  With the anchor clamped into the chunk (ac, band_search) nearness is 2047 - |ac - off|, and
//   key = (count << 22) | (near << 11) | (2047 - off) = (count << 22) + Q0 - ((|ac - off| << 11) + off),   Q0 = 2047 * 2049:
// one v_sad_u32, two shift-adds and a subtraction (Q0 also carries the + 1 of the count).
__device__ __forceinline__ uint32_t vote_direct(WaveLds& s, uint32_t off, uint32_t ac, uint32_t Q0, uint32_t mx)
{
    const uint32_t bsel = (off & 3u) * 8u;
    const uint32_t old = atomicAdd(&s.diag[off >> 2], 1u << bsel);
#ifdef IM_VOTE_DOUBLE_ATOMIC                                        // diagnostic builds only: what one more atomic of the same address pattern costs
    { uint32_t z = 0u; asm volatile("" : "+v"(z)); const uint32_t o2 = atomicAdd(&s.diag[off >> 2], z); asm volatile("" :: "v"(o2)); }
#endif
    const uint32_t cnt = __builtin_amdgcn_ubfe(old, off << 3, 8u);  // v_bfe_u32 takes the field offset modulo 32: (off & 3) * 8 without the mask
    uint32_t dist;                                                  // |off - ac|, one instruction
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(dist) : "v"(off), "s"(ac));
    const uint32_t u = Q0 - ((dist << 11) + off);                   // shift-add, subtract, shift-add: Q0 arrives in a register (band_search)
    return max(mx, (cnt << 22) + u);
}

// One unit of the vote (k <= 6, direct table): the 512 window starts whose packed dwords are in cur[],
// eight per lane.  Loads the next unit's dwords into nxt[] first (cur and nxt swap roles unit by
// unit, so nothing is copied).  `check`: the unit holds starts past the window's last k-mer, or the window
// has several chunks -- every vote is then tested.  Returns the largest key any of this lane's votes produced.
template <int KT>
__device__ __forceinline__ uint32_t vote_unit_direct(WaveLds& s, const uint32_t (&cur)[8], uint32_t (&nxt)[8],
                                                     const uint8_t* __restrict__ src_next, bool more,
                                                     uint32_t bsh, uint32_t kmask, uint32_t i0, uint32_t span,
                                                     uint32_t obase, bool check, uint32_t off_limit, uint32_t ac, uint32_t Q0, uint32_t mx,
                                                     const uint32_t* codes)
{
    // all eight table reads first, then the (rare) hits
    uint32_t v[8];
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0): the current dwords, loaded a unit ago -- one wait instead of eight counted ones
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (cur[j] >> bsh) & (KT == 6 || KT == 7 ? 0xFFFu : kmask);
    // The next unit's loads go out BEHIND the uses of the current dwords (the empty asm pins that): the
    // counted waits in front of those uses are merged over both values of `more`, and with the new loads already issued
    // the merged count makes every use wait for them too -- a memory round trip per unit instead of an overlap.
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
#ifndef IM_VOTE_UNCOND_LOAD
    if (more)
#endif
#pragma unroll
    for (int j = 0; j < 8; j++) nxt[j] = load_u32_unaligned(src_next + 16 * j);
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = lds_byte(s.tbl, v[j]);
    // pack the table bytes four to a word (v_perm), flag the non-zero bytes (bit 7 of each), merge the
    // two flag words into one mask (bit 8*jj + h for byte jj of word h) and walk only its bits
    const uint32_t xa = __builtin_amdgcn_perm(__builtin_amdgcn_perm(v[3], v[2], 0x0C0C0400u), __builtin_amdgcn_perm(v[1], v[0], 0x0C0C0400u), 0x05040100u);
    const uint32_t xb = __builtin_amdgcn_perm(__builtin_amdgcn_perm(v[7], v[6], 0x0C0C0400u), __builtin_amdgcn_perm(v[5], v[4], 0x0C0C0400u), 0x05040100u);
    const uint32_t ma = (((xa & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xa) & 0x80808080u;
    const uint32_t mb = (((xb & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xb) & 0x80808080u;
    uint32_t m = (ma >> 7) | (mb >> 3);
    // mask bit b = 8 jj + 4 h stands for byte jj of word h, i.e. for this lane's load j8 = 4 h + jj = (b >> 3) | (b & 4): start i0 + 64 j8
    const uint32_t ob = obase + i0;
    const int jmax = (int)(span - i0) >> 6;                 // starts up to the window's last k-mer vote: j8 <= jmax (negative: none of this lane's)
    (void)check;
    while (m) {
        const uint32_t bit = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        const uint32_t j8 = (bit >> 3) | (bit & 4u);
        uint32_t val = __builtin_amdgcn_perm(xb, xa, j8) & 0xFFu;      // byte j8 of the eight table bytes
        bool votes = (int)j8 <= jmax;                       // a start past the window's last k-mer does not vote
        if constexpr (KT == 7) {
            // k = 7..13: the table went by the k-mer's first six bases; the entry names a read position whose WHOLE k-mer is in
            // codes[] -- the same k-mer votes, a flagged one (it occurs twice in the read) does not, another one means the slot was
            // taken by a neighbour: the next slot then (rare: a hundred entries in 4096 slots)
            // cur[j8] by a tree of selects on j8's bits: registers, not an indexed array (which the compiler would keep in scratch)
            const bool s0 = (j8 & 1u) != 0u, s1 = (j8 & 2u) != 0u, s2 = (j8 & 4u) != 0u;
            const uint32_t a0 = s0 ? cur[1] : cur[0], a1 = s0 ? cur[3] : cur[2], a2 = s0 ? cur[5] : cur[4], a3 = s0 ? cur[7] : cur[6];
            const uint32_t b0 = s1 ? a1 : a0, b1 = s1 ? a3 : a2;
            const uint32_t w = s2 ? b1 : b0;
            const uint32_t wcode = (w >> bsh) & kmask;
            uint32_t slot = wcode & 0xFFFu;
            for (;;) {
                const uint32_t rc = codes[val - 1u];
                if (rc == wcode) break;
                if ((rc & 0x7FFFFFFFu) == wcode) { votes = false; break; }
                slot = (slot + 1u) & 0xFFFu;
                val = lds_byte(s.tbl, slot);
                if (val == 0u) { votes = false; break; }
            }
        }
        const uint32_t off = ob + (j8 << 6) - val;          // diagonal index - c0
        // (several chunks only, off_limit is all ones otherwise) a diagonal that belongs to another chunk does not vote either
        if (votes && off < off_limit) mx = vote_direct(s, off, ac, Q0, mx);
    }
    return mx;
}

// find_best_band (src/alignment.c:393-447): read_seeds x2 (29-68), bin_diagonals
// (70-128), bin_bands (130-140), select_band (142-181).  Window = contig[w0,w1),
// read piece = read[p0,p1), anchor in contig coordinates.
template <int KT, bool DIRECT>
__device__ __forceinline__ Band band_search(WaveLds& s, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ contig,
                            uint32_t w0, uint32_t w1, uint32_t anchor,
                            uint32_t p0, uint32_t p1, uint32_t k, uint32_t g, int lane, uint32_t read_pk8, uint32_t* codes IM_STAMP_ARG)
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

    // The window's k-mers are read from the 2-bit packed contig (4 bases per byte, first base in the
    // low bits).  A unit covers 512 consecutive k-mer starts; lane l takes starts l, l+64, l+128, ...
    // of it, eight in all, each one unaligned dword load at a fixed 16-byte stride.  Interleaving
    // matters: the read's true locus is a run of ~L consecutive hits, and with this mapping the run
    // is spread over all lanes instead of piling up in six of them.
    int bc = 0, bd = INT_MAX, bi = 0;                             // select_band's max, dist, indx
    // The diagonal scan that follows reads the window's raw bytes -- a different array from pk, behind an
    // address that depends on the vote: touch one byte per 128-byte line now (the value is never used) so
    // that the scan's load is an L2 hit.
    uint32_t touch = 0;
    if (w0 + 128u * lane < w1) touch = contig[w0 + 128u * lane];
    const uint32_t off_limit = numdiag > (uint32_t)kDiagChunk ? (uint32_t)kDiagChunk : 0xFFFFFFFFu;

    for (uint32_t c0 = 0; c0 < numdiag; c0 += step) {
        const int p_lo = max(0, (int)c0 - (int)nq);
        const int p_hi = min((int)npos - 1, (int)(c0 + kDiagChunk) - 2);
        const bool any = p_lo <= p_hi;
        const uint32_t span = any ? (uint32_t)(p_hi - p_lo) : 0u;           // starts p_lo + i, i in [0, span]
        const uint32_t nunit = any ? span / 512u + 1u : 0u;        // units of 512 starts = 8 per lane
        const uint32_t P0 = w0 + (uint32_t)(any ? p_lo : 0) + (uint32_t)lane;   // contig coordinate of this lane's first start
        const uint8_t* src = pk + (P0 >> 2);
        const uint32_t bsh = 2u * (P0 & 3u);                       // the same for all of a lane's starts: 64 starts = 16 bytes
        uint32_t wd[8], we[8];
#pragma unroll
        for (int j = 0; j < 8; j++) asm volatile("" : "=v"(we[j]));  // no value needed before the first load (saves the compiler's zero fill)
        if (DIRECT) {                                              // unconditional: P0 is a valid place whatever `any` says
#pragma unroll
            for (int j = 0; j < 8; j++) wd[j] = load_u32_unaligned(src + 16 * j);
        }
        if (c0 == 0) {
            table_build<KT, DIRECT>(s, p0, nq, k, lane, read_pk8, codes);
            IM_STAMP_B(0);
        }
        // clear the histogram: 16 bytes per lane and store, whole array (two stores for 1984 bytes)
        {
            uint4* d4 = reinterpret_cast<uint4*>(s.diag);
#pragma unroll
            for (int i = 0; i < (kDiagWords / 4 + 63) / 64; i++)
                if (64 * i + lane < kDiagWords / 4) d4[64 * i + lane] = make_uint4(0u, 0u, 0u, 0u);
        }
        wave_lds_sync();
        IM_STAMP_B(1);

        // vote: window k-mer at p and read-unique k-mer at q land on diagonal p - q + nq (102-105)
        const uint32_t obase = (uint32_t)p_lo + nq - c0 + 1u;      // off = obase + i - table value
        uint32_t mx = 0;                                           // largest (count, nearness, index) key this lane's votes produced (direct path)
        // the anchor relative to the chunk, and the smallest distance any of the chunk's diagonals has to it: distances
        // minus that are below 1920 whatever the anchor (inside the chunk: both sides; outside: monotone in the index)
        const int nbc = (int)(min(c0 + step, numdiag) - c0);
        const int arel = anchor_rel - (int)c0;
        const int dmin = arel < 0 ? -arel : (arel >= nbc ? arel - (nbc - 1) : 0);
        const int kd = 2047 + dmin;
        if (DIRECT) {
            // the anchor clamped into the chunk: every diagonal's nearness 2047 + dmin - |arel - off| is what it was (outside the
            // chunk the distance is monotone in the index, and dmin moves with the anchor), and the key needs no sign
            const uint32_t ac = (uint32_t)min(max(arel, 0), nbc - 1);
            uint32_t Q0 = 2047u * 2049u + (1u << 22);
            asm("" : "+s"(Q0));     // not a constant to the compiler: it would move it to the end of the key's sum, a fourth instruction per vote
            // units whose every start lies inside the window vote unchecked (one chunk only)
            const uint32_t nfull = off_limit == 0xFFFFFFFFu ? (span + 1u) / 512u : 0u;
            for (uint32_t u = 0; u < nunit; u += 2) {
                mx = vote_unit_direct<KT>(s, wd, we, src + 128u * (u + 1u), u + 1 < nunit, bsh, kmask,
                                          512u * u + (uint32_t)lane, span, obase, u >= nfull, off_limit, ac, Q0, mx, codes);
                if (u + 1 < nunit)
                    mx = vote_unit_direct<KT>(s, we, wd, src + 128u * (u + 2u), u + 2 < nunit, bsh, kmask,
                                              512u * (u + 1u) + (uint32_t)lane, span, obase, u + 1 >= nfull, off_limit, ac, Q0, mx, codes);
            }
        } else {
            for (uint32_t u = 0; u < nunit; u++) {
                const uint32_t i0 = 512u * u + (uint32_t)lane;     // index of this lane's j = 0 start
#pragma unroll 4
                for (int j = 0; j < 8; j++) {
                    const uint32_t i = i0 + 64u * j;
                    if (i > span) continue;
                    uint64_t dd;
                    __builtin_memcpy(&dd, src + 128u * u + 16 * j, 8);
                    const uint32_t v = table_lookup<DIRECT>(s, (uint32_t)(dd >> bsh) & kmask);
                    if (v == 0u || v == 0xFFu) continue;
                    const uint32_t off = obase + i - v;
                    if (off < (uint32_t)kDiagChunk) {
                        // the same key as the direct path: the vote that lifts a diagonal to its final count names it
                        const uint32_t bsel = (off & 3u) * 8u;
                        const uint32_t old = atomicAdd(&s.diag[off >> 2], 1u << bsel);
                        const uint32_t cnt = ((old >> bsel) & 255u) + 1u;
                        const uint32_t near = (uint32_t)(kd - abs(arel - (int)off));
                        mx = max(mx, (cnt << 22) | (near << 11) | (2047u - off));
                    }
                }
            }
        }
        wave_lds_sync();
        IM_STAMP_B(2);

        // bin_bands + select_band over i in [c0, iend)
        const uint32_t iend = min(c0 + step, numdiag);
        const uint32_t nband = numdiag - g;                       // bands[i] == 0 for i >= nband (135)
        if (g == 0) {
            // One diagonal per band: select_band's winner of the chunk (most votes, nearest the anchor, smallest index),
            // merged into the running winner over the chunks.
            {
                // the votes reported (count, nearness, index) as they landed: one reduction names the chunk's band
                const uint32_t K = (uint32_t)wave_max((int)mx);          // keys stay below 2^30
                const int M = (int)(K >> 22);
                if (M > 0 && M >= bc) {
                    const int d = dmin + 2047 - (int)((K >> 11) & 2047u);
                    const int i = (int)c0 + 2047 - (int)(K & 2047u);
                    if (M > bc || d < bd || (d == bd && i < bi)) { bc = M; bd = d; bi = i; }
                }
            }
        } else {
            for (uint32_t dw = lane; 4u * dw < iend - c0; dw += 64) {
                const uint32_t v = s.diag[dw];
#pragma unroll
                for (int bb = 0; bb < 4; bb++) {
                    const uint32_t i = c0 + 4u * dw + bb;
                    if (i >= iend) break;
                    int cnt = (int)((v >> (8 * bb)) & 255u);
                    // bin_bands: sum of g+1 neighbouring diagonals
                    if (i < nband) for (uint32_t e = 1; e <= g; e++) cnt += (int)lds_byte(s.diag, i - c0 + e);
                    else cnt = 0;
                    const int d = abs(anchor_rel - (int)i);
                    if (cnt > bc || (cnt == bc && (d < bd || (d == bd && (int)i < bi)))) { bc = cnt; bd = d; bi = (int)i; }
                }
            }
        }
        wave_lds_sync();
        IM_STAMP_B(3);
    }
    // select_band's order: most votes, then nearest the anchor, then smallest index
    if (g == 0) {
        if (bc == 0) {
            // no vote anywhere: every diagonal ties at 0 and the nearest to the anchor wins
            bi = min(max(anchor_rel, 0), (int)numdiag - 1);
        }
    } else {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int oc = __shfl_xor(bc, o), od = __shfl_xor(bd, o), oi = __shfl_xor(bi, o);
            if (oc > bc || (oc == bc && (od < bd || (od == bd && oi < bi)))) { bc = oc; bd = od; bi = oi; }
        }
    }
    asm volatile("" :: "v"(touch));                                // keeps the touch loads alive, emits nothing
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
    uint32_t eqbits;        // per lane: bit j = read position 4*lane+j is an aligned '=' (read coordinates)
};

// local_align + ALIGN + fetch_cigar for low == up == d (src/localalign.c:100-176
// with band == 1; closed form validated in SURVEY.md A.5a):
//   forward : c_t = max(0, c_{t-1} + w_t); end = first t where c_t is the strict maximum
//   reverse : start = largest s <= end with sum_{s..end} w == best
// Match flags of the aligned positions come back in Aln::eqbits, in read coordinates.
template <bool WANT_RUNS>            // the leading / trailing '=' runs are only used of the first piece
__device__ __forceinline__ Aln diag_scan(WaveLds& s, const uint8_t* __restrict__ contig,
                         uint32_t w0, uint32_t w1, uint32_t p0, uint32_t p1, int d, int lane)
{
    Aln a;
    a.st = 0; a.q1 = a.q2 = a.r1 = a.r2 = 0; a.f = a.l = 0; a.eqbits = 0u;
    const int M = (int)(p1 - p0), N = (int)(w1 - w0);
    if (M <= 0 || N <= 0 || d < -M || d > N) { a.st = IM_ST_ABORT; return a; }   // src/localalign.c:31-32,70-77
    const int t_lo = max(0, -d), t_hi = min(M, N - d);

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
    // the best score and the FIRST cell that holds it in one reduction: scores are 0..255 (c_t >= 0, at most one per base
    // of a read of up to 255), cells 0..255 -- (score << 8) | (255 - t), largest wins
    int bk = (c[0] << 8) | (255 - t0);
#pragma unroll
    for (int j = 1; j < 4; j++) bk = max(bk, (c[j] << 8) | (255 - t0 - j));
    bk = wave_max(bk);
    const int best = bk >> 8;
    if (best <= 0) return a;                                       // score <= 0 (src/alignment.c:365-372)
    const int end = 255 - (bk & 255);
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
            if (eq[j]) flags |= 1u << j;
            else { fm = min(fm, t); lm = max(lm, t); }
        }
    }
    if (WANT_RUNS) {
        fm = wave_min(fm); lm = wave_max(lm);
        a.f = (fm == INT_MAX ? end + 1 : fm) - start;
        a.l = end - (lm < 0 ? start - 1 : lm);
    }
    // flags live in piece coordinates t (lane owns t0..t0+3); read position x = p0 + t belongs to
    // lane x >> 2, so lane L collects bits 4L - p0 .. 4L - p0 + 3 of the 256-bit flag string from
    // the two lanes that hold them (ds_bpermute: no LDS storage, no barrier)
    {
        const uint32_t f4 = flags;
        if (p0 == 0u) a.eqbits = f4;
        else {
            const int base = 4 * lane - (int)p0;
            const int ql = base >> 2, r = base & 3;                 // floor division
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((ql & 63) << 2, (int)f4);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(((ql + 1) & 63) << 2, (int)f4);
            const uint32_t lo_ok = (ql >= 0 && ql < 64) ? lo : 0u;
            const uint32_t hi_ok = (ql + 1 >= 0 && ql + 1 < 64) ? hi : 0u;
            a.eqbits = ((lo_ok | (hi_ok << 4)) >> r) & 0xFu;
        }
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

// attempt_pe_alignment -> attempt_diagonal_alignments (src/alignment.c:539-799)
template <int KT, bool DIRECT>
__device__ __forceinline__ void realign_one(WaveLds& s, const RealignArgs& A, int c, int lane, uint32_t* codes)
{
    IM_STAMP_DECL
    im_read_result* out = &A.batch.out[c];
    // Every per-read scalar is the same in all 64 lanes.  Saying so (v_readfirstlane) puts the
    // values in SGPRs and turns the control flow below into scalar branches instead of exec-mask
    // bookkeeping -- the compiler cannot prove uniformity of loaded values on its own.
    const int64_t off = sload(A.batch.base_off + c);
    const int64_t Lraw = sload(A.batch.read_len + c);
    const int tid = sload(A.batch.tid + c);
    const int anchor = sload(A.batch.anchor + c);
    const int R = sload(A.batch.range_max + c);
    const uint32_t k = KT == 6 ? 6u : A.P.klength, g = KT ? 0u : A.P.numgaps, eth = A.P.ethreshold;

    if (lane < 16) reinterpret_cast<uint32_t*>(&out->band[0])[lane] = 0u;
    if (lane < 7) out->reserved[lane] = 0;
    if (!A.keep_slots) write_slots(A, c, 0, -1, 0, 0, lane);      // keep: the CIGAR-derived evidence stays unless replaced (src/indelminer.c:504-510)
    if (Lraw <= 0 || Lraw > kShortRead || tid < 0 || tid >= A.ref.n_contigs || (off & 3)) {
        finish(out, (Lraw > kShortRead || (off & 3)) ? IM_ST_UNSUPPORTED : IM_ST_ABORT, 0, lane);
        return;
    }
    const int L = (int)Lraw;
    const uint8_t* contig = A.ref.ascii + sload(A.ref.asc_off + tid);
    const uint8_t* pk = A.ref.pk + sload(A.ref.pk_off + tid);
    const int clen = sload(A.ref.len + tid);

    // stage the read; lane l also keeps the 2-bit codes of its four bases (read_pk8, for the k-mer table)
    uint32_t rdw = 0;
    if (4 * lane < L) rdw = *reinterpret_cast<const uint32_t*>(A.batch.bases + off + 4 * lane);
    {
        const int rem = L - 4 * lane;                              // zero the bytes past the read
        if (rem < 4) rdw &= (rem <= 0) ? 0u : ((1u << (8 * rem)) - 1u);
    }
    const uint32_t read_pk8 = code2x4(rdw);      // 2-bit codes of this lane's four bases, first base in the low bits
    s.rd[lane] = rdw;
    if (lane < 4) s.rd[64 + lane] = 0u;
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
    const Band b1 = band_search<KT, DIRECT>(s, pk, contig, (uint32_t)left1, (uint32_t)right1, (uint32_t)anchor, 0u, (uint32_t)L, k, g, lane, read_pk8, codes IM_STAMP_PASS(1));
    if (b1.st) { finish(out, b1.st, 1, lane); return; }
    const Aln a1 = diag_scan<true>(s, contig, (uint32_t)left1, (uint32_t)right1, 0u, (uint32_t)L, b1.low, lane);
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
    const Band b2 = band_search<KT, DIRECT>(s, pk, contig, w0, w1, anc, p0, p1, k, g, lane, read_pk8, codes IM_STAMP_PASS(8));
    if (b2.st) { finish(out, b2.st, 2, lane); return; }
    const Aln a2 = diag_scan<false>(s, contig, w0, w1, p0, p1, b2.low, lane);
    store_band(out, 1, b2, a2, lane);
    IM_STAMP(13);
    if (a2.st) { finish(out, a2.st, 2, lane); return; }
    const int r3 = a2.r1, r4 = a2.r2, q3 = a2.q1, q4 = a2.q2;
    if (want_tail) { if (q4 != L || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 623-627, 679-683
    else           { if (q3 != 0 || q3 == q4) { finish(out, IM_ST_NONE, 2, lane); return; } }   // 645-649, 701-705
    if (!(q1 < q2 && q3 < q4)) { finish(out, IM_ST_ABORT, 2, lane); return; }             // 720-721

    // combine (723-754).  "A" = the piece that starts at read offset 0, "B" = the one that ends at L.
    uint32_t wa, wb;                    // Aln::eqbits of the A / B piece
    int qa2, rA, qb1, rB;               // A = read[0,qa2) at contig rA.. ; B = read[qb1,L) at contig rB..
    bool split;                         // true: overlapping pieces, choose the split point (K4)
    if (q1 > q3 && q1 <= q4)        { wa = a2.eqbits; qa2 = q4; rA = r3; wb = a1.eqbits; qb1 = q1; rB = r1; split = true;  }
    else if (q3 > q1 && q3 <= q2)   { wa = a1.eqbits; qa2 = q2; rA = r1; wb = a2.eqbits; qb1 = q3; rB = r3; split = true;  }
    else if (q1 > q4 && r1 == r4)   { wa = a2.eqbits; qa2 = q4; rA = r3; wb = a1.eqbits; qb1 = q1; rB = r1; split = false; }
    else if (q3 > q2 && r2 == r3)   { wa = a1.eqbits; qa2 = q2; rA = r1; wb = a2.eqbits; qb1 = q3; rB = r3; split = false; }
    else { finish(out, IM_ST_NONE, 2, lane); return; }
    // find_best_del_candidate asserts its first piece starts at read offset 0 (314-315)
    // (holds by the accept conditions above: the A piece has q == 0)

    // per-position match flags of A on [0,qa2) and B on [qb1,L)
    const int x0 = 4 * lane;
    int fa[4], fb[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int x = x0 + j;
        fa[j] = (x < qa2) ? (int)((wa >> j) & 1u) : 0;
        fb[j] = (x >= qb1 && x < L) ? (int)((wb >> j) & 1u) : 0;
    }
    const int ta = fa[0] + fa[1] + fa[2] + fa[3], tb = fb[0] + fb[1] + fb[2] + fb[3];
    const int iab = wave_scan_add(ta | (tb << 16), lane);          // both counts in one scan: each stays below 2^15
    const int tot = __builtin_amdgcn_readlane(iab, 63);
    const int ia = iab & 0xFFFF, ib = iab >> 16;
    const int totA = tot & 0xFFFF, totB = tot >> 16;
    int pa[4], pb[4];                   // exclusive prefix counts at x
    pa[0] = ia - ta; pb[0] = ib - tb;
#pragma unroll
    for (int j = 1; j < 4; j++) { pa[j] = pa[j - 1] + fa[j - 1]; pb[j] = pb[j - 1] + fb[j - 1]; }

    int index, nextindex, matches;
    if (split) {
        // count_matches(i) = '=' of A in read[0,i) + '=' of B in read[i,L); X counts are
        // L - that, so "max matches, then min mismatches, first wins" is the first maximum.
        // one reduction for both: (matches << 8) | (255 - x), largest wins -- matches and x stay below 256
        int bk = -1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int x = x0 + j;
            if (x >= qb1 && x <= qa2) bk = max(bk, ((pa[j] + (totB - pb[j])) << 8) | (255 - x));
        }
        bk = wave_max(bk);
        if (bk < 0) { finish(out, IM_ST_ABORT, 2, lane); return; }                         // forceassert(index != -1)
        index = 255 - (bk & 255);
        nextindex = index;
        matches = bk >> 8;
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
    seg_indel = __builtin_amdgcn_readlane(seg_indel, index >> 2);     // the lane that owns read position `index` set it
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
__global__ __launch_bounds__(64, IM_WAVES_PER_SIMD) void realign_kernel(RealignArgs A)
{
    __shared__ WaveLds s;
    const int lane = threadIdx.x;
    // ONE read per wave and no loop around it: nothing of the launch's arguments has to stay in registers for a next
    // read (launch_realign cuts a batch beyond the largest grid into slices).  The batch size may live on the device
    // (the triage kernel's running count): the launch is sized for an upper bound and the workgroups beyond the count
    // leave at once.
    const int n = A.n_dev ? min(sload(A.n_dev), A.batch.n) : A.batch.n;
    const int left = n - A.first;                           // reads of this slice and behind it
    const int G = min((int)gridDim.x, (left + 7) / 8 * 8);  // multiple of 8
    if ((int)blockIdx.x >= G) return;
    // blocks with equal blockIdx % 8 share an XCD (observed round-robin placement,
    // speed only): give each XCD a contiguous run of reads.
    const int c = A.first + (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3);
    if (c >= n) return;
    uint32_t* codes = nullptr;
    if constexpr (KT == 7) {
        __shared__ uint32_t codes_lds[256];          // the read's whole k-mers, by read position (prefix-table mode)
        codes = codes_lds;
    }
    realign_one<KT, DIRECT>(s, A, c, lane, codes);
}

// ---- numgaps > 0: banded affine-gap path ------------------------------------------
//
// With -g N the band handed to local_align is N+1 diagonals wide and the reference runs a banded Gotoh
// local pass forward, a reverse pass from the end cell, and ALIGN's linear-space global alignment through
// the middle diagonal's crossing points (src/localalign.c:15-196, src/globalalign.c:66-401).  Co-optimal
// paths are common with +1/-10/-20/-10 scoring and split-read clusters key on exact breakpoints, so every
// comparison keeps the reference's strictness.  Here the dynamic programs run across the
// wave: LANE = DIAGONAL of the band (band <= 61), one step per read row:
//   * the vertical gap state and the diagonal move come from the neighbouring lane's previous row (DPP shifts);
//   * the horizontal gap chain along a row -- the only sequential dependency inside a row -- is a prefix maximum:
//       e(curd) = max over origins o < curd of  x(o) - g - h (curd - o),   x = max(diagonal, vertical) of cell o
//     (a cell whose best is itself the horizontal state never re-opens a gap profitably: g > 0), ties to the
//     leftmost origin because the reference lets an extension win over a new opening; one 6-step wave scan of a
//     packed (value + h lane, lane) key, the crossing-point pointer of the winning origin fetched with one bpermute;
//   * ALIGN's recursion (middle diagonal, the triangles on either side) is an explicit frame stack in LDS;
//     the crossing-point records of the middle diagonal (MP/MT) are one packed word per row.
// The alignment is kept per READ POSITION (class =/X/I plus the length of the reference gap in front of the
// position), from which the CIGAR, the split-point search (lane = candidate split) and the final segment
// list (lane = op) are made without a serial pass over the read.

#define BFAIL(G, code) do { if (threadIdx.x == 0) (G).tmp[IM_MAX_OPS + 7] = (code); } while (0)
constexpr int kGapMaxBand = 62;          // numgaps <= 60; lane band and band + 1 hold the MININT borders
constexpr int kGapNegInf = -9999999;     // MININT, src/localalign.c:3
constexpr int kGapOpen = 10, kGapExt = 10;
constexpr int kKeyBias = 30000;          // scores of cells that can be gap origins stay far inside +-30000 (M, N <= ~1300)

enum : uint8_t { kPosEq = 0, kPosX = 1, kPosI = 2, kPosNone = 255 };

struct BandLds {
    uint32_t mp[264];                    // per row of the middle diagonal: MP0+1 (9) | MT0 (2) | copy1 | MP1+1 (9) | copy2 | MP2+1 (9)
    uint16_t fp[264];                    // forward dividing points: (next + 1) | type << 10
    uint8_t  wb[1024];                   // staged window bytes: B index j (1-based) lives at wb[j - wb_base]
    uint8_t  kind[2][260];               // per piece position p (0-based): class of read base p
    uint16_t dlen[2][264];               // reference bases skipped in front of position p (p = 0 .. M)
    uint32_t ops[2][IM_MAX_OPS + 4];     // the two band alignments' CIGARs
    int32_t  stk[10][16];                // ALIGN's recursion as frames
    int32_t  nops[2];
    int32_t  aln[2][4];                  // r1 r2 q1 q2
    int32_t  wb_base;
    int32_t  tmp[IM_MAX_OPS + 8];
};

__device__ __forceinline__ int gw(uint32_t a, uint32_t b) { return a == b ? kScoreMatch : kScoreMismatch; }
__device__ __forceinline__ int wshl1(int v, int fill) { return dpp_mov<kDppWaveShl1>(fill, v); }    // lane i <- lane i + 1
__device__ __forceinline__ int wshr1(int v, int fill) { return dpp_mov<kDppWaveShr1>(fill, v); }    // lane i <- lane i - 1
__device__ __forceinline__ int lane_get(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// inclusive running maximum over the lanes of the band: STEPS doublings cover a band of 2^STEPS lanes; a band of <= 16
// diagonals lies in one DPP row (four steps at most), wider ones take the two row broadcasts as well
template <int STEPS>
__device__ __forceinline__ int band_scan_max(int v)
{
    v = max(v, dpp_mov<kDppRowShr1>(INT_MIN, v));
    if (STEPS >= 2) v = max(v, dpp_mov<kDppRowShr2>(INT_MIN, v));
    if (STEPS >= 3) v = max(v, dpp_mov<kDppRowShr4>(INT_MIN, v));
    if (STEPS >= 4) v = max(v, dpp_mov<kDppRowShr8>(INT_MIN, v));
    if (STEPS >= 6) {
        v = max(v, dpp_mov<kDppBcast15, 0xa>(INT_MIN, v));
        v = max(v, dpp_mov<kDppBcast31, 0xc>(INT_MIN, v));
    }
    return v;
}
__device__ __forceinline__ int band_scan_steps(int band) { return band <= 2 ? 1 : band <= 4 ? 2 : band <= 8 ? 3 : band <= 16 ? 4 : 6; }

// key of the horizontal-gap scan: larger value wins, then the smaller lane
__device__ __forceinline__ int ekey_pack(int x, int lane) { const int v = max(x, -kKeyBias + 1) + kGapExt * lane + kKeyBias; return (v << 6) | (63 - lane); }

struct BandAln { int st, r1, r2, q1, q2; };

// stage the slice of the window a band alignment can touch: B indices [jlo, jhi] (1-based); outside the window 0
__device__ __forceinline__ int band_stage_window(BandLds& G, const uint8_t* contig, int w0, int N, int M, int low, int up, int lane)
{
    const int lo = max(-M, low), hi = min(N, up);
    int jlo = max(1, lo + 1) - 2, jhi = min(N, M + hi) + 2;
    if (jlo < 0) jlo = 0;
    if (jhi - jlo + 1 > (int)sizeof(G.wb)) { BFAIL(G, 1); return IM_ST_OVERFLOW; }
    for (int t = lane; t <= jhi - jlo; t += 64) {
        const int j = jlo + t;
        G.wb[t] = (j >= 1 && j <= N) ? contig[(int64_t)w0 + j - 1] : 0;
    }
    if (lane == 0) G.wb_base = jlo;
    wave_lds_sync();
    return 0;
}

// ---- ALIGN (src/globalalign.c:333-401): per-position emission ----------------------------------

struct Emit { int ia, jb; };     // cursors: read bases / reference bases of the aligned sub-strings consumed so far

__device__ __forceinline__ void emit_rep1(BandLds& G, int which, const uint8_t* A1, const uint8_t* B1, int pos0, Emit& E, int lane)
{
    // A1[i], B1[j]: 1-based sub-strings
    if (lane == 0) G.kind[which][pos0 + E.ia] = A1[E.ia + 1] == B1[E.jb + 1] ? kPosEq : kPosX;
    E.ia++; E.jb++;
}
__device__ __forceinline__ void emit_rep_run(BandLds& G, int which, const uint8_t* A1, const uint8_t* B1, int pos0, Emit& E, int n, int lane)
{
    for (int p = lane; p < n; p += 64) G.kind[which][pos0 + E.ia + p] = A1[E.ia + p + 1] == B1[E.jb + p + 1] ? kPosEq : kPosX;
    E.ia += n; E.jb += n;
}
__device__ __forceinline__ void emit_del(BandLds& G, int which, int pos0, Emit& E, int n, int lane)     // read bases without a partner
{
    for (int p = lane; p < n; p += 64) G.kind[which][pos0 + E.ia + p] = kPosI;
    E.ia += n;
}
__device__ __forceinline__ void emit_ins(BandLds& G, int which, int pos0, Emit& E, int n, int lane)     // reference bases skipped
{
    if (lane == 0) G.dlen[which][pos0 + E.ia] = (uint16_t)(G.dlen[which][pos0 + E.ia] + n);
    E.jb += n;
}

// The forward pass of align() (src/globalalign.c:100-248) on A1[1..M], B1[1..N] with the band [low, up]: fills the
// crossing records of the middle diagonal and returns the trace start (k, l) and the score.  lane = curd - 1.
struct FwdOut { int k, l, v, rmid; };
template <int STEPS>
__device__ __forceinline__ FwdOut band_global_forward_t(BandLds& G, const uint8_t* A1, const uint8_t* B1, int M, int N, int low, int up,
                                                        int tb, int te, int lane)
{
    const int g = kGapOpen, h = kGapExt, m = g + h;
    const int band = up - low + 1;
    const int midd = band / 2 + 1;
    FwdOut o; o.rmid = low + midd - 1;
    int leftd = 1 - low, rightd = up - low + 1;
    const int curd = lane + 1;
    // initialisation (112-146)
    int CP, DP;
    if (leftd < midd)      { CP = DP = curd < midd ? -1 : 0; if (lane == 0) G.mp[0] = 0u; }
    else if (leftd > midd) { const int fr = leftd - midd; CP = DP = curd <= midd ? fr : -1; if (lane == 0) G.mp[fr] = 0u; }
    else                   { CP = DP = 0; if (lane == 0) G.mp[0] = 0u; }
    int CC = kGapNegInf, DD = kGapNegInf;
    {
        const int t0 = (tb == 2) ? 0 : -g;
        if (curd == leftd) { CC = 0; DD = (tb == 1) ? 0 : -g; }
        else if (curd > leftd && curd <= rightd) { CC = t0 - h * (curd - leftd); DD = CC - g; }
    }
    int c_fin = 0, d_fin = 0, e_fin = 0, ip_fin = 0;
    for (int i = 1; i <= M; i++) {
        if (i > N - up) rightd--;
        if (leftd > 1) leftd--;
        const uint32_t ai = A1[i];
        const int ccn = wshl1(CC, kGapNegInf), ddn = wshl1(DD, kGapNegInf), cpn = wshl1(CP, -1), dpn = wshl1(DP, -1);
        const int co = ccn - m, dx = ddn - h;
        const bool opend = co > dx;
        const int d = opend ? co : dx;
        const int dpt = opend ? cpn : dpn;
        const int jb = curd + low - 1 + i;
        const bool active = curd >= leftd && curd <= rightd;
        const uint32_t bch = (active && jb >= 1) ? B1[jb] : 256u;
        const int diag = CC + gw(ai, bch);
        const bool is_left = curd == leftd;
        // the cell without its horizontal state: value x, pointer xp
        int x, xp;
        if (is_left) {
            const int c0 = jb > 0 ? diag : co;
            if (d > c0 || jb <= 0) { x = d; xp = dpt; } else { x = c0; xp = CP; }
        } else if (diag < d) { x = d; xp = dpt; } else { x = diag; xp = CP; }
        // horizontal chain: prefix maximum over the origins to the left
        const int key = active ? ekey_pack(x, lane) : 0;
        const int ex = wshr1(band_scan_max<STEPS>(key), 0);
        const bool has_e = !is_left && ex != 0;
        const int e = has_e ? (ex >> 6) - kKeyBias - h * lane - g : kGapNegInf;
        const int origin = 63 - (ex & 63);
        const int xpf = (curd == midd) ? i : xp;             // the middle cell hands on its own row as the crossing point
        int ip = __builtin_amdgcn_ds_bpermute(origin << 2, xpf);
        if (origin + 1 <= midd && curd > midd) ip = i;
        // the cell (168-187)
        int c = x, cp = xp;
        if (!is_left) {
            c = diag; cp = CP;
            if (diag < d || diag < e) { if (e > d) { c = e; cp = ip; } else { c = d; cp = dpt; } }
        }
        int dpn_new = dpt;
        if (curd == midd) {
            if (!is_left && active) {
                // the crossing records of row i (189-231)
                const int mp1 = ip, mp2 = dpt;
                int mp0, mt0;
                if (diag < d || diag < e) { if (e > d) { mp0 = mp1; mt0 = 2; } else { mp0 = mp2; mt0 = 1; } }
                else { mp0 = i - 1; mt0 = 0; }
                const uint32_t copy1 = (c - g > e) ? 1u : 0u, copy2 = (c - g > d) ? 1u : 0u;
                G.mp[i] = (uint32_t)(mp0 + 1) | ((uint32_t)mt0 << 9) | (copy1 << 11) | ((uint32_t)(mp1 + 1) << 12) | (copy2 << 21) | ((uint32_t)(mp2 + 1) << 22);
            }
            cp = i; dpn_new = i;
        }
        const int e_out = is_left ? c - g : e;
        const int ip_out = (curd == midd) ? i : (is_left ? cp : ip);
        if (active) { CC = c; DD = d; CP = cp; DP = dpn_new; }
        if (i == M) {
            const int last = rightd - 1;                        // lane of the row's last cell
            c_fin = lane_get(c, last); d_fin = lane_get(d, last); e_fin = lane_get(e_out, last); ip_fin = lane_get(ip_out, last);
        }
    }
    const int last = rightd - 1;
    // which state the trace starts in (233-248)
    if (te == 1 && d_fin + g > c_fin)      { o.k = lane_get(DP, last); o.l = 2; }
    else if (te == 2 && e_fin + g > c_fin) { o.k = ip_fin;             o.l = 1; }
    else                                   { o.k = lane_get(CP, last); o.l = 0; }
    if (o.rmid > N - M) o.l = 2;
    else if (o.rmid < N - M) o.l = 1;
    o.v = c_fin;
    wave_lds_sync();
    return o;
}

__device__ __forceinline__ FwdOut band_global_forward(BandLds& G, const uint8_t* A1, const uint8_t* B1, int M, int N, int low, int up,
                                                      int tb, int te, int lane)
{
    switch (band_scan_steps(up - low + 1)) {
    case 1: return band_global_forward_t<1>(G, A1, B1, M, N, low, up, tb, te, lane);
    case 2: return band_global_forward_t<2>(G, A1, B1, M, N, low, up, tb, te, lane);
    case 3: return band_global_forward_t<3>(G, A1, B1, M, N, low, up, tb, te, lane);
    case 4: return band_global_forward_t<4>(G, A1, B1, M, N, low, up, tb, te, lane);
    default: return band_global_forward_t<6>(G, A1, B1, M, N, low, up, tb, te, lane);
    }
}

__device__ __forceinline__ void mp_decode(uint32_t w, int l, int& k_out, int& l_out)
{
    const int mp0 = (int)(w & 511u) - 1, mt0 = (int)((w >> 9) & 3u);
    if (l == 0) { k_out = mp0; l_out = mt0; }
    else if (l == 1) { if ((w >> 11) & 1u) { k_out = mp0; l_out = mt0; } else { k_out = (int)((w >> 12) & 511u) - 1; l_out = 2; } }
    else { if ((w >> 21) & 1u) { k_out = mp0; l_out = mt0; } else { k_out = (int)((w >> 22) & 511u) - 1; l_out = 1; } }
}
__device__ __forceinline__ void fp_fetch(const BandLds& G, int k, int& l_out, int& kt_out)
{
    const uint32_t w = G.fp[k];
    l_out = (int)(w & 1023u) - 1; kt_out = (int)(w >> 10);
}

// ALIGN's divide and conquer (src/globalalign.c:66-307) on sub-strings A0[1..Ma], B0[1..Na]: frames instead of recursion
__device__ int band_global_align(BandLds& G, int which, const uint8_t* A0, const uint8_t* B0, int Ma, int Na, int low, int up,
                                 int pos0, int lane, int* score_out)
{
    enum { F_AO, F_BO, F_M, F_N, F_LOW, F_UP, F_TB, F_TE, F_PHASE, F_K, F_L, F_KT, F_RMID };
    Emit E; E.ia = 0; E.jb = 0;
    int sp = 0, top_score = 0, guard = 0; bool first = true;
    auto push = [&](int ao, int bo, int M, int N, int lo, int u, int tb, int te) {
        if (lane == 0) { int32_t* f = G.stk[sp]; f[F_AO] = ao; f[F_BO] = bo; f[F_M] = M; f[F_N] = N; f[F_LOW] = lo; f[F_UP] = u; f[F_TB] = tb; f[F_TE] = te; f[F_PHASE] = 0; }
        sp++;
    };
    push(0, 0, Ma, Na, low, up, 0, 0);
    wave_lds_sync();
    while (sp > 0) {
        if (sp > 9) { BFAIL(G, 2); return IM_ST_OVERFLOW; }
        int32_t* f = G.stk[sp - 1];
        const int ao = uni(f[F_AO]), bo = uni(f[F_BO]), M = uni(f[F_M]), N = uni(f[F_N]), lo = uni(f[F_LOW]), u = uni(f[F_UP]);
        const int tb = uni(f[F_TB]), te = uni(f[F_TE]), phase = uni(f[F_PHASE]);
        int k = uni(f[F_K]), l = uni(f[F_L]), kt = uni(f[F_KT]);
        int rmid = uni(f[F_RMID]);
        const uint8_t* A1 = A0 + ao; const uint8_t* B1 = B0 + bo;
        int nphase = phase;
        if (++guard > 4096) { BFAIL(G, 3); return IM_ST_ABORT; }               // every frame makes progress; this only bounds a corrupted walk
        bool do_push = false; int c_ao = 0, c_bo = 0, c_M = 0, c_N = 0, c_lo = 0, c_u = 0, c_tb = 0, c_te = 0;
        bool pop = false;
        wave_lds_sync();
        if (phase == 0) {
            if (N <= 0) { if (M > 0) emit_del(G, which, pos0, E, M, lane); pop = true; if (first) { top_score = -1; first = false; } }
            else if (M <= 0) { emit_ins(G, which, pos0, E, N, lane); pop = true; if (first) { top_score = -1; first = false; } }
            else if (u - lo + 1 <= 1) { emit_rep_run(G, which, A0, B0, pos0, E, M, lane); pop = true; if (first) { top_score = -1; first = false; } }
            else {
                if (u - lo + 1 > kGapMaxBand) { BFAIL(G, 4); return IM_ST_OVERFLOW; }
                const FwdOut fo = band_global_forward(G, A1, B1, M, N, lo, u, tb, te, lane);
                if (first) { top_score = fo.v; first = false; }
                // trace back through the crossing records, turning them into forward pointers (252-257)
                // The chain is a pointer chase from row to row: the records travel in registers (row 64 j + lane in mr[j], M <= 255)
                // and every hop is a v_readlane instead of an LDS round trip.
                int kk = fo.k, ll = fo.l, r = -1;
                uint32_t mr[4];
#pragma unroll
                for (int j = 0; j < 4; j++) mr[j] = (64 * j + lane <= M) ? G.mp[64 * j + lane] : 0u;
                while (kk > -1) {
                    if (kk > M || kk > 255) { BFAIL(G, 5); return IM_ST_ABORT; }
                    if (lane == 0) G.fp[kk] = (uint16_t)((uint32_t)(r + 1) | ((uint32_t)ll << 10));
                    const int rr = kk; int nk, nl;
                    const int hi = rr >> 6;
                    const uint32_t mine = hi == 0 ? mr[0] : hi == 1 ? mr[1] : hi == 2 ? mr[2] : mr[3];
                    mp_decode((uint32_t)__builtin_amdgcn_readlane((int)mine, rr & 63), ll, nk, nl);
                    if (nk >= rr) { BFAIL(G, 6); return IM_ST_ABORT; }           // crossing points strictly descend
                    r = rr; kk = nk; ll = nl;
                }
                wave_lds_sync();
                if (r == -1) {
                    // the optimal alignment did not cross the middle diagonal: same strings, half the band (259-262)
                    if (lane == 0) { if (fo.rmid < 0) f[F_LOW] = fo.rmid + 1; else f[F_UP] = fo.rmid - 1; }
                } else {
                    k = r; fp_fetch(G, k, l, kt);
                    rmid = fo.rmid;
                    if (lane == 0) f[F_RMID] = fo.rmid;
                    // first block (268-275)
                    if (fo.rmid < 0) { nphase = 1; do_push = true; c_ao = ao; c_bo = bo; c_M = r - 1; c_N = r + fo.rmid; c_lo = fo.rmid + 1; c_u = min(u, r + fo.rmid); c_tb = tb; c_te = 1; }
                    else if (fo.rmid > 0) { nphase = 2; do_push = true; c_ao = ao; c_bo = bo; c_M = r; c_N = r + fo.rmid - 1; c_lo = max(-r, lo); c_u = fo.rmid - 1; c_tb = tb; c_te = 2; }
                    else nphase = 3;
                }
            }
        } else if (phase == 1) { emit_del(G, which, pos0, E, 1, lane); nphase = 3; }
        else if (phase == 2) { emit_ins(G, which, pos0, E, 1, lane); nphase = 3; }
        else if (phase == 4) { emit_del(G, which, pos0, E, 1, lane); k = l; fp_fetch(G, k, l, kt); nphase = 3; }
        else if (phase == 5) { emit_ins(G, which, pos0, E, 1, lane); k = l; fp_fetch(G, k, l, kt); nphase = 3; }
        else if (phase == 6) pop = true;
        if (nphase == 3 && !do_push && !pop) {
            const int t2 = u - rmid - 1, t3 = lo - rmid + 1;
            // intermediate blocks (280-295): runs of diagonal crossings are walked here without touching the stack
            // A diagonal crossing leads to the next row (REP, 283-285): lane t looks at the record of row k + t, the leading lanes
            // whose records say "diagonal, on to the row below" are one run, emitted at once.  Rows off the chain hold records of
            // earlier passes; they lie behind the first lane that fails, which ends the run.
            while (l > -1 && kt == 0) {
                const int row = k + lane;
                const uint32_t w = row <= M ? (uint32_t)G.fp[row] : 0xFFFFu;
                const bool on = (int)(w & 1023u) - 1 == row + 1 && (w >> 10) == 0u;
                const uint64_t off = ~__ballot(on);
                const int n = off ? __builtin_ctzll(off) : 64;
                if (n == 0) { emit_rep1(G, which, A0, B0, pos0, E, lane); k = l; }       // the record's own word (never the case: a diagonal step is one row)
                else { emit_rep_run(G, which, A0, B0, pos0, E, n, lane); k += n; }
                fp_fetch(G, k, l, kt);
            }
            if (l > -1) {
                const int t1 = l - k - 1;
                if (kt == 1) { emit_ins(G, which, pos0, E, 1, lane); nphase = 4; do_push = true; c_ao = ao + k; c_bo = bo + k + rmid + 1; c_M = t1; c_N = t1; c_lo = 0; c_u = min(t1, t2); c_tb = 2; c_te = 1; }
                else { emit_del(G, which, pos0, E, 1, lane); nphase = 5; do_push = true; c_ao = ao + k + 1; c_bo = bo + k + rmid; c_M = t1; c_N = t1; c_lo = max(-t1, t3); c_u = 0; c_tb = 1; c_te = 2; }
            } else {
                // last block (297-305)
                if (N - M > rmid) { emit_ins(G, which, pos0, E, 1, lane); const int t1 = k + rmid + 1; nphase = 6; do_push = true; c_ao = ao + k; c_bo = bo + t1; c_M = M - k; c_N = N - t1; c_lo = 0; c_u = min(N - t1, t2); c_tb = 2; c_te = te; }
                else if (N - M < rmid) { emit_del(G, which, pos0, E, 1, lane); const int t1 = M - (k + 1); nphase = 6; do_push = true; c_ao = ao + k + 1; c_bo = bo + k + rmid; c_M = t1; c_N = N - (k + rmid); c_lo = max(-t1, t3); c_u = 0; c_tb = 1; c_te = te; }
                else pop = true;
            }
        }
        if (lane == 0) { f[F_PHASE] = nphase; f[F_K] = k; f[F_L] = l; f[F_KT] = kt; }
        if (pop) sp--;
        if (do_push) push(c_ao, c_bo, c_M, c_N, c_lo, c_u, c_tb, c_te);
        wave_lds_sync();
    }
    *score_out = top_score;
    return 0;
}

// local_align's forward pass (src/localalign.c:88-133): best cell (first strict maximum in row-major order).  Every lane
// (diagonal) keeps the first row at which it reached its own maximum; the first strict maximum of the whole matrix is then
// the smallest such row among the lanes that hold the overall maximum, the leftmost lane within that row -- one reduction
// behind the loop instead of one per row.
template <int STEPS>
__device__ __forceinline__ void band_local_forward_t(const uint8_t* A1, const uint8_t* B1, int M, int N, int low, int up, int lane,
                                                     int& best, int& endi, int& endj)
{
    const int g = kGapOpen, h = kGapExt, m = g + h;
    const int band = up - low + 1;
    int leftd = low > 0 ? 1 : (up < 0 ? band : 1 - low), rightd = band;
    const int si = max(0, -up), ei = min(M, N - low);
    const int curd = lane + 1;
    int CC = kGapNegInf, DD = kGapNegInf;
    if (curd == leftd) { CC = 0; DD = -g; }
    else if (curd > leftd && curd <= rightd) { CC = 0; DD = -g; }
    int bv = 0, bi = INT_MAX;                                  // this diagonal's maximum (above 0) and the first row that has it
    const int hl = h * lane + g + kKeyBias;
    for (int i = si + 1; i <= ei; i++) {
        if (i > N - up) rightd--;
        if (leftd > 1) leftd--;
        const uint32_t ai = A1[i];
        const int ccn = wshl1(CC, kGapNegInf), ddn = wshl1(DD, kGapNegInf);
        const int co = ccn - m, dx = ddn - h;
        const int d = co > dx ? co : dx;
        const int jb = curd + low - 1 + i;
        const bool active = curd >= leftd && curd <= rightd;
        const uint32_t bch = (active && jb >= 1) ? B1[jb] : 256u;
        const int diag = CC + gw(ai, bch);
        const bool is_left = curd == leftd;
        int x = is_left ? (jb > 0 ? diag : co) : diag;
        if (d > x) x = d;
        const int x0 = max(x, 0);                              // a cell clamped at 0 opens gaps from 0
        const int key = active ? ekey_pack(x0, lane) : 0;
        const int ex = wshr1(band_scan_max<STEPS>(key), 0);
        const int e = (!is_left && ex != 0) ? (ex >> 6) - hl : kGapNegInf;
        const int c = max(x0, e);
        if (active) {
            CC = c; DD = d;
            if (c > bv) { bv = c; bi = i; }
        }
    }
    best = wave_max(bv);
    endi = si; endj = si + low;
    if (best > 0) {
        endi = wave_min(bv == best ? bi : INT_MAX);
        const int at = wave_min((bv == best && bi == endi) ? lane : 64);
        endj = at + 1 + low - 1 + endi;
    }
}
__device__ __forceinline__ void band_local_forward(const uint8_t* A1, const uint8_t* B1, int M, int N, int low, int up, int lane,
                                                   int& best, int& endi, int& endj)
{
    switch (band_scan_steps(up - low + 1)) {
    case 1: band_local_forward_t<1>(A1, B1, M, N, low, up, lane, best, endi, endj); break;
    case 2: band_local_forward_t<2>(A1, B1, M, N, low, up, lane, best, endi, endj); break;
    case 3: band_local_forward_t<3>(A1, B1, M, N, low, up, lane, best, endi, endj); break;
    case 4: band_local_forward_t<4>(A1, B1, M, N, low, up, lane, best, endi, endj); break;
    default: band_local_forward_t<6>(A1, B1, M, N, low, up, lane, best, endi, endj); break;
    }
}

// local_align's reverse pass (src/localalign.c:135-176) from the end cell: the first cell, rows upwards and diagonals
// downwards, whose score equals best.  lane = band - curd (the row is walked from its right end).
template <int STEPS>
__device__ __forceinline__ bool band_local_reverse_t(const uint8_t* A1, const uint8_t* B1, int N, int low, int up, int lane,
                                                     int best, int endi, int endj, int& starti, int& startj)
{
    const int g = kGapOpen, h = kGapExt, m = g + h;
    const int band = up - low + 1;
    int leftd = max(1, -endi - low + 1);
    int rightd = band - (up - (endj - endi));
    const int curd = band - lane;                              // lanes >= band: curd <= 0, never active
    int CC = kGapNegInf, DD = kGapNegInf;
    if (curd == rightd) { CC = 0; DD = -g; }
    else if (curd < rightd && curd >= leftd) { CC = -g - h * (rightd - curd); DD = CC - g; }
    const int hl = h * lane + g + kKeyBias;
    for (int i = endi; i >= 1; i--) {
        if (i + low <= 0) leftd++;
        if (rightd < band) rightd++;
        const uint32_t ai = A1[i];
        const int ccn = wshl1(CC, kGapNegInf), ddn = wshl1(DD, kGapNegInf);     // curd - 1 lives one lane up
        const int co = ccn - m, dx = ddn - h;
        const int d = co > dx ? co : dx;
        const int jb = curd + low - 1 + i;
        const bool active = curd >= leftd && curd <= rightd;
        const bool is_right = curd == rightd;
        const uint32_t bch = (active && jb >= 1 && jb <= N + 2) ? B1[jb] : 256u;
        const int diag = CC + gw(ai, bch);
        int x = is_right ? (jb <= N ? diag : co) : diag;
        if (d > x) x = d;
        const int key = active ? ekey_pack(x, lane) : 0;
        const int ex = wshr1(band_scan_max<STEPS>(key), 0);
        const int e = (!is_right && ex != 0) ? (ex >> 6) - hl : kGapNegInf;
        const int c = max(x, e);
        if (active) { CC = c; DD = d; }
        const uint64_t hits = __ballot(active && c == best);   // the lowest lane = the largest diagonal, the first one met
        if (hits) { const int hit = __builtin_ctzll(hits); starti = i; startj = (band - hit) + low - 1 + i; return true; }
    }
    return false;
}
__device__ __forceinline__ bool band_local_reverse(const uint8_t* A1, const uint8_t* B1, int N, int low, int up, int lane,
                                                   int best, int endi, int endj, int& starti, int& startj)
{
    switch (band_scan_steps(up - low + 1)) {
    case 1: return band_local_reverse_t<1>(A1, B1, N, low, up, lane, best, endi, endj, starti, startj);
    case 2: return band_local_reverse_t<2>(A1, B1, N, low, up, lane, best, endi, endj, starti, startj);
    case 3: return band_local_reverse_t<3>(A1, B1, N, low, up, lane, best, endi, endj, starti, startj);
    case 4: return band_local_reverse_t<4>(A1, B1, N, low, up, lane, best, endi, endj, starti, startj);
    default: return band_local_reverse_t<6>(A1, B1, N, low, up, lane, best, endi, endj, starti, startj);
    }
}

// fetch_cigar (src/globalalign.c:507-604) from the per-position form: [AP S] runs of = / X / I with the D ops in front of
// the positions that carry one, [tail S].  The tail clip keeps the reference's arithmetic (deletion lengths are counted
// into the consumed total, 541-595).  Returns 0 or IM_ST_OVERFLOW; ops land in G.ops[which].
__device__ __forceinline__ int band_make_cigar(BandLds& G, int which, int M, int starti, int endi, int lane)
{
    const int Ma = endi - starti + 1, pos0 = starti - 1;
    const uint8_t* kind = G.kind[which] + pos0;
    const uint16_t* dl = G.dlen[which] + pos0;
    // ops in front of / at position p: a D if dl[p] > 0, then a new run if the class changes (or a D sat in between)
    int nb = 0, dsum = 0;
    int cnt[4], isb[4], isd[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = 4 * lane + j;
        isd[j] = (p <= Ma) && dl[p] > 0;                        // p == Ma: a D behind the last base
        isb[j] = (p < Ma) && (p == 0 || kind[p] != kind[p - 1] || isd[j]);
        cnt[j] = isd[j] + isb[j];
        nb += cnt[j];
        if (isd[j]) dsum += dl[p];
    }
    const int inb = wave_scan_add(nb, lane);
    const int total = lane_get(inb, 63);
    const int dtot = wave_sum(dsum);
    const int lead = pos0 > 0 ? 1 : 0;
    const int numtotal = pos0 + Ma + dtot;
    const int tail = numtotal < M ? 1 : 0;
    const int n = lead + total + tail;
    if (n > IM_MAX_OPS) { BFAIL(G, 7); return IM_ST_OVERFLOW; }
    uint32_t* ops = G.ops[which];
    // run starts, in order, so that a run's length is the distance to the next start
    int32_t* bpos = G.tmp;
    {
        int r = 0;
        // rank among run starts only
        int nbs = isb[0] + isb[1] + isb[2] + isb[3];
        const int ib = wave_scan_add(nbs, lane);
        r = ib - nbs;
#pragma unroll
        for (int j = 0; j < 4; j++) if (isb[j]) bpos[r++] = 4 * lane + j;
        if (lane == 63) bpos[ib] = Ma;
    }
    wave_lds_sync();
    {
        int slot = lead + inb - nb, r;
        int nbs = isb[0] + isb[1] + isb[2] + isb[3];
        r = wave_scan_add(nbs, lane) - nbs;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = 4 * lane + j;
            if (isd[j]) ops[slot++] = ((uint32_t)dl[p] << 4) | IM_OP_D;
            if (isb[j]) {
                const uint32_t op = kind[p] == kPosEq ? IM_OP_EQ : kind[p] == kPosX ? IM_OP_X : IM_OP_I;
                ops[slot++] = ((uint32_t)(bpos[r + 1] - p) << 4) | op;
                r++;
            }
        }
    }
    if (lane == 0) {
        if (lead) ops[0] = ((uint32_t)pos0 << 4) | IM_OP_S;
        if (tail) ops[n - 1] = ((uint32_t)(M - numtotal) << 4) | IM_OP_S;
        G.nops[which] = n;
    }
    wave_lds_sync();
    return 0;
}

// attempt_band_alignment = local_align + ALIGN + fetch_cigar (src/alignment.c:343-391) across the wave.
// Window = contig[w0, w0 + N), read piece = rd[p0, p0 + M).
__device__ BandAln band_alignment(BandLds& G, const uint8_t* contig, const uint8_t* rd, int p0, int M, int w0, int N,
                                  int low_in, int up_in, int which, int lane)
{
    BandAln a; a.st = 0; a.r1 = a.r2 = a.q1 = a.q2 = 0;
    if (lane == 0) { G.nops[which] = 0; G.aln[which][0] = G.aln[which][1] = G.aln[which][2] = G.aln[which][3] = 0; }
    if (low_in > up_in || M <= 0 || N <= 0) { BFAIL(G, 101); a.st = IM_ST_ABORT; return a; }
    const int low = max(-M, low_in), up = min(N, up_in);
    const int band = up - low + 1;
    if (band < 1) { BFAIL(G, 102); a.st = IM_ST_ABORT; return a; }
    if (band > kGapMaxBand) { a.st = IM_ST_OVERFLOW; return a; }
    a.st = band_stage_window(G, contig, w0, N, M, low_in, up_in, lane);
    if (a.st) return a;
    const uint8_t* A1 = rd + p0 - 1;                            // A1[i], i = 1..M
    const uint8_t* B1 = G.wb - G.wb_base;                       // B1[j], staged slice
    int best, endi, endj, starti = 0, startj = 0;
    band_local_forward(A1, B1, M, N, low, up, lane, best, endi, endj);
    if (!band_local_reverse(A1, B1, N, low, up, lane, best, endi, endj, starti, startj)) return a;   // starti = startj = 0: rejected below in the reference too
    if (starti < 0 || starti > M || startj < 0 || startj > N) return a;
    if (endi - starti == 0 || endj - startj == 0) return a;
    // ALIGN on the located sub-strings (src/globalalign.c:333-401)
    const int Ma = endi - starti + 1, Na = endj - startj + 1;
    int lo2 = low - (startj - starti), up2 = up - (startj - starti);
    lo2 = min(max(-Ma, lo2), min(Na - Ma, 0));
    up2 = max(min(Na, up2), max(Na - Ma, 0));
    const uint8_t* A0 = A1 + starti - 1;                        // A0[i] = A1[starti - 1 + i]
    const uint8_t* B0 = B1 + startj - 1;
    const int pos0 = starti - 1;
    for (int p = lane; p <= M; p += 64) { if (p < M) G.kind[which][p] = kPosNone; G.dlen[which][p] = 0; }
    wave_lds_sync();
    int score;
    if (up2 - lo2 + 1 <= 1) {
        int s = 0;
        for (int p = lane; p < Ma; p += 64) {
            const bool eq = A0[p + 1] == B0[p + 1];
            G.kind[which][pos0 + p] = eq ? kPosEq : kPosX;
            s += eq ? kScoreMatch : kScoreMismatch;
        }
        score = wave_sum(s);
        wave_lds_sync();
    } else {
        // Equal lengths and at most three mismatches on the main diagonal: that alignment is the ONLY optimal one -- any other
        // path between the same corners holds at least one inserted and one deleted base (two gaps, 40 or more) and at most
        // Ma - 1 matches, so it scores below Ma - 41 < Ma - 11 * 3 -- and ALIGN, which returns an optimal alignment inside the
        // band (diagonal 0 always lies in it), can only come back with it.  Most band alignments of a split read end here
        // (the piece on one side of an indel wider than the band), without the divide-and-conquer passes.
        int mm = 4;
        if (Ma == Na) {
            int c = 0;
            for (int p = lane; p < Ma; p += 64) c += A0[p + 1] != B0[p + 1] ? 1 : 0;
            mm = wave_sum(c);
        }
        if (mm <= 3) {
            for (int p = lane; p < Ma; p += 64) G.kind[which][pos0 + p] = A0[p + 1] == B0[p + 1] ? kPosEq : kPosX;
            score = Ma * kScoreMatch + mm * (kScoreMismatch - kScoreMatch);
            wave_lds_sync();
        } else {
            const int st = band_global_align(G, which, A0, B0, Ma, Na, lo2, up2, pos0, lane, &score);
            if (st) { a.st = st; return a; }
            // (a CHECK_SCORE mismatch makes the reference print a line and carry on; nothing to do here)
        }
    }
    if (score <= 0) return a;
    a.st = band_make_cigar(G, which, M, starti, endi, lane);
    if (a.st) return a;
    a.r1 = startj + w0 - 1; a.r2 = endj + w0; a.q1 = starti + p0 - 1; a.q2 = endi + p0;
    if (lane == 0) { G.aln[which][0] = a.r1; G.aln[which][1] = a.r2; G.aln[which][2] = a.q1; G.aln[which][3] = a.q2; }
    return a;
}

// The '=' / 'X' read bases of a band alignment as bits, four read positions per lane in READ coordinates (position
// 4 * lane + j: bit j = '=', bit 4 + j = 'X'), taken from the per-position form while it is still in LDS (the second
// band search overwrites piece 1's).  Piece position p is read position p0 + p.
__device__ __forceinline__ uint32_t band_piece_bits(const BandLds& G, int which, int p0, int M, int lane)
{
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = 4 * lane + j - p0;
        if (p >= 0 && p < M) {
            const uint32_t kd = G.kind[which][p];
            v |= (kd == kPosEq ? 1u : 0u) << j;
            v |= (kd == kPosX ? 1u : 0u) << (4 + j);
        }
    }
    return v;
}

// find_best_del_candidate (src/alignment.c:306-339) over the per-position form.  A = the piece that starts at read
// offset 0 and ends at qa2, B = the piece on [qb1, L).  For the split i the reference counts the '=' and 'X' bases of
// A over read[0, i) and of B over read[i, L) (count_matches, 219-303): with E / X the running counts of the bits
// above that is EA(i) + EB(L) - EB(i), likewise for X -- two prefix counts per piece, every candidate of [qb1, qa2] at
// once (lane l holds i = 4l .. 4l + 3).  "Most matches, then fewest mismatches, then the first" is one packed
// maximum; the reference's early exit at a perfect candidate picks the same index (nothing in front of the first
// perfect candidate ties with it).
__device__ __forceinline__ int band_best_split(BandLds& G, uint32_t bitsA, int qa2, uint32_t bitsB, int qb1, int L, int lane, int* pindex)
{
    if (qb1 > qa2 || qa2 > L) { BFAIL(G, 8); return IM_ST_ABORT; }
    const int x0 = 4 * lane;
    uint32_t ea = 0, xa = 0, eb = 0, xb = 0;        // A only counts in front of qa2, B only from qb1 on
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool inA = x0 + j < qa2, inB = x0 + j >= qb1 && x0 + j < L;
        ea |= inA ? (bitsA & (1u << j)) : 0u;       xa |= inA ? ((bitsA >> 4) & (1u << j)) : 0u;
        eb |= inB ? (bitsB & (1u << j)) : 0u;       xb |= inB ? ((bitsB >> 4) & (1u << j)) : 0u;
    }
    // one scan carries the four counts (each <= 255)
    const int mine = __popc(ea) | (__popc(xa) << 8) | (__popc(eb) << 16) | (__popc(xb) << 24);
    const int incl = wave_scan_add(mine, lane);
    const int tot = lane_get(incl, 63), excl = incl - mine;
    const int totEB = (tot >> 16) & 255, totXB = (tot >> 24) & 255;
    int best = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = x0 + j;
        const uint32_t below = (1u << j) - 1u;
        const int EA = (excl & 255) + __popc(ea & below), XA = ((excl >> 8) & 255) + __popc(xa & below);
        const int EB = ((excl >> 16) & 255) + __popc(eb & below), XB = ((excl >> 24) & 255) + __popc(xb & below);
        const int matches = EA + totEB - EB, mm = XA + totXB - XB;
        if (i >= qb1 && i <= qa2) best = max(best, (matches << 16) | ((255 - mm) << 8) | (255 - i));
    }
    best = wave_max(best);
    if (best < 0) { BFAIL(G, 10); return IM_ST_ABORT; }          // forceassert(index != -1)
    *pindex = 255 - (best & 255);
    return 0;
}

// update_readsegs (src/readaln.c:348-458) + one evidence per D / I segment (new_evidence, src/evidence.c:4-34, with the flank
// reductions of src/variant.c:217-290,704-775): lane = CIGAR op.  c1 / c2 live in LDS, the result goes to the record.
__device__ int band_build_result(im_read_result* out, BandLds& G, int r1, const uint32_t* c1, int n1, int index, int q2, int r2,
                                 const uint32_t* c2, int n2, int lane, const RealignArgs& A, int cidx)
{
    // --- piece 1 up to read offset `index` (362-385)
    const bool h1 = lane < n1;
    const int op1 = h1 ? (int)(c1[lane] & 15u) : 0, len1 = h1 ? (int)(c1[lane] >> 4) : 0;
    if (wave_max((h1 && len1 <= 0) ? 1 : 0)) { BFAIL(G, 11); return IM_ST_ABORT; }            // forceassert(oplen > 0)
    const int J1 = wave_scan_add((h1 && op1 != IM_OP_D) ? len1 : 0, lane);
    const int tstop = wave_min((h1 && J1 > index) ? lane : 64);             // first op that runs past index: cut there
    int e1 = 0;                                                             // emitted length of this lane's op
    if (h1 && lane < tstop) e1 = len1;
    else if (h1 && lane == tstop) { const int part = index - (J1 - len1); e1 = part > 0 ? part : 0; }
    const bool refc1 = op1 == IM_OP_EQ || op1 == IM_OP_X || op1 == IM_OP_D || op1 == IM_OP_M;
    const int refindx = r1 + wave_sum(refc1 ? e1 : 0);
    const int nA = wave_sum(e1 > 0 ? 1 : 0);
    // --- where piece 2 takes over (389-421)
    const bool h2 = lane < n2;
    const int op2 = h2 ? (int)(c2[lane] & 15u) : 0, len2 = h2 ? (int)(c2[lane] >> 4) : 0;
    const int J2 = wave_scan_add((h2 && op2 != IM_OP_D) ? len2 : 0, lane);
    int rindex = r2, nextindex = index, ilen = 0;
    if (index >= q2) {
        int off = 0;
        if (h2 && op2 != IM_OP_I) {
            if (J2 > q2 && J2 <= index) { off = len2; if (J2 - len2 <= q2) off -= q2 - (J2 - len2); }
            else if (J2 > index) { if (J2 - len2 <= index) off = index - (J2 - len2); }
        }
        rindex = r2 + wave_sum(off);
    } else { ilen = q2 - index; nextindex = q2; }
    const int dlen = refindx < rindex ? rindex - refindx : 0;
    // --- piece 2 from read offset nextindex (432-453)
    const int tgo = wave_min((h2 && J2 > nextindex) ? lane : 64);
    int e2 = 0;
    if (h2 && lane == tgo) e2 = J2 - nextindex;
    else if (h2 && lane > tgo) e2 = len2;
    const int nB = wave_sum(e2 > 0 || (h2 && lane > tgo) ? 1 : 0);
    const int n = nA + (ilen > 0 ? 1 : 0) + (dlen > 0 ? 1 : 0) + nB;
    if (n > IM_MAX_OPS) { BFAIL(G, 12); return IM_ST_OVERFLOW; }
    // --- assemble the final list in LDS (G.tmp), then one lane per final op
    uint32_t* fin = reinterpret_cast<uint32_t*>(G.tmp);
    {
        const int ra = wave_scan_add(e1 > 0 ? 1 : 0, lane) - (e1 > 0 ? 1 : 0);
        if (e1 > 0) fin[ra] = ((uint32_t)e1 << 4) | (uint32_t)op1;
        int at = nA;
        if (lane == 0) {
            if (ilen > 0) fin[at++] = ((uint32_t)ilen << 4) | IM_OP_I;
            if (dlen > 0) fin[at++] = ((uint32_t)dlen << 4) | IM_OP_D;
        }
        at = nA + (ilen > 0 ? 1 : 0) + (dlen > 0 ? 1 : 0);
        const bool emitB = h2 && lane >= tgo && tgo < 64;
        const int rb = wave_scan_add(emitB ? 1 : 0, lane) - (emitB ? 1 : 0);
        if (emitB) fin[at + rb] = ((uint32_t)e2 << 4) | (uint32_t)op2;
    }
    wave_lds_sync();
    const bool hf = lane < n;
    const uint32_t w = hf ? fin[lane] : 0u;
    const int op = (int)(w & 15u), len = (int)(w >> 4);
    if (hf) out->ops[lane] = w;
    const bool refc = hf && (op == IM_OP_EQ || op == IM_OP_X || op == IM_OP_D);
    const int refpos = r1 + wave_scan_add(refc ? len : 0, lane) - (refc ? len : 0);
    const int readpos = wave_scan_add((hf && op != IM_OP_D) ? len : 0, lane) - ((hf && op != IM_OP_D) ? len : 0);
    const int fl = hf && (op == IM_OP_EQ || op == IM_OP_X || op == IM_OP_I) ? len : 0;
    const int ndp = hf && (op == IM_OP_X || op == IM_OP_I || op == IM_OP_D) ? len : 0;
    const int ndf = ndp + ((hf && op == IM_OP_S) ? len : 0);
    const int fl_in = wave_scan_add(fl, lane), fl_tot = lane_get(fl_in, 63);
    const int ndp_tot = wave_sum(ndp), ndf_tot = wave_sum(ndf);
    const bool indel = hf && (op == IM_OP_D || op == IM_OP_I);
    const int erank = wave_scan_add(indel ? 1 : 0, lane) - (indel ? 1 : 0);
    const int ne = wave_sum(indel ? 1 : 0);
    if (ne > IM_MAX_EV) { BFAIL(G, 13); return IM_ST_OVERFLOW; }
    if (indel) {
        im_evidence* e = &out->ev[erank];
        e->cls = op == IM_OP_D ? IM_CLS_DELETION : IM_CLS_INSERTION;
        e->b1 = refpos; e->b2 = op == IM_OP_D ? refpos + len : refpos;
        e->seg = lane; e->read_off = readpos;
        e->lflank = fl_in - fl; e->rflank = fl_tot - fl_in;
        e->nd_print = ndp_tot - ndp; e->nd_filter = ndf_tot - ndf;
        if (A.batch.ev_cls) {
            const int64_t sl = (int64_t)cidx * IM_MAX_EV + erank;
            A.batch.ev_cls[sl] = e->cls; A.batch.ev_b1[sl] = e->b1; A.batch.ev_b2[sl] = e->b2;
        }
    }
    if (ne > 0 && A.batch.ev_cls && lane < IM_MAX_EV && lane >= ne) {
        const int64_t sl = (int64_t)cidx * IM_MAX_EV + lane;
        A.batch.ev_cls[sl] = -1; A.batch.ev_b1[sl] = 0; A.batch.ev_b2[sl] = 0;
    }
    if (lane == 0) { out->ref_start = r1; out->n_ops = n; out->n_ev = ne; }
    return ne > 0 ? IM_ST_EVIDENCE : IM_ST_NONE;
}

template <bool DIRECT>
__global__ __launch_bounds__(64, 5) void realign_band_kernel(RealignArgs A)
{
    // The band alignment's state lives where the vote's histogram and k-mer table are (both dead between two band
    // searches; band_search<0, ..> rebuilds them from scratch): 6.4 KB of LDS per wave instead of 12, twice the waves per CU.
    __shared__ WaveLds s;
    static_assert(offsetof(WaveLds, tbl) == sizeof(s.diag), "diag and tbl are contiguous");
    constexpr bool kOverlay = sizeof(BandLds) <= sizeof(s.diag) + sizeof(s.tbl);     // a smaller histogram (IM_DIAG_CHUNK) leaves no room: own block then
    __shared__ uint32_t band_own[kOverlay ? 1 : (sizeof(BandLds) + 3) / 4];
    BandLds& G = *reinterpret_cast<BandLds*>(kOverlay ? s.diag : band_own);
    __shared__ uint32_t c1_keep[IM_MAX_OPS + 4];            // the first piece's CIGAR outlives the second band search
    const int lane = threadIdx.x;
    const uint32_t k = A.P.klength, g = A.P.numgaps, eth = A.P.ethreshold;
    const int n_reads = A.n_dev ? min(*A.n_dev, A.batch.n) : A.batch.n;
    for (int c = blockIdx.x; c < n_reads; c += gridDim.x) {
        im_read_result* out = &A.batch.out[c];
        const int64_t off = uni64(A.batch.base_off[c]);
        const int64_t Lraw = uni(A.batch.read_len[c]);
        const int tid = uni(A.batch.tid[c]);
        const int anchor = uni(A.batch.anchor[c]);
        const int R = uni(A.batch.range_max[c]);
        if (lane < 16) reinterpret_cast<uint32_t*>(&out->band[0])[lane] = 0u;
        if (lane < 7) out->reserved[lane] = 0;
        if (!A.keep_slots) write_slots(A, c, 0, -1, 0, 0, lane);
        wave_lds_sync();
        if (Lraw <= 0 || Lraw > kShortRead || tid < 0 || tid >= A.ref.n_contigs || (off & 3)) {
            finish(out, (Lraw > kShortRead || (off & 3)) ? IM_ST_UNSUPPORTED : IM_ST_ABORT, 0, lane);
            continue;
        }
        const int L = (int)Lraw;
        const uint8_t* contig = A.ref.ascii + uni64(A.ref.asc_off[tid]);
        const uint8_t* pk = A.ref.pk + uni64(A.ref.pk_off[tid]);
        const int clen = uni(A.ref.len[tid]);
        {
            uint32_t v = 0;
            if (4 * lane < L) v = *reinterpret_cast<const uint32_t*>(A.batch.bases + off + 4 * lane);
            const int rem = L - 4 * lane;
            if (rem < 4) v &= (rem <= 0) ? 0u : ((1u << (8 * rem)) - 1u);
            s.rd[lane] = v;
            if (lane < 4) s.rd[64 + lane] = 0u;
        }
        wave_lds_sync();
        const uint8_t* rd = reinterpret_cast<const uint8_t*>(s.rd);
        int distance = R;
        const int left1  = anchor >= distance ? anchor - distance : 0;
        const int right1 = clen < (anchor + distance) ? clen : anchor + distance;
        distance = R + (int)A.P.maxdelsize;
        const int left2  = anchor >= distance ? anchor - distance : 0;
        const int right2 = clen < (anchor + distance) ? clen : anchor + distance;
        if (!(anchor >= left1 && anchor >= left2 && anchor <= right1 && anchor <= right2 && left2 >= 0 && right2 > 0)) {
            finish(out, IM_ST_ABORT, 0, lane); continue;
        }
        // piece 1: the whole read in [left1, right1)
        const Band b1 = band_search<0, DIRECT>(s, pk, contig, (uint32_t)left1, (uint32_t)right1, (uint32_t)anchor, 0u, (uint32_t)L, k, g, lane, 0u, nullptr IM_STAMP_PASS(16));
        if (b1.st) { finish(out, b1.st, 1, lane); continue; }
        const int up1 = ((uint32_t)L < k) ? b1.low : b1.low + (int)g;      // read shorter than k: low == up (408-412)
        const BandAln a1 = band_alignment(G, contig, rd, 0, L, left1, right1 - left1, b1.low, up1, 0, lane);
        const int r1 = a1.r1, r2 = a1.r2, q1 = a1.q1, q2 = a1.q2;
        if (lane == 0) {
            im_band_aln* o = &out->band[0];
            o->r1 = r1; o->r2 = r2; o->q1 = q1; o->q2 = q2; o->low = b1.low; o->votes = b1.votes; o->win_bytes = b1.win; o->piece_bytes = b1.piece;
        }
        if (a1.st) { if (lane == 0) out->reserved[6] = G.tmp[IM_MAX_OPS + 7]; finish(out, a1.st, 1, lane); continue; }
        if (q1 == q2) { finish(out, IM_ST_NONE, 1, lane); continue; }
        wave_lds_sync();
        const int n1 = uni(G.nops[0]);
        if (lane < n1) c1_keep[lane] = G.ops[0][lane];
        wave_lds_sync();
        const uint32_t* c1 = c1_keep;
        const uint32_t bits1 = band_piece_bits(G, 0, 0, L, lane);
        if (q1 == 0 && q2 == L) {       // whole read aligned: evidence only from I/D ops inside the CIGAR (575-582)
            const int rs = band_build_result(out, G, r1, c1, n1, L, 0, -1, c1, 0, lane, A, c);
            finish(out, rs, 1, lane);
            continue;
        }
        // leading / trailing '=' runs of the first CIGAR (585-599): runs are maximal, so it is the first / last op behind a clip
        uint32_t f = 0, l = 0;
        {
            const int i0 = (n1 > 0 && (c1[0] & 15u) == IM_OP_S) ? 1 : 0;
            if (i0 < n1 && (c1[i0] & 15u) == IM_OP_EQ) f = c1[i0] >> 4;
            const int i1 = (n1 > 0 && (c1[n1 - 1] & 15u) == IM_OP_S) ? n1 - 2 : n1 - 1;
            if (i1 >= 0 && (c1[i1] & 15u) == IM_OP_EQ) l = c1[i1] >> 4;
            f = (uint32_t)uni((int)f); l = (uint32_t)uni((int)l);
        }
        const uint32_t uL = (uint32_t)L;
        uint32_t w0 = 0, w1 = 0, anc = 0, p0 = 0, p1 = 0; bool want_tail = false; int none = 0, abortc = 0;
        if (r1 > anchor) {
            if (q1 == 0) {
                if (!(uL > f)) abortc = 1;
                else if ((uL - f) < eth || ((uint32_t)right2 - (uint32_t)r1 - f) < eth) none = 1;
                w0 = (uint32_t)r1 + f; w1 = (uint32_t)right2; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
            } else if (q2 == L) {
                if (!(uL > l)) abortc = 1;
                else if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)anchor) < eth) none = 1;
                w0 = (uint32_t)anchor; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l;
            } else none = 1;
        } else if (r1 < anchor) {
            if (r2 >= anchor) none = 1;
            else if (q1 == 0) {
                if (!(uL > f)) abortc = 1;
                else if ((uL - f) < eth || ((uint32_t)anchor - (uint32_t)r1 - f) < eth) none = 1;
                w0 = (uint32_t)r1 + f; w1 = (uint32_t)anchor; anc = (uint32_t)r1; p0 = f; p1 = uL; want_tail = true;
            } else if (q2 == L) {
                if (!(uL > l)) abortc = 1;
                else if ((uL - l) < eth || ((uint32_t)r2 - l - (uint32_t)left2) < eth) none = 1;
                w0 = (uint32_t)left2; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = uL - l;
            } else none = 1;
        } else none = 1;
        if (abortc) { finish(out, IM_ST_ABORT, 1, lane); continue; }
        if (none) { finish(out, IM_ST_NONE, 1, lane); continue; }
        if ((int32_t)(w1 - w0) <= 0) { finish(out, IM_ST_ABORT, 1, lane); continue; }
        // piece 2
        const Band b2 = band_search<0, DIRECT>(s, pk, contig, w0, w1, anc, p0, p1, k, g, lane, 0u, nullptr IM_STAMP_PASS(21));
        if (b2.st) { finish(out, b2.st, 2, lane); continue; }
        const int up2 = ((p1 - p0) < k) ? b2.low : b2.low + (int)g;
        const BandAln a2 = band_alignment(G, contig, rd, (int)p0, (int)(p1 - p0), (int)w0, (int)(w1 - w0), b2.low, up2, 1, lane);
        if (lane == 0) {
            im_band_aln* o = &out->band[1];
            o->r1 = a2.r1; o->r2 = a2.r2; o->q1 = a2.q1; o->q2 = a2.q2;
            o->low = b2.low; o->votes = b2.votes; o->win_bytes = b2.win; o->piece_bytes = b2.piece;
        }
        int st = a2.st;
        if (st == 0) {
            const int r3 = a2.r1, r4 = a2.r2, q3 = a2.q1, q4 = a2.q2;
            uint32_t* c2 = G.ops[1]; int n2 = uni(G.nops[1]);
            st = -100;                                      // "go on to combine"
            if (want_tail) { if (q4 != L || q3 == q4) st = IM_ST_NONE; }
            else           { if (q3 != 0 || q3 == q4) st = IM_ST_NONE; }
            if (st == -100) {
                // add_prefix_soft_clip / add_suffix_soft_clip (src/alignment.c:478-532)
                if (want_tail && f > 0) {
                    if (n2 > 0 && (c2[0] & 15u) == IM_OP_S) { if (lane == 0) c2[0] = (((c2[0] >> 4) + f) << 4) | IM_OP_S; }
                    else if (n2 >= IM_MAX_OPS) st = IM_ST_OVERFLOW;
                    else {
                        const uint32_t mine = lane < n2 ? c2[lane] : 0u;
                        wave_lds_sync();
                        if (lane < n2) c2[lane + 1] = mine;
                        if (lane == 0) c2[0] = (f << 4) | IM_OP_S;
                        n2++;
                    }
                } else if (!want_tail && l > 0) {
                    if (n2 <= 0) st = IM_ST_ABORT;
                    else if ((c2[n2 - 1] & 15u) == IM_OP_S) { if (lane == 0) c2[n2 - 1] = (((c2[n2 - 1] >> 4) + l) << 4) | IM_OP_S; }
                    else if (n2 >= IM_MAX_OPS) st = IM_ST_OVERFLOW;
                    else { if (lane == 0) c2[n2] = (l << 4) | IM_OP_S; n2++; }
                }
                wave_lds_sync();
            }
            if (st == -100) {
                if (!(q1 < q2 && q3 < q4)) { BFAIL(G, 103); st = IM_ST_ABORT; }
                else {
                    int index = -1;
                    const uint32_t bits2 = band_piece_bits(G, 1, (int)p0, (int)(p1 - p0), lane);
                    // find_best_del_candidate asserts that its first piece starts at read offset 0 (314-315); its second piece
                    // ends at L by the accept conditions above
                    if (q1 > q3 && q1 <= q4) {
                        st = (q3 != 0 || q2 != L) ? IM_ST_ABORT : band_best_split(G, bits2, q4, bits1, q1, L, lane, &index);
                        if (st == 0) st = band_build_result(out, G, r3, c2, n2, index, q1, r1, c1, n1, lane, A, c);
                    } else if (q3 > q1 && q3 <= q2) {
                        st = (q1 != 0 || q4 != L) ? IM_ST_ABORT : band_best_split(G, bits1, q2, bits2, q3, L, lane, &index);
                        if (st == 0) st = band_build_result(out, G, r1, c1, n1, index, q3, r3, c2, n2, lane, A, c);
                    } else if (q1 > q4 && r1 == r4) st = band_build_result(out, G, r3, c2, n2, q4, q1, r1, c1, n1, lane, A, c);
                    else if (q3 > q2 && r2 == r3) st = band_build_result(out, G, r1, c1, n1, q2, q3, r3, c2, n2, lane, A, c);
                    else st = IM_ST_NONE;
                }
            }
        }
        if (st < 0 && lane == 0) out->reserved[6] = G.tmp[IM_MAX_OPS + 7];
        finish(out, st, 2, lane);
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
        uint64_t v = 0;                 // base 32w + n in bits 2n, 2n+1: in memory, 4 bases per byte, first base low
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (uint64_t)code2((d[i] >> (8 * j)) & 255u) << (2 * (4 * i + j));
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
    // 6352 B of LDS per one-wave workgroup (6400 allocated) -> 25 fit a CU; <= 80 VGPRs -> 6 waves per SIMD.
    int64_t want = (int64_t)n_cu * IM_BLOCKS_PER_CU;
    int64_t need = ((int64_t)a.batch.n + 7) / 8 * 8;
    if (a.P.numgaps > 0) {
        // one wave per read; the band kernel holds more LDS per wave than the single-diagonal one
        // up to 128 one-wave workgroups per CU: reads differ widely in cost (one or two band alignments, with or without the
        // divide-and-conquer passes), a short stride leaves the balancing to the dispatcher (measured: 24 -> 128 per CU, +19 %)
        int gg = (int)(need < (int64_t)n_cu * 128 ? need : (int64_t)n_cu * 128);
        if (a.P.klength <= (uint32_t)kDirectMaxK)
            hipLaunchKernelGGL((realign_band_kernel<true>), dim3(gg), dim3(64), 0, stream, a);
        else
            hipLaunchKernelGGL((realign_band_kernel<false>), dim3(gg), dim3(64), 0, stream, a);
    }
    else {
        // one read per wave: a batch beyond the largest grid goes out in slices
        const int64_t cap = want / 8 * 8;
        for (int64_t first = 0; first < (int64_t)a.batch.n; first += cap) {
            RealignArgs b = a;
            b.first = (int32_t)first;
            const int64_t rest = ((int64_t)a.batch.n - first + 7) / 8 * 8;
            const int g = (int)(rest < cap ? rest : cap);
            if (a.P.klength == 6)
                hipLaunchKernelGGL((realign_kernel<6, true>), dim3(g), dim3(64), 0, stream, b);     // reference defaults
            else if (a.P.klength <= (uint32_t)kDirectMaxK)
                hipLaunchKernelGGL((realign_kernel<-1, true>), dim3(g), dim3(64), 0, stream, b);    // any other k <= 6: the same clean-table protocol, mask at run time
            else if (a.P.klength <= 13u)
                hipLaunchKernelGGL((realign_kernel<7, true>), dim3(g), dim3(64), 0, stream, b);     // k = 7..13: table by the first six bases, whole k-mers checked at the vote
            else
                hipLaunchKernelGGL((realign_kernel<0, false>), dim3(g), dim3(64), 0, stream, b);     // k = 14, 15: the hash table
        }
    }
    return hipGetLastError();
}

}  // namespace im
