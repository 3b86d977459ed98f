// im_triage.hip -- fetch_func's per-record decisions for a whole chunk of BAM records on gfx950.
//
// Replaces, for every delivered record at once (src/indelminer.c:339-515):
//   the flag / pairing filters                                   348-366
//   the read-group -> range[] lookup (must_find_hashtable)       369-376
//   the three candidate cases and their mapping-quality gates    384-515
//   new_unaligned_readaln's 4-bit -> ASCII decode + revcomp      src/readaln.c:242-267, src/sequences.c:204-220
//   check_variants (CIGAR-derived evidence)                      285-337
//   the read filter and match segments of the DP= pileup         src/shared.c:160-176, bam_pileup.c:171-172
// Discordant pairs (516-615) need the host's pair table; they are only labelled here.
//
// Shape.  HBM streaming: every byte of a record is read once (core, CIGAR, aux by the record's
// own lane; the packed bases of the ~4 % candidates by the whole wave), nothing is staged.
// Three launches per chunk, all one lane per record in 256-record workgroups:
//   classify   class + read-group range per record, per-workgroup candidate / byte totals,
//              pileup segments scattered into the genome-wide difference array
//   scan       one workgroup: exclusive scan of the workgroup totals on top of the running
//              candidate / byte counters (candidates are appended in RECORD ORDER -- arrival
//              order decides tie order inside clusters)
//   emit       candidate index = running base + position in the workgroup; the owner lane
//              writes the per-read scalars and the CIGAR-derived evidence slots, then the wave
//              decodes each of its candidates' bases together, four bases per lane
// Algorithmic bytes: the record bytes in + (4-byte padded read + 24 B of scalars + 48 B of
// slots) per candidate out + 1 B class per record.

#include "im_device.hpp"

namespace im {
namespace {

constexpr int kTriBlock = 256;
constexpr int kTriGroup = 32;           // workgroups that count their publication into one word (see the classify kernel's tail)
constexpr int kTriGroupsMax = 2048;

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

struct RecView {
    const uint8_t* p;       // record start (the 32-byte core)
    uint32_t len;           // bytes of the record
    int32_t tid, pos, mtid, mpos, isize, l_seq;
    uint32_t l_qname, mapq, n_cigar, flag;
    uint32_t o_cigar, o_seq, o_aux;
    uint32_t cig[4];        // the first four CIGAR words, loaded together with the aux window
    bool ok;
};

constexpr int kAuxWin = 24;     // bytes of the aux area a lane holds in LDS at a time: the tag walk slides the window along (aux_slide); more LDS
                                // would cost the kernel its sixth workgroup per CU

__device__ __forceinline__ RecView view_record(const uint8_t* raw, uint32_t off, uint32_t end)
{
    RecView r;
    r.p = raw + off; r.len = end - off; r.ok = false;
    r.tid = r.pos = r.mtid = r.mpos = r.isize = r.l_seq = 0;
    r.l_qname = r.mapq = r.n_cigar = r.flag = 0; r.o_cigar = r.o_seq = r.o_aux = 0;
    r.cig[0] = r.cig[1] = r.cig[2] = r.cig[3] = 0;
    if (end < off || r.len < 32u) return r;
    const uint32_t* c = reinterpret_cast<const uint32_t*>(r.p);     // 4-byte aligned by contract
    r.tid = (int32_t)c[0]; r.pos = (int32_t)c[1];
    const uint32_t w2 = c[2], w3 = c[3];
    r.l_qname = w2 & 255u; r.mapq = (w2 >> 8) & 255u;
    r.n_cigar = w3 & 0xFFFFu; r.flag = w3 >> 16;
    r.l_seq = (int32_t)c[4]; r.mtid = (int32_t)c[5]; r.mpos = (int32_t)c[6]; r.isize = (int32_t)c[7];
    if (r.l_seq < 0) return r;
    r.o_cigar = 32u + r.l_qname;
    r.o_seq = r.o_cigar + 4u * r.n_cigar;
    // a record delivered without its base qualities says so in its bin field (include/indelminer_amd.h, im_dev_records)
    const bool no_qual = (w2 >> 16) == 0xFFFFu;
    const uint64_t o_aux = (uint64_t)r.o_seq + (((uint64_t)r.l_seq + 1u) >> 1) + (no_qual ? 0ull : (uint64_t)r.l_seq);
    if (o_aux > r.len) return r;
    r.o_aux = (uint32_t)o_aux;
    r.ok = true;
    return r;
}

// CIGAR word k: the first four travel in registers (one round trip with the aux window), the rest come from memory
__device__ __forceinline__ uint32_t cigar_word(const RecView& r, uint32_t k)
{
    return k == 0u ? r.cig[0] : k == 1u ? r.cig[1] : k == 2u ? r.cig[2] : k == 3u ? r.cig[3] : ld_u32(r.p + r.o_cigar + 4u * k);
}

__device__ __forceinline__ int aux_size(uint32_t t)
{
    switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    default: return 0;
    }
}

// a byte of the record at offset o: the lane's aux window (LDS) when it covers o, memory otherwise
struct AuxWin { uint32_t* lds; uint32_t o0; };
__device__ __forceinline__ uint32_t rec_byte(const RecView& r, const AuxWin& w, uint32_t o)
{
    const uint32_t d = o - w.o0;
    return d < (uint32_t)kAuxWin ? reinterpret_cast<const uint8_t*>(w.lds)[d] : r.p[o];
}
// The window moved to offset o: six dword loads by the lane that needs them.  What an aligner writes in front of RG and MQ
// (NM MD AS XS MC ...: 30-100 bytes) used to be walked through memory byte by byte behind the first 24 -- every byte a 64-line
// gather: records with such fields took classify from 25 to 77 us per 300 000 (profiles/r04_m_*).  Reads past the record stay
// inside the chunk buffer (>= 64 spare bytes behind the last record).
__device__ __forceinline__ void aux_slide(const RecView& r, AuxWin& w, uint32_t o)
{
    w.o0 = o;
#pragma unroll
    for (int k = 0; k < kAuxWin / 4; k++) w.lds[k] = ld_u32(r.p + o + 4u * k);
}
// the byte at o for a walk that only moves forward: the window follows
__device__ __forceinline__ uint32_t walk_byte(const RecView& r, AuxWin& w, uint32_t o)
{
    if (o - w.o0 >= (uint32_t)kAuxWin) aux_slide(r, w, o);
    return reinterpret_cast<const uint8_t*>(w.lds)[o - w.o0];
}
// [o, o + need) inside the window (need <= kAuxWin)
__device__ __forceinline__ void aux_cover(const RecView& r, AuxWin& w, uint32_t o, uint32_t need)
{
    if (!(o >= w.o0 && o + need <= w.o0 + (uint32_t)kAuxWin)) aux_slide(r, w, o);
}

// bam_aux_get for RG and MQ in one walk (bam_aux.c:27-54): offsets of the TYPE byte of the first
// occurrence, 0 = absent.  The walk stops where samtools' would (unknown type, truncated B array).
__device__ __forceinline__ void find_rg_mq(const RecView& r, AuxWin& w, uint32_t& o_rg, uint32_t& o_mq)
{
    o_rg = 0; o_mq = 0;
    uint32_t s = r.o_aux;
    const uint32_t end = r.len;
    while (s + 4u <= end) {      // a tail of < 4 bytes is alignment padding (include/indelminer_amd.h, im_dev_records)
        aux_cover(r, w, s, 8u);  // tag, type and what a B array's header takes
        const uint32_t t0 = rec_byte(r, w, s), t1 = rec_byte(r, w, s + 1), type = rec_byte(r, w, s + 2);
        if (t0 == 'R' && t1 == 'G' && !o_rg) o_rg = s + 2u;
        if (t0 == 'M' && t1 == 'Q' && !o_mq) o_mq = s + 2u;
        if (o_rg && o_mq) return;
        s += 3u;
        if (type == 'Z' || type == 'H') { while (s < end && walk_byte(r, w, s)) s++; s++; }
        else if (type == 'B') {
            if (s + 5u > end) return;
            const int sz = aux_size(rec_byte(r, w, s));
            const uint32_t cnt = rec_byte(r, w, s + 1) | (rec_byte(r, w, s + 2) << 8) | (rec_byte(r, w, s + 3) << 16) | (rec_byte(r, w, s + 4) << 24);
            const uint64_t ns = (uint64_t)s + 5u + (uint64_t)sz * cnt;
            if (ns > end) return;
            s = (uint32_t)ns;
        } else {
            const int sz = aux_size(type);
            if (sz == 0) return;
            s += (uint32_t)sz;
        }
    }
}

// bam_aux2i (bam_aux.c:163-174)
__device__ __forceinline__ int32_t aux_int(const RecView& r, const AuxWin& w, uint32_t o)
{
    const uint32_t type = rec_byte(r, w, o);
    if (o + 1u >= r.len) return 0;
    const uint32_t b0 = rec_byte(r, w, o + 1);
    switch (type) {
    case 'c': return (int32_t)(int8_t)b0;
    case 'C': return (int32_t)b0;
    case 's': return (o + 3u <= r.len) ? (int32_t)(int16_t)(b0 | (rec_byte(r, w, o + 2) << 8)) : 0;
    case 'S': return (o + 3u <= r.len) ? (int32_t)(b0 | (rec_byte(r, w, o + 2) << 8)) : 0;
    case 'i': case 'I':
        return (o + 5u <= r.len) ? (int32_t)(b0 | (rec_byte(r, w, o + 2) << 8) | (rec_byte(r, w, o + 3) << 16) | (rec_byte(r, w, o + 4) << 24)) : 0;
    default: return 0;
    }
}

// The insert-length table as one blob (LDS copy when it is small, else the device original):
// [bin_start 17][name_off n][name_len n][range_max n][names]
struct RgView {
    const int32_t* bin_start; const int32_t* name_off; const int32_t* name_len; const int32_t* range_max; const uint8_t* names;
};
__device__ __forceinline__ RgView rg_view(const uint8_t* blob, int32_t n)
{
    RgView v;
    const int32_t* w = reinterpret_cast<const int32_t*>(blob);
    const int32_t m = n > 0 ? n : 1;
    v.bin_start = w; v.name_off = w + 20; v.name_len = w + 20 + m; v.range_max = w + 20 + 2 * m;
    v.names = blob + 4 * (20 + 3 * m);
    return v;
}

// must_find_hashtable(insertlengths, rgname, strlen(rgname)) (src/indelminer.c:374-376): DJB2 over the
// bytes back to front (src/hashfunc.c:23-30), 16 bins, the chain walked head to tail, strncmp prefix
// match, LAST hit wins (src/hashtable.c:62-81).  Returns false when the reference would exit.
template <typename NameAt>
__device__ __forceinline__ bool rg_lookup(const RgView& T, NameAt name_at, uint32_t len, int32_t& range_max)
{
    uint32_t h = 5381u;
    for (int i = (int)len - 1; i >= 0; i--) h += (h << 5) + (uint32_t)(int32_t)(int8_t)name_at((uint32_t)i);
    const uint32_t bin = h & 15u;
    bool hit = false;
    for (int32_t e = T.bin_start[bin]; e < T.bin_start[bin + 1]; e++) {
        const uint32_t el = (uint32_t)T.name_len[e];
        if (el < len) continue;                      // the stored name ends first: strncmp sees NUL != byte
        const uint8_t* en = T.names + T.name_off[e];
        bool same = true;
        for (uint32_t i = 0; i < len && same; i++) same = en[i] == name_at(i);
        if (same) { hit = true; range_max = T.range_max[e]; }
    }
    return hit;
}

// new_readaln (src/readaln.c:186-240) runs for aligned proper pairs whose mate is aligned (src/indelminer.c:425), after
// the filters of 348-366: whether this record gets that far, from the core alone
__device__ __forceinline__ bool reaches_new_readaln(const RecView& r)
{
    const uint32_t f = r.flag;
    return r.ok && !(f & (0x100u | 0x200u | 0x400u | 0x800u)) && (f & 0x1u) && (f & 0x2u) && !(f & (0x4u | 0x8u)) && r.tid == r.mtid;
}

// How many read bases new_readaln decodes: the query bases of the CIGAR, taken where the CIGAR says -- also past l_seq --
// but not past the record.
__device__ __forceinline__ uint32_t bases_in_cigar_reach(const RecView& r)
{
    uint64_t qtot = 0;
    for (uint32_t i = 0; i < r.n_cigar; i++) {
        const uint32_t cw = cigar_word(r, i), op = cw & 15u;
        if (op == 0u || op == 1u || op == 4u || op == 7u || op == 8u) qtot += cw >> 4;
    }
    const uint64_t avail = 2ull * (r.len - r.o_seq);
    return (uint32_t)(qtot < avail ? qtot : avail);
}

// Of eight packed bases (one dword): bit 4 p set where nibble p holds a code bit2char takes.  A popcount per nibble (the
// codes of A C G T have one bit, N has four), then "count is 1 or 4" on its three bits.
__device__ __forceinline__ uint32_t good_codes(uint32_t x)
{
    const uint32_t c = x - ((x >> 1) & 0x77777777u) - ((x >> 2) & 0x33333333u) - ((x >> 3) & 0x11111111u);
    return (c >> 2) | (c & ~(c >> 1));
}

// index (0..31) of the first refused code in 16 bytes of packed bases; only called when there is one
__device__ __forceinline__ uint32_t first_refused(const uint4& x)
{
    const uint32_t w[4] = { x.x, x.y, x.z, x.w };
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t bad = ~good_codes(w[k]) & 0x11111111u;                  // base 2 b in the high nibble of byte b
        if (bad) return 8u * k + ((uint32_t)__builtin_ctz(((bad & 0x0F0F0F0Fu) << 4) | ((bad >> 4) & 0x0F0F0F0Fu)) >> 2);
    }
    return 32u;
}

__device__ __forceinline__ bool all_good(const uint4& x)
{
    return ((good_codes(x.x) & good_codes(x.y) & good_codes(x.z) & good_codes(x.w)) & 0x11111111u) == 0x11111111u;
}

// The first base bit2char would refuse in each of a wave's 64 records, swept by the wave together: four lanes per record
// (eight when a read of the wave is longer than 128 bases) read its packed bases as consecutive 16-byte pieces.  A lane
// sweeping its own record alone makes every load a 64-line gather (measured +30 % on the whole kernel); here a load
// instruction touches 16 records.  The launch is one occupancy round -- what a wave waits for in sequence is the kernel's
// duration -- so the sweep covers l_seq bases, known from the record's core: its loads go out with the loads of the
// CIGAR and the aux window (issue), and are looked at when those have arrived (finish).  What the CIGAR makes of it
// (fewer bases, or more: new_readaln reads on behind l_seq) is settled in classify.
struct BaseSweep {
    uint4 x[8];
    uint32_t lm[8];
    int shift;      // log2 of the lanes per record: 2 or 3, the same for the whole wave
};

__device__ __forceinline__ void sweep_issue(BaseSweep& S, const uint8_t* raw, uint32_t so, uint32_t lim, int lane)
{
    S.shift = __ballot(lim > 128u) != 0ull ? 3 : 2;
    const uint32_t sub = (uint32_t)lane & ((1u << S.shift) - 1u);
    const int grp = lane >> S.shift, per_pass = 64 >> S.shift;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        S.lm[j] = 0u; S.x[j] = make_uint4(0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u);
        if (j < (1 << S.shift)) {
            const int src = per_pass * j + grp;
            S.lm[j] = (uint32_t)__shfl((int)lim, src);
            const uint32_t so_j = (uint32_t)__shfl((int)so, src);
            // a lane with nothing to read takes the buffer's first bytes: the loads stay unconditional and go out together
            __builtin_memcpy(&S.x[j], raw + (32u * sub < S.lm[j] ? so_j + 16u * sub : 0u), 16);
        }
    }
}

// s_fb: the wave's 64 entries, ~0 coming in
__device__ __forceinline__ void sweep_finish(const BaseSweep& S, const uint8_t* raw, uint32_t so, uint32_t lim, uint32_t* s_fb, int lane)
{
    const uint32_t lanes = 1u << S.shift, sub = (uint32_t)lane & (lanes - 1u);
    const int grp = lane >> S.shift, per_pass = 64 >> S.shift;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (j < (1 << S.shift) && 32u * sub < S.lm[j] && !all_good(S.x[j])) {
            const uint32_t idx = 32u * sub + first_refused(S.x[j]);
            if (idx < S.lm[j]) atomicMin(&s_fb[per_pass * j + grp], idx);
        }
    }
    if (__ballot(lim > 256u) != 0ull) {                                         // reads longer than 256 bases: the rest, piece by piece
        for (int j = 0; j < 8; j++) {
            const int src = per_pass * j + grp;
            const uint32_t lm = (uint32_t)__shfl((int)lim, src), so_j = (uint32_t)__shfl((int)so, src);
            for (uint32_t first = 256u + 32u * sub; first < lm; first += 256u) {
                uint4 x; __builtin_memcpy(&x, raw + so_j + (first >> 1), 16);
                if (!all_good(x)) { const uint32_t idx = first + first_refused(x); if (idx < lm) atomicMin(&s_fb[src], idx); }
            }
        }
    }
}

struct Verdict {
    uint32_t cls;           // IM_REC_*
    bool revcomp;
    int32_t range_max;
};

// generic_ok / generic_range: the lookup of "generic" (records without an RG tag, src/indelminer.c:370), done once per workgroup
__device__ __forceinline__ Verdict classify(const RecView& r, AuxWin w, const im_triage_params& tp, const RgView& T,
                                            bool generic_ok, int32_t generic_range, uint32_t fb)
{
    Verdict v; v.cls = IM_REC_SKIP; v.revcomp = false; v.range_max = 0;
    if (!r.ok) { v.cls = IM_REC_ERR_LIMIT; return v; }
    const uint32_t flag = r.flag;
    if (flag & (0x100u | 0x200u | 0x400u | 0x800u)) return v;                  // 348-351
    const bool aligned = !(flag & 0x4u), mate_aligned = !(flag & 0x8u);
    const bool proper = (flag & 0x2u) != 0, is_rc = (flag & 0x10u) != 0, mate_rc = (flag & 0x20u) != 0;
    if (!(flag & 0x1u)) return v;                                              // 361
    if (aligned && mate_aligned && r.tid != r.mtid) return v;                  // 364-366
    v.cls = IM_REC_COUNTED;

    uint32_t o_rg, o_mq;
    find_rg_mq(r, w, o_rg, o_mq);
    if (tp.defer_ranges) {
        // the table is still being estimated: only what does not need it is checked (a read-group tag of a type bam_aux2Z refuses)
        if (o_rg) { const uint32_t type = rec_byte(r, w, o_rg); if (!(type == 'Z' || type == 'H')) { v.cls = IM_REC_ERR_RG; return v; } }
        v.range_max = 0;
    } else if (!o_rg) {
        if (!generic_ok) { v.cls = IM_REC_ERR_RG; return v; }
        v.range_max = generic_range;
    } else {
        const uint32_t type = rec_byte(r, w, o_rg);
        bool ok = type == 'Z' || type == 'H';                                  // else bam_aux2Z returns NULL: strlen(NULL)
        if (ok) {
            // the name's length; a name that runs out of the window brings the window to the tag (names of up to 22 bytes then lie in it)
            uint32_t len = 0;
            while (o_rg + 1u + len < r.len) {
                const uint32_t o = o_rg + 1u + len;
                if (o - w.o0 >= (uint32_t)kAuxWin && w.o0 != o_rg) aux_slide(r, w, o_rg);
                if (!rec_byte(r, w, o)) break;
                len++;
            }
            const uint32_t o_name = o_rg + 1u;
            ok = rg_lookup(T, [&](uint32_t k) { return rec_byte(r, w, o_name + k); }, len, v.range_max);
        }
        if (!ok) { v.cls = IM_REC_ERR_RG; return v; }
    }

    if (aligned && !mate_aligned) return v;                                    // 384-385
    if (!aligned && mate_aligned) {                                            // 386-424
        int32_t mmq = (int32_t)r.mapq;
        if (o_mq) {
            aux_cover(r, w, o_mq, 5u);
            const uint32_t t = rec_byte(r, w, o_mq);
            if (!(t == 'I' || t == 'i' || t == 'C' || t == 'c' || t == 'S' || t == 's')) { v.cls = IM_REC_ERR_MQ; return v; }
            mmq = aux_int(r, w, o_mq);
        }
        if (mmq >= tp.qthreshold) { v.cls = IM_REC_CAND_UNMAPPED; v.revcomp = !mate_rc; }
        return v;
    }
    if (aligned && mate_aligned && proper) {                                   // 425-515
        // new_readaln (src/readaln.c:186-240) builds the segment list of EVERY proper pair first, op by op: N / H / P and
        // unknown ops are fatal (163-182), and so is a base code other than A C G T N in an op that carries read bases
        // (bit2char, 4-16) -- whichever comes first along the CIGAR.  fb comes in as the first bad base among the l_seq
        // bases of the read (the wave's cooperative sweep); only what lies inside the CIGAR's reach counts, and a CIGAR
        // that reaches past l_seq has the bytes behind the packed bases looked at here.  Then the ops are walked: a bad op
        // at read offset q loses to a bad base in front of it.
        const uint32_t lim = bases_in_cigar_reach(r);
        if (fb >= lim) {
            fb = 0xFFFFFFFFu;
            for (uint32_t j = (uint32_t)r.l_seq; j < lim; j++) {
                const uint32_t code = (r.p[r.o_seq + (j >> 1)] >> ((~j & 1u) << 2)) & 15u;
                if (!(code == 1u || code == 2u || code == 4u || code == 8u || code == 15u)) { fb = j; break; }
            }
        }
        uint32_t ndel = 0, nins = 0, nclip = 0; bool three = false;
        uint64_t q = 0;
        for (uint32_t i = 0; i < r.n_cigar; i++) {
            const uint32_t cw = cigar_word(r, i), op = cw & 15u;
            if (op == 3u || op == 5u || op == 6u || op > 8u) { v.cls = (uint64_t)fb < q ? IM_REC_ERR_BASE : IM_REC_ERR_CIGAR; return v; }   // new_readseg_bam
            if (op == 0u || op == 1u || op == 4u || op == 7u || op == 8u) q += cw >> 4;
            ndel += op == 2u; nins += op == 1u; nclip += op == 4u;
            if (op == 4u && ((!is_rc && i == r.n_cigar - 1u) || (is_rc && i == 0u))) three = true;
        }
        if (fb != 0xFFFFFFFFu) { v.cls = IM_REC_ERR_BASE; return v; }
        if (ndel + nins + nclip == 0u) return v;
        if ((nclip == 0u || (nclip == 1u && three)) && ndel == 0u && nins == 0u) return v;            // 457-460
        if (o_mq) aux_cover(r, w, o_mq, 5u);
        const int32_t mmq = o_mq ? aux_int(r, w, o_mq) : (int32_t)r.mapq;
        if (mmq >= tp.qthreshold) { v.cls = IM_REC_CAND_PROPER; v.revcomp = is_rc == mate_rc; }
        return v;
    }
    if (aligned && mate_aligned && !proper) {                                  // 516-521
        const int64_t a = r.isize < 0 ? -(int64_t)r.isize : (int64_t)r.isize;
        const uint32_t a32 = (uint32_t)(r.isize < 0 ? -r.isize : r.isize);     // abs() as int, compared with a uint
        if (a > (int64_t)v.range_max && a32 < tp.maxpedelsize && is_rc != mate_rc) v.cls = IM_REC_PE;
        return v;
    }
    return v;
}

struct TriageArgs {
    im_dev_records recs;
    im_dev_cands out;
    im_triage_params tp;
    RefDev ref;
    RgTable rg;
    int32_t* depth_diff;    // genome-wide difference array (index = ascii offset of the position) or null
    uint32_t* info;         // [n] class | revcomp << 8
    int32_t* rmax;          // [n]
    uint2* blk;             // [blocks] one packed word per workgroup: padded read bytes << 32 | candidates | counted << 9 | errors << 18
    uint32_t* blocks_done;  // [1] zero between launches: groups of workgroups that are complete; the group that comes last scans grp[]
    uint32_t* grp_done;     // [kTriGroupsMax] zero between launches: workgroups of a group that have published their totals
    unsigned long long* grp;// [2 * groups] a group's totals: bytes << 32 | candidates, counted << 32 | errors
    uint2* grp_base;        // [groups] candidates / bytes in front of a group (running counters included), written by the last group
    int32_t group_size;     // workgroups per group
    uint32_t* seq_at;       // [cap_cand] byte offset of a candidate's packed bases in recs.raw, | 1 << 31 = reverse complement
    int32_t* chunk_base;    // [1] candidates before this chunk (written by the scan)
};

__device__ __forceinline__ uint32_t padded4(int32_t l) { return ((uint32_t)l + 3u) & ~3u; }

constexpr int kRgLds = 2048;        // an insert-length table up to this size is copied to LDS
constexpr int kDepthWin = 4096;     // positions of the depth difference array a workgroup gathers in LDS before it touches memory

__global__ __launch_bounds__(kTriBlock, 6) void triage_classify_kernel(TriageArgs A)
{
    __shared__ uint32_t s_cnt[kTriBlock / 64], s_bytes[kTriBlock / 64], s_counted[kTriBlock / 64], s_err[kTriBlock / 64];
    __shared__ uint32_t s_aux[kTriBlock][kAuxWin / 4];
    __shared__ uint32_t s_rg[kRgLds / 4];
    __shared__ int32_t s_generic[2];
    // The pileup's +1 / -1 of this workgroup's records, gathered here first: a coordinate-sorted BAM puts the 256 records of
    // a workgroup within a few hundred positions of each other, so their ~512 scattered device atomics become LDS adds and
    // a few whole-line atomic instructions (measured: the scatter was 45 us of a 300 000-record launch, as much as the
    // rest of the kernel).  Events outside the window or on another contig go to memory directly.
    __shared__ int32_t s_dd[kDepthWin];
    __shared__ int32_t s_wpos[kTriBlock / 64], s_wtid[kTriBlock / 64];
    __shared__ uint32_t s_fb[kTriBlock];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * kTriBlock + t;
    if (A.depth_diff) {
        int4* z = reinterpret_cast<int4*>(s_dd);
#pragma unroll
        for (int k = 0; k < kDepthWin / 4 / kTriBlock; k++) z[t + k * kTriBlock] = make_int4(0, 0, 0, 0);
    }

    // the record: offsets, core, then CIGAR head + aux window + the wave's sweep over the packed bases in one round trip
    RecView r; r.ok = false; r.l_seq = 0; r.flag = 0; r.n_cigar = 0; r.tid = -1; r.pos = 0; r.o_cigar = 0; r.o_seq = 0; r.p = A.recs.raw; r.len = 0; r.o_aux = 0;
    s_fb[t] = 0xFFFFFFFFu;
    if (i < A.recs.n) r = view_record(A.recs.raw, A.recs.rec_off[i], A.recs.rec_off[i + 1]);
    const uint32_t sw_lim = reaches_new_readaln(r) ? (uint32_t)r.l_seq : 0u;   // r.ok: the packed bases lie inside the record
    const uint32_t sw_so = sw_lim ? (uint32_t)(r.p - A.recs.raw) + r.o_seq : 0u;
    BaseSweep S;
    sweep_issue(S, A.recs.raw, sw_so, sw_lim, lane);
    if (r.ok) {
        // reads past the record stay inside the chunk buffer (>= 64 spare bytes behind the last record)
#pragma unroll
        for (int k = 0; k < 4; k++) r.cig[k] = ld_u32(r.p + r.o_cigar + 4u * k);
        // the aux window: as many dwords as the record's aux area has (a record with the MQ tag alone: one or two)
        const uint32_t aux_bytes = r.len - r.o_aux;
#pragma unroll
        for (int k = 0; k < kAuxWin / 4; k++) if (4u * k < aux_bytes) s_aux[t][k] = ld_u32(r.p + r.o_aux + 4u * k);
    }
    sweep_finish(S, A.recs.raw, sw_so, sw_lim, s_fb + 64 * wave, lane);
    // the depth window starts at the first pileup-eligible record of the workgroup (records are sorted inside a contig)
    const bool piles = A.depth_diff && r.ok && r.tid >= 0 && r.tid < A.ref.n_contigs && !(r.flag & (0x4u | 0x100u | 0x200u | 0x400u));
    if (A.depth_diff) {
        const uint64_t mp = __ballot(piles);
        const int first = mp ? (int)__builtin_ctzll(mp) : 0;
        const int fp = __shfl(r.pos, first), ft = __shfl(r.tid, first);
        if (lane == 0) { s_wpos[wave] = fp < 0 ? 0 : fp; s_wtid[wave] = mp ? ft : -1; }
    }
    // the insert-length table: LDS copy when it is small
    const bool rg_lds = A.rg.bytes <= kRgLds;
    if (rg_lds) for (int k = t; 4 * k < A.rg.bytes; k += kTriBlock) s_rg[k] = reinterpret_cast<const uint32_t*>(A.rg.blob)[k];
    __syncthreads();
    const RgView T = rg_view(rg_lds ? reinterpret_cast<const uint8_t*>(s_rg) : A.rg.blob, A.rg.n);
    if (t == 0) {
        int32_t rm = 0;
        const bool ok = rg_lookup(T, [](uint32_t k) { return (uint32_t)("generic"[k]); }, 7u, rm);
        s_generic[0] = ok ? 1 : 0; s_generic[1] = rm;
    }
    __syncthreads();
    // window base: the first wave that holds a pileup-eligible record decides (its contig, its position)
    int32_t dd_pos = 0, dd_tid = -1;
    if (A.depth_diff) {
#pragma unroll
        for (int wv = kTriBlock / 64 - 1; wv >= 0; wv--) if (s_wtid[wv] >= 0) { dd_tid = s_wtid[wv]; dd_pos = s_wpos[wv]; }
    }

    uint32_t cls = IM_REC_SKIP, bytes = 0;
    bool cand = false;
    if (i < A.recs.n) {
        AuxWin w; w.lds = s_aux[t]; w.o0 = r.o_aux;
        const Verdict v = classify(r, w, A.tp, T, s_generic[0] != 0, s_generic[1], s_fb[t]);
        cls = v.cls;
        cand = cls == IM_REC_CAND_UNMAPPED || cls == IM_REC_CAND_PROPER;
        if (cand) bytes = padded4(r.l_seq);
        A.info[i] = cls | (v.revcomp ? 0x100u : 0u);
        A.rmax[i] = v.range_max;
        if (A.out.rec_class) A.out.rec_class[i] = (uint8_t)cls;
        // what samtools' pileup counts (bam_pileup.c:171-172, 238-265): M/=/X of records that are mapped,
        // primary, not QC-failed, not duplicates
        if (piles) {
            const int64_t base = A.ref.asc_off[r.tid];
            const int64_t clen = A.ref.len[r.tid];
            const bool here = r.tid == dd_tid;
            int64_t x = r.pos;
            for (uint32_t k = 0; k < r.n_cigar; k++) {
                const uint32_t cw = cigar_word(r, k), op = cw & 15u;
                const int64_t len = cw >> 4;
                if (op == 0u || op == 7u || op == 8u) {
                    int64_t a = x < 0 ? 0 : x, b = x + len > clen ? clen : x + len;
                    if (a < b) {
                        const int64_t ra = a - dd_pos, rb = b - dd_pos;
                        if (here && ra >= 0 && ra < kDepthWin) atomicAdd(&s_dd[ra], 1); else atomicAdd(&A.depth_diff[base + a], 1);
                        if (here && rb >= 0 && rb < kDepthWin) atomicAdd(&s_dd[rb], -1); else atomicAdd(&A.depth_diff[base + b], -1);
                    }
                    x += len;
                } else if (op == 2u || op == 3u) x += len;
            }
        }
    }
    // workgroup totals
    const uint64_t mc = __ballot(cand);
    uint32_t wb = bytes;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wb += (uint32_t)__shfl_xor((int)wb, o);
    const uint64_t mk = __ballot(cls != IM_REC_SKIP), me = __ballot(cls >= IM_REC_ERR_RG);
    if (lane == 0) { s_cnt[wave] = (uint32_t)__popcll(mc); s_bytes[wave] = wb; s_counted[wave] = (uint32_t)__popcll(mk); s_err[wave] = (uint32_t)__popcll(me); }
    __syncthreads();
    if (A.depth_diff && dd_tid >= 0) {
        // the gathered window out: consecutive lanes hold consecutive positions, only non-zero entries touch memory
        int32_t* dst = A.depth_diff + A.ref.asc_off[dd_tid] + dd_pos;
        const int64_t room = (int64_t)A.ref.len[dd_tid] + 1 - dd_pos;          // the contig's run has len + 1 entries
#pragma unroll 4
        for (int k = 0; k < kDepthWin / kTriBlock; k++) {
            const int idx = t + k * kTriBlock;
            const int32_t v = s_dd[idx];
            if (v != 0 && idx < room) atomicAdd(&dst[idx], v);
        }
    }
    // Publication.  A workgroup's totals go out with a RETURNING agent-scope atomic and are counted only once the return is
    // in: the count cannot overtake the totals, and no fence is needed (an agent-scope fence is a whole-L2 write-back per
    // workgroup on this chip, profiles/README.md).  The counting is in two levels: a thousand workgroups adding to ONE word
    // took 12 of the kernel's 36 us (same-address atomics are served one after the other, ~9 ns each; measured by leaving the
    // tail out, profiles/r04_d_*), so a workgroup counts into its GROUP's word, the workgroup that completes a group adds the
    // group's totals up and counts the group, and the workgroup that completes the last group scans the groups.
    __shared__ uint32_t s_role;
    const int32_t n_blocks = (int32_t)gridDim.x;
    const int32_t gsz = A.group_size, g = (int32_t)blockIdx.x / gsz, g_first = g * gsz;
    const int32_t g_n = min(gsz, n_blocks - g_first), n_groups = (n_blocks + gsz - 1) / gsz;
    if (t == 0) {
        uint32_t c = 0, b = 0, k = 0, e = 0;
        for (int wv = 0; wv < kTriBlock / 64; wv++) { c += s_cnt[wv]; b += s_bytes[wv]; k += s_counted[wv]; e += s_err[wv]; }
        // one word per workgroup: padded read bytes | candidates, counted records and error records (each <= 256: 9 bits)
        unsigned long long seen = atomicExch(reinterpret_cast<unsigned long long*>(&A.blk[blockIdx.x]),
                                             ((unsigned long long)b << 32) | c | (k << 9) | (e << 18));
        asm volatile("" :: "v"(seen));
        s_role = atomicAdd(&A.grp_done[g], 1u) == (uint32_t)g_n - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (!s_role) return;
    // last workgroup of its group: the group's totals (the members' words come back through the atomic path they went out on)
    if (t < 64) {
        unsigned long long cb = 0, ke = 0;
        for (int32_t j = t; j < g_n; j += 64) {
            const unsigned long long pv = atomicAdd(reinterpret_cast<unsigned long long*>(&A.blk[g_first + j]), 0ull);
            cb += (pv & 0xFFFFFFFF00000000ull) | ((uint32_t)pv & 511u);
            ke += ((unsigned long long)(((uint32_t)pv >> 9) & 511u) << 32) | (((uint32_t)pv >> 18) & 511u);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            cb += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(cb >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)cb, o);
            ke += ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(ke >> 32), o) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)ke, o);
        }
        if (t == 0) {
            unsigned long long s0 = atomicExch(&A.grp[2 * g], cb), s1 = atomicExch(&A.grp[2 * g + 1], ke);
            asm volatile("" :: "v"(s0), "v"(s1));
            atomicExch(&A.grp_done[g], 0u);                                     // ready for the next launch
            s_role = atomicAdd(A.blocks_done, 1u) == (uint32_t)n_groups - 1u ? 2u : 0u;
        }
    }
    __syncthreads();
    if (s_role != 2u) return;
    // Last group to complete: exclusive scan of the groups' totals on top of the running counters (candidates are appended in
    // record order).  A few hundred totals at most; the emit kernel adds the workgroups in front of it inside its group.
    __shared__ uint32_t wc[kTriBlock / 64], wbb[kTriBlock / 64];
    __shared__ uint32_t carry_c, carry_b, s_ke[2];
    // restart: this launch opens a new batch -- the running counters count as zero whatever they hold (no memset in front)
    const bool fresh = A.tp.restart != 0;
    if (t == 0) {
        carry_c = fresh ? 0u : (uint32_t)A.out.counters[0]; carry_b = fresh ? 0u : (uint32_t)A.out.counters[1];
        A.chunk_base[0] = (int32_t)carry_c;
        s_ke[0] = 0u; s_ke[1] = 0u;
    }
    uint32_t sum_k = 0, sum_e = 0;
    __syncthreads();
    for (int32_t base = 0; base < n_groups; base += kTriBlock) {
        const int32_t j = base + t;
        uint2 v = make_uint2(0u, 0u);
        if (j < n_groups) {
            const unsigned long long cb = atomicAdd(&A.grp[2 * j], 0ull), ke = atomicAdd(&A.grp[2 * j + 1], 0ull);
            v.x = (uint32_t)cb; v.y = (uint32_t)(cb >> 32);
            sum_k += (uint32_t)(ke >> 32); sum_e += (uint32_t)ke;
        }
        uint32_t c = v.x, b = v.y;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t tc = (uint32_t)__shfl_up((int)c, o), tb = (uint32_t)__shfl_up((int)b, o);
            if (lane >= o) { c += tc; b += tb; }
        }
        if (lane == 63) { wc[wave] = c; wbb[wave] = b; }
        __syncthreads();
        uint32_t oc = 0, ob = 0;
        for (int wv = 0; wv < wave; wv++) { oc += wc[wv]; ob += wbb[wv]; }
        const uint32_t cc = carry_c, cb0 = carry_b;
        if (j < n_groups) A.grp_base[j] = make_uint2(cc + oc + c - v.x, cb0 + ob + b - v.y);
        __syncthreads();
        if (t == kTriBlock - 1) { carry_c = cc + oc + c; carry_b = cb0 + ob + b; }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_k += (uint32_t)__shfl_xor((int)sum_k, o); sum_e += (uint32_t)__shfl_xor((int)sum_e, o); }
    if (lane == 0) { atomicAdd(&s_ke[0], sum_k); atomicAdd(&s_ke[1], sum_e); }
    __syncthreads();
    if (t == 0) {
        A.out.counters[0] = (int32_t)carry_c; A.out.counters[1] = (int32_t)carry_b;
        A.out.counters[2] = (fresh ? 0 : A.out.counters[2]) + (int32_t)s_ke[0];        // records counted
        A.out.counters[3] = (fresh ? 0 : A.out.counters[3]) + (int32_t)s_ke[1];        // error records (emit / decode add theirs later)
        if (fresh) A.out.counters[4] = 0;                                              // overflows: counted by the emit kernel
        *A.blocks_done = 0u;
    }
}

// 4-bit base code -> ASCII (bit2char, src/readaln.c:4-17); 0 = a code the reference exits on
__device__ __forceinline__ uint32_t base_ascii(uint32_t code)
{
    return code == 1u ? 'A' : code == 2u ? 'C' : code == 4u ? 'G' : code == 8u ? 'T' : code == 15u ? 'N' : 0u;
}
// complement in code space (A<->T, C<->G, N<->N: src/sequences.c:22-26) = reversal of the four bits
__device__ __forceinline__ uint32_t comp_code(uint32_t c) { return ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3); }

// per candidate, by the lane that owns its record: place in the batch, scalars, CIGAR-derived evidence slots
__global__ __launch_bounds__(kTriBlock) void triage_emit_kernel(TriageArgs A)
{
    __shared__ uint32_t s_cnt[kTriBlock / 64], s_bytes[kTriBlock / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * kTriBlock + t;
    uint32_t info = 0;
    RecView r; r.ok = false; r.l_seq = 0; r.o_seq = 0; r.p = A.recs.raw;
    if (i < A.recs.n) {
        info = A.info[i];
        const uint32_t cls = info & 255u;
        if (cls == IM_REC_CAND_UNMAPPED || cls == IM_REC_CAND_PROPER) {
            r = view_record(A.recs.raw, A.recs.rec_off[i], A.recs.rec_off[i + 1]);
#pragma unroll
            for (int k = 0; k < 4; k++) r.cig[k] = ld_u32(r.p + r.o_cigar + 4u * k);
        }
    }
    const bool cand = r.ok;
    const uint32_t bytes = cand ? padded4(r.l_seq) : 0u;
    // position among the workgroup's candidates / bytes
    const uint64_t mc = __ballot(cand);
    const uint32_t below = (uint32_t)__popcll(mc & ((1ull << lane) - 1ull));
    uint32_t ib = bytes;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t tb = (uint32_t)__shfl_up((int)ib, o); if (lane >= o) ib += tb; }
    if (lane == 63) { s_cnt[wave] = (uint32_t)__popcll(mc); s_bytes[wave] = ib; }
    // candidates / bytes in front of this workgroup: its group's base (the classify kernel's scan) + the workgroups of the group in front
    __shared__ uint2 s_base;
    if (t < 64) {
        const int32_t gsz = A.group_size, g = (int32_t)blockIdx.x / gsz, g_first = g * gsz, mine = (int32_t)blockIdx.x - g_first;
        uint32_t fc = 0, fb = 0;
        for (int32_t j = t; j < mine; j += 64) { const uint2 pv = A.blk[g_first + j]; fc += pv.x & 511u; fb += pv.y; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { fc += (uint32_t)__shfl_xor((int)fc, o); fb += (uint32_t)__shfl_xor((int)fb, o); }
        if (t == 0) { const uint2 gb = A.grp_base[g]; s_base = make_uint2(gb.x + fc, gb.y + fb); }
    }
    __syncthreads();
    if (!cand) return;
    uint32_t oc = 0, ob = 0;
    for (int w = 0; w < wave; w++) { oc += s_cnt[w]; ob += s_bytes[w]; }
    const uint2 base = s_base;
    const uint32_t ci = base.x + oc + below;
    const uint32_t bo = base.y + ob + ib - bytes;
    if (!(ci < (uint32_t)A.out.cap_cand && (uint64_t)bo + bytes + 16u <= (uint64_t)A.out.cap_bases)) { atomicAdd(&A.out.counters[4], 1); return; }
    const im_dev_batch& B = A.out.batch;
    const_cast<int64_t*>(B.base_off)[ci] = (int64_t)bo;
    const_cast<int32_t*>(B.read_len)[ci] = r.l_seq;
    const_cast<int32_t*>(B.tid)[ci] = r.mtid;
    const_cast<int32_t*>(B.anchor)[ci] = r.mpos;
    const_cast<int32_t*>(B.range_max)[ci] = A.rmax[i];
    A.out.cand_rec[ci] = A.recs.rec_base + (int32_t)i;
    // chunk offsets fit 31 bits (chunks are far below 2 GiB); indexed by the candidate's place in THIS chunk
    A.seq_at[ci - (uint32_t)A.chunk_base[0]] = ((uint32_t)(r.p - A.recs.raw) + r.o_seq) | ((info & 0x100u) ? 0x80000000u : 0u);
    // check_variants (src/indelminer.c:285-337): one evidence per I / D op that is far enough from both ends
    int ne = 0;
    uint32_t err = 0;
    int32_t e_cls[IM_MAX_EV] = { -1, -1, -1, -1 }, e_b1[IM_MAX_EV] = { 0, 0, 0, 0 }, e_b2[IM_MAX_EV] = { 0, 0, 0, 0 };
    if ((info & 255u) == IM_REC_CAND_PROPER) {
        uint32_t tpos = 0, rpos = 0;
        for (uint32_t k = 0; k < r.n_cigar; k++) {
            const uint32_t w = cigar_word(r, k), op = w & 15u;
            if (op == 7u || op == 8u || op == 0u || op == 1u) tpos += w >> 4;
        }
        int32_t refpos = r.pos;
        bool over = false;      // more evidence than the slots hold: the kernel's own limit, behind anything the reference refuses further on
        for (uint32_t k = 0; k < r.n_cigar && !err; k++) {
            const uint32_t w = cigar_word(r, k), op = w & 15u, len = w >> 4;
            if (op == 2u || op == 1u) {
                if (rpos > A.tp.ethreshold_vcfcheck && (tpos - rpos) > A.tp.ethreshold_vcfcheck) {
                    const int32_t c = op == 2u ? IM_CLS_DELETION : IM_CLS_INSERTION, x1 = refpos, x2 = op == 2u ? refpos + (int32_t)len : refpos;
                    // static slot selection keeps the three small arrays in registers
                    if (ne == 0) { e_cls[0] = c; e_b1[0] = x1; e_b2[0] = x2; }
                    else if (ne == 1) { e_cls[1] = c; e_b1[1] = x1; e_b2[1] = x2; }
                    else if (ne == 2) { e_cls[2] = c; e_b1[2] = x1; e_b2[2] = x2; }
                    else if (ne == 3) { e_cls[3] = c; e_b1[3] = x1; e_b2[3] = x2; }
                    else over = true;
                    ne += ne < IM_MAX_EV ? 1 : 0;
                }
            } else if (op == 0u || op == 7u || op == 8u) rpos += len;
            else if (op == 4u) { if (!(k == 0u || k == r.n_cigar - 1u)) err = IM_REC_ERR_CLIP; }
            else err = IM_REC_ERR_CIGAR;
            if (op == 0u || op == 7u || op == 8u || op == 2u) refpos += (int32_t)len;
        }
        if (!err && over) err = IM_REC_ERR_LIMIT;
    }
    if (B.ev_cls) {
#pragma unroll
        for (int k = 0; k < IM_MAX_EV; k++) {
            const int64_t sl = (int64_t)ci * IM_MAX_EV + k;
            if (A.out.consumed) A.out.consumed[sl] = 0;        // a fresh candidate: no flush has consumed its evidence
            B.ev_cls[sl] = err ? -1 : e_cls[k];
            B.ev_b1[sl] = err ? 0 : e_b1[k];
            B.ev_b2[sl] = err ? 0 : e_b2[k];
        }
    }
    if (err) {
        A.info[i] = (info & ~255u) | err;                      // the decode below leaves a record that already failed alone
        if (A.out.rec_class) A.out.rec_class[i] = (uint8_t)err;
        atomicAdd(&A.out.counters[3], 1);
    }
}

// new_unaligned_readaln's decode (src/readaln.c:242-267) + reverse_complement_string (src/sequences.c:204-220) for the
// chunk's candidates, 32 lanes per read, four bases per lane and pass: candidates cluster around the indel sites, so
// the work is spread over the candidate list and not over the records (decoding inside the emit kernel, each wave its own
// candidates one after the other, was tried in round 4: a wave at an indel site holds thirty candidates, its neighbours none --
// 57 us against 6.2 + 5.6 at 300 000 records, 103 against 15 + 20 at 1.875 M)
__global__ __launch_bounds__(256) void triage_decode_kernel(TriageArgs A)
{
    const int sub = threadIdx.x >> 5, l32 = threadIdx.x & 31;
    const int32_t c0 = A.chunk_base[0];
    const int32_t c1 = min(A.out.counters[0], A.out.cap_cand);
    for (int32_t ci = c0 + (int32_t)blockIdx.x * 8 + sub; ci < c1; ci += (int32_t)gridDim.x * 8) {
        const uint32_t at = A.seq_at[ci - c0];
        const bool rc = (at & 0x80000000u) != 0;
        const uint8_t* seq = A.recs.raw + (at & 0x7FFFFFFFu);
        const int32_t L = A.out.batch.read_len[ci];
        uint8_t* dst = const_cast<uint8_t*>(A.out.batch.bases) + A.out.batch.base_off[ci];
        bool bad = false;
        for (int32_t p0 = 4 * l32; p0 < (int32_t)padded4(L); p0 += 128) {
            // the lane's (up to) four bases in ONE load: forwards they are two bytes (p0 is even), backwards they lie in the
            // three bytes from that of the last base on -- four bytes from there stay inside the record or the chunk's slack
            const int32_t n_here = min(4, L - p0);
            const int32_t qlo = rc ? L - p0 - n_here : p0;
            const uint32_t packed = rc ? ld_u32(seq + (qlo >> 1)) : ld_u16(seq + (p0 >> 1));
            const int32_t b0 = qlo >> 1;
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j >= n_here) break;
                const int32_t q = rc ? L - 1 - p0 - j : p0 + j;
                const uint32_t byte = (packed >> (8 * ((q >> 1) - b0))) & 255u;
                uint32_t code = (q & 1) ? (byte & 15u) : (byte >> 4);
                if (rc) code = comp_code(code);
                const uint32_t ch = base_ascii(code);
                bad |= ch == 0u;
                word |= ch << (8 * j);
            }
            *reinterpret_cast<uint32_t*>(dst + p0) = word;
        }
        // a base code the reference exits on (bit2char): the candidate's record gets the error class
        const uint64_t mb = __ballot(bad);
        const uint32_t half = (uint32_t)(mb >> (32 * ((threadIdx.x >> 5) & 1)));
        if (half != 0u && l32 == 0) {
            // check_variants comes first in the reference (src/indelminer.c:470 before 484): its error stands
            const int32_t rec = A.out.cand_rec[ci] - A.recs.rec_base;
            if ((A.info[rec] & 255u) < IM_REC_ERR_RG) {
                if (A.out.rec_class) A.out.rec_class[rec] = (uint8_t)IM_REC_ERR_BASE;
                atomicAdd(&A.out.counters[3], 1);
            }
        }
    }
}

}  // namespace

// fixed words at the head of the scratch: chunk_base (256 B), blocks_done (256 B), grp_done (zero between launches), grp, grp_base
constexpr size_t kTriFixedZero = 512 + 4 * (size_t)kTriGroupsMax;
constexpr size_t kTriFixed = kTriFixedZero + 16 * (size_t)kTriGroupsMax + 8 * (size_t)kTriGroupsMax;

size_t triage_scratch_bytes(int32_t n_records)
{
    const size_t n = (size_t)(n_records > 0 ? n_records : 1);
    const size_t blocks = (n + kTriBlock - 1) / kTriBlock;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(n * 4) + up(n * 4) + up(blocks * 8) + up(n * 4) + kTriFixed;
}

// offset of the words that must be zero before the first launch on a scratch buffer
size_t triage_scratch_zero_offset(int32_t n_records, size_t* bytes)
{
    (void)n_records;
    *bytes = kTriFixedZero;
    return 0;
}

hipError_t launch_triage(const RefDev& ref, const RgTable& rg, int32_t* depth_diff, const im_triage_params& tp,
                         const im_dev_records& recs, const im_dev_cands& out, void* scratch, hipStream_t stream)
{
    if (recs.n <= 0) return hipSuccess;
    const size_t n = (size_t)recs.n;
    const int blocks = (int)((n + kTriBlock - 1) / kTriBlock);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    TriageArgs A;
    A.recs = recs; A.out = out; A.tp = tp; A.ref = ref; A.rg = rg;
    A.depth_diff = tp.want_depth ? depth_diff : nullptr;
    // fixed words first (their place must not depend on the launch's record count), then the per-record arrays
    char* s = static_cast<char*>(scratch);
    A.chunk_base = reinterpret_cast<int32_t*>(s); s += 256;
    A.blocks_done = reinterpret_cast<uint32_t*>(s); s += 256;      // zeroed by im_dev_triage_scratch_init, left zero by every launch
    A.grp_done = reinterpret_cast<uint32_t*>(s); s += 4 * (size_t)kTriGroupsMax;       // likewise
    A.grp = reinterpret_cast<unsigned long long*>(s); s += 16 * (size_t)kTriGroupsMax;
    A.grp_base = reinterpret_cast<uint2*>(s); s += 8 * (size_t)kTriGroupsMax;
    A.group_size = kTriGroup;
    while ((blocks + A.group_size - 1) / A.group_size > kTriGroupsMax) A.group_size *= 2;
    A.info = reinterpret_cast<uint32_t*>(s); s += up(n * 4);
    A.rmax = reinterpret_cast<int32_t*>(s); s += up(n * 4);
    A.blk = reinterpret_cast<uint2*>(s); s += up((size_t)blocks * 8);
    A.seq_at = reinterpret_cast<uint32_t*>(s);
    hipLaunchKernelGGL(triage_classify_kernel, dim3(blocks), dim3(kTriBlock), 0, stream, A);
    hipLaunchKernelGGL(triage_emit_kernel, dim3(blocks), dim3(kTriBlock), 0, stream, A);
    int dgrid = (int)((n + 63) / 64);           // one 32-lane group per candidate, 8 per workgroup; ~1/8 of the records at most pays off
    if (dgrid > 16384) dgrid = 16384;           // a candidate per 32-lane group while that stays a sane grid: a group's candidates are a chain of dependent loads each
    hipLaunchKernelGGL(triage_decode_kernel, dim3(dgrid), dim3(256), 0, stream, A);
    return hipGetLastError();
}

}  // namespace im
