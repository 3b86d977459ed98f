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

__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint32_t ld_u16(const uint8_t* p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }

struct RecView {
    const uint8_t* p;       // record start (the 32-byte core)
    uint32_t len;           // bytes of the record
    int32_t tid, pos, mtid, mpos, isize, l_seq;
    uint32_t l_qname, mapq, n_cigar, flag;
    uint32_t o_cigar, o_seq, o_aux;
    bool ok;
};

__device__ __forceinline__ RecView view_record(const uint8_t* raw, uint32_t off, uint32_t end)
{
    RecView r;
    r.p = raw + off; r.len = end - off; r.ok = false;
    r.tid = r.pos = r.mtid = r.mpos = r.isize = r.l_seq = 0;
    r.l_qname = r.mapq = r.n_cigar = r.flag = 0; r.o_cigar = r.o_seq = r.o_aux = 0;
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
    const uint64_t o_aux = (uint64_t)r.o_seq + (((uint64_t)r.l_seq + 1u) >> 1) + (uint64_t)r.l_seq;
    if (o_aux > r.len) return r;
    r.o_aux = (uint32_t)o_aux;
    r.ok = true;
    return r;
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

// bam_aux_get for RG and MQ in one walk (bam_aux.c:27-54): offsets of the TYPE byte of the first
// occurrence, 0 = absent.  The walk stops where samtools' would (unknown type, truncated B array).
__device__ __forceinline__ void find_rg_mq(const RecView& r, uint32_t& o_rg, uint32_t& o_mq)
{
    o_rg = 0; o_mq = 0;
    uint32_t s = r.o_aux;
    const uint32_t end = r.len;
    while (s + 3u <= end) {
        const uint32_t t0 = r.p[s], t1 = r.p[s + 1], type = r.p[s + 2];
        if (t0 == 'R' && t1 == 'G' && !o_rg) o_rg = s + 2u;
        if (t0 == 'M' && t1 == 'Q' && !o_mq) o_mq = s + 2u;
        if (o_rg && o_mq) return;
        s += 3u;
        if (type == 'Z' || type == 'H') { while (s < end && r.p[s]) s++; s++; }
        else if (type == 'B') {
            if (s + 5u > end) return;
            const int sz = aux_size(r.p[s]);
            const uint32_t cnt = ld_u32(r.p + s + 1);
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
__device__ __forceinline__ int32_t aux_int(const RecView& r, uint32_t o)
{
    const uint32_t type = r.p[o];
    const uint8_t* s = r.p + o + 1;
    if (o + 1u >= r.len) return 0;
    switch (type) {
    case 'c': return (int32_t)(int8_t)s[0];
    case 'C': return (int32_t)s[0];
    case 's': return (o + 3u <= r.len) ? (int32_t)(int16_t)ld_u16(s) : 0;
    case 'S': return (o + 3u <= r.len) ? (int32_t)ld_u16(s) : 0;
    case 'i': case 'I': return (o + 5u <= r.len) ? (int32_t)ld_u32(s) : 0;
    default: return 0;
    }
}

// must_find_hashtable(insertlengths, rgname, strlen(rgname)) (src/indelminer.c:374-376): DJB2 over the
// bytes back to front (src/hashfunc.c:23-30), 16 bins, the chain walked head to tail, strncmp prefix
// match, LAST hit wins (src/hashtable.c:62-81).  Returns false when the reference would exit.
__device__ __forceinline__ bool rg_lookup(const RgTable& T, const uint8_t* name, uint32_t len, int32_t& range_max)
{
    uint32_t h = 5381u;
    for (int i = (int)len - 1; i >= 0; i--) h += (h << 5) + (uint32_t)(int32_t)(int8_t)name[i];
    const uint32_t bin = h & 15u;
    bool hit = false;
    for (int32_t e = T.bin_start[bin]; e < T.bin_start[bin + 1]; e++) {
        const uint32_t el = (uint32_t)T.name_len[e];
        if (el < len) continue;                      // the stored name ends first: strncmp sees NUL != byte
        const uint8_t* en = T.names + T.name_off[e];
        bool same = true;
        for (uint32_t i = 0; i < len && same; i++) same = en[i] == name[i];
        if (same) { hit = true; range_max = T.range_max[e]; }
    }
    return hit;
}

__device__ const uint8_t kGeneric[8] = { 'g', 'e', 'n', 'e', 'r', 'i', 'c', 0 };      // src/indelminer.c:370

struct Verdict {
    uint32_t cls;           // IM_REC_*
    bool revcomp;
    int32_t range_max;
};

__device__ __forceinline__ Verdict classify(const RecView& r, const im_triage_params& tp, const RgTable& T)
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
    find_rg_mq(r, o_rg, o_mq);
    {
        const uint8_t* name = kGeneric; uint32_t len = 7;
        bool bad = false;
        if (o_rg) {
            const uint32_t type = r.p[o_rg];
            if (type != 'Z' && type != 'H') bad = true;                        // bam_aux2Z returns NULL: strlen(NULL)
            else {
                name = r.p + o_rg + 1; len = 0;
                while (o_rg + 1u + len < r.len && name[len]) len++;
            }
        }
        if (bad || !rg_lookup(T, name, len, v.range_max)) { v.cls = IM_REC_ERR_RG; return v; }
    }

    if (aligned && !mate_aligned) return v;                                    // 384-385
    if (!aligned && mate_aligned) {                                            // 386-424
        int32_t mmq = (int32_t)r.mapq;
        if (o_mq) {
            const uint32_t t = r.p[o_mq];
            if (!(t == 'I' || t == 'i' || t == 'C' || t == 'c' || t == 'S' || t == 's')) { v.cls = IM_REC_ERR_MQ; return v; }
            mmq = aux_int(r, o_mq);
        }
        if (mmq >= tp.qthreshold) { v.cls = IM_REC_CAND_UNMAPPED; v.revcomp = !mate_rc; }
        return v;
    }
    if (aligned && mate_aligned && proper) {                                   // 425-515
        uint32_t ndel = 0, nins = 0, nclip = 0; bool three = false;
        for (uint32_t i = 0; i < r.n_cigar; i++) {
            const uint32_t op = ld_u32(r.p + r.o_cigar + 4u * i) & 15u;
            if (op == 3u || op == 5u || op == 6u || op > 8u) { v.cls = IM_REC_ERR_CIGAR; return v; }   // new_readseg_bam
            ndel += op == 2u; nins += op == 1u; nclip += op == 4u;
            if (op == 4u && ((!is_rc && i == r.n_cigar - 1u) || (is_rc && i == 0u))) three = true;
        }
        if (ndel + nins + nclip == 0u) return v;
        if ((nclip == 0u || (nclip == 1u && three)) && ndel == 0u && nins == 0u) return v;            // 457-460
        const int32_t mmq = o_mq ? aux_int(r, o_mq) : (int32_t)r.mapq;
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
    uint2* blk;             // [blocks] candidates, padded read bytes -> exclusive bases after the scan
};

__device__ __forceinline__ uint32_t padded4(int32_t l) { return ((uint32_t)l + 3u) & ~3u; }

__global__ __launch_bounds__(kTriBlock) void triage_classify_kernel(TriageArgs A)
{
    __shared__ uint32_t s_cnt[kTriBlock / 64], s_bytes[kTriBlock / 64], s_counted[kTriBlock / 64], s_err[kTriBlock / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t i = (int64_t)blockIdx.x * kTriBlock + t;
    uint32_t cls = IM_REC_SKIP, bytes = 0;
    bool cand = false;
    if (i < A.recs.n) {
        const RecView r = view_record(A.recs.raw, A.recs.rec_off[i], A.recs.rec_off[i + 1]);
        const Verdict v = classify(r, A.tp, A.rg);
        cls = v.cls;
        cand = cls == IM_REC_CAND_UNMAPPED || cls == IM_REC_CAND_PROPER;
        if (cand) bytes = padded4(r.l_seq);
        A.info[i] = cls | (v.revcomp ? 0x100u : 0u);
        A.rmax[i] = v.range_max;
        if (A.out.rec_class) A.out.rec_class[i] = (uint8_t)cls;
        // what samtools' pileup counts (bam_pileup.c:171-172, 238-265): M/=/X of records that are mapped,
        // primary, not QC-failed, not duplicates
        if (A.depth_diff && r.ok && r.tid >= 0 && r.tid < A.ref.n_contigs && !(r.flag & (0x4u | 0x100u | 0x200u | 0x400u))) {
            const int64_t base = A.ref.asc_off[r.tid];
            const int64_t clen = A.ref.len[r.tid];
            int64_t x = r.pos;
            for (uint32_t k = 0; k < r.n_cigar; k++) {
                const uint32_t w = ld_u32(r.p + r.o_cigar + 4u * k), op = w & 15u;
                const int64_t len = w >> 4;
                if (op == 0u || op == 7u || op == 8u) {
                    int64_t a = x < 0 ? 0 : x, b = x + len > clen ? clen : x + len;
                    if (a < b) { atomicAdd(&A.depth_diff[base + a], 1); atomicAdd(&A.depth_diff[base + b], -1); }
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
    if (t == 0) {
        uint32_t c = 0, b = 0, k = 0, e = 0;
        for (int w = 0; w < kTriBlock / 64; w++) { c += s_cnt[w]; b += s_bytes[w]; k += s_counted[w]; e += s_err[w]; }
        A.blk[blockIdx.x] = make_uint2(c, b);
        if (k) atomicAdd(&A.out.counters[2], (int32_t)k);
        if (e) atomicAdd(&A.out.counters[3], (int32_t)e);
    }
}

// exclusive scan of the workgroup totals by one workgroup, on top of the running counters
__global__ __launch_bounds__(1024) void triage_scan_kernel(uint2* __restrict__ blk, int32_t n_blocks, int32_t* __restrict__ counters)
{
    __shared__ uint32_t wc[16], wb[16];
    __shared__ uint32_t carry_c, carry_b;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) { carry_c = (uint32_t)counters[0]; carry_b = (uint32_t)counters[1]; }
    __syncthreads();
    for (int32_t base = 0; base < n_blocks; base += 1024) {
        const int32_t i = base + t;
        const uint2 v = (i < n_blocks) ? blk[i] : make_uint2(0u, 0u);
        uint32_t c = v.x, b = v.y;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t tc = (uint32_t)__shfl_up((int)c, o), tb = (uint32_t)__shfl_up((int)b, o);
            if (lane >= o) { c += tc; b += tb; }
        }
        if (lane == 63) { wc[wave] = c; wb[wave] = b; }
        __syncthreads();
        uint32_t oc = 0, ob = 0;
        for (int w = 0; w < wave; w++) { oc += wc[w]; ob += wb[w]; }
        const uint32_t cc = carry_c, cb = carry_b;
        if (i < n_blocks) blk[i] = make_uint2(cc + oc + c - v.x, cb + ob + b - v.y);
        __syncthreads();
        if (t == 1023) { carry_c = cc + oc + c; carry_b = cb + ob + b; }
        __syncthreads();
    }
    if (t == 0) { counters[0] = (int32_t)carry_c; counters[1] = (int32_t)carry_b; }
}

// 4-bit base code -> ASCII (bit2char, src/readaln.c:4-17); 0 = a code the reference exits on
__device__ __forceinline__ uint32_t base_ascii(uint32_t code)
{
    // index:        0  1   2   3  4   5  6  7  8   9 10 11 12 13 14 15
    //               -  A   C   -  G   -  -  -  T   -  -  -  -  -  -  N
    return code == 1u ? 'A' : code == 2u ? 'C' : code == 4u ? 'G' : code == 8u ? 'T' : code == 15u ? 'N' : 0u;
}
// complement in code space (A<->T, C<->G, N<->N: src/sequences.c:22-26) = reversal of the four bits
__device__ __forceinline__ uint32_t comp_code(uint32_t c) { return ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3); }

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
        if (cls == IM_REC_CAND_UNMAPPED || cls == IM_REC_CAND_PROPER) r = view_record(A.recs.raw, A.recs.rec_off[i], A.recs.rec_off[i + 1]);
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
    __syncthreads();
    uint32_t oc = 0, ob = 0;
    for (int w = 0; w < wave; w++) { oc += s_cnt[w]; ob += s_bytes[w]; }
    const uint2 base = A.blk[blockIdx.x];
    const uint32_t ci = base.x + oc + below;
    const uint32_t bo = base.y + ob + ib - bytes;
    const bool fits = cand && ci < (uint32_t)A.out.cap_cand && (uint64_t)bo + bytes + 16u <= (uint64_t)A.out.cap_bases;
    if (cand && !fits) atomicAdd(&A.out.counters[4], 1);
    uint32_t err = 0;
    if (fits) {
        const im_dev_batch& B = A.out.batch;
        const_cast<int64_t*>(B.base_off)[ci] = (int64_t)bo;
        const_cast<int32_t*>(B.read_len)[ci] = r.l_seq;
        const_cast<int32_t*>(B.tid)[ci] = r.mtid;
        const_cast<int32_t*>(B.anchor)[ci] = r.mpos;
        const_cast<int32_t*>(B.range_max)[ci] = A.rmax[i];
        A.out.cand_rec[ci] = A.recs.rec_base + (int32_t)i;
        // check_variants (src/indelminer.c:285-337): one evidence per I / D op that is far enough from both ends
        int ne = 0;
        int32_t ecls[IM_MAX_EV], eb1[IM_MAX_EV], eb2[IM_MAX_EV];
        if ((info & 255u) == IM_REC_CAND_PROPER) {
            uint32_t tpos = 0, rpos = 0;
            for (uint32_t k = 0; k < r.n_cigar; k++) {
                const uint32_t w = ld_u32(r.p + r.o_cigar + 4u * k), op = w & 15u;
                if (op == 7u || op == 8u || op == 0u || op == 1u) tpos += w >> 4;
            }
            int32_t refpos = r.pos;
            for (uint32_t k = 0; k < r.n_cigar && !err; k++) {
                const uint32_t w = ld_u32(r.p + r.o_cigar + 4u * k), op = w & 15u, len = w >> 4;
                if (op == 2u || op == 1u) {
                    if (rpos > A.tp.ethreshold_vcfcheck && (tpos - rpos) > A.tp.ethreshold_vcfcheck) {
                        if (ne >= IM_MAX_EV) { err = IM_REC_ERR_LIMIT; break; }
                        ecls[ne] = op == 2u ? IM_CLS_DELETION : IM_CLS_INSERTION;
                        eb1[ne] = refpos; eb2[ne] = op == 2u ? refpos + (int32_t)len : refpos;
                        ne++;
                    }
                } else if (op == 0u || op == 7u || op == 8u) rpos += len;
                else if (op == 4u) { if (!(k == 0u || k == r.n_cigar - 1u)) err = IM_REC_ERR_CLIP; }
                else err = IM_REC_ERR_CIGAR;
                if (op == 0u || op == 7u || op == 8u || op == 2u) refpos += (int32_t)len;
            }
        }
        if (B.ev_cls) {
#pragma unroll
            for (int k = 0; k < IM_MAX_EV; k++) {
                const int64_t sl = (int64_t)ci * IM_MAX_EV + k;
                const bool live = k < ne && !err;
                B.ev_cls[sl] = live ? ecls[k] : -1;
                B.ev_b1[sl] = live ? eb1[k] : 0;
                B.ev_b2[sl] = live ? eb2[k] : 0;
            }
        }
    }
    // the wave decodes the bases of its candidates one read at a time, four bases per lane and pass
    uint64_t todo = __ballot(fits);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const uint32_t s_off = (uint32_t)__shfl((int)(uint32_t)(r.p - A.recs.raw), src);     // chunk offsets fit 32 bits
        const uint32_t s_seq = (uint32_t)__shfl((int)r.o_seq, src);
        const int32_t L = __shfl(r.l_seq, src);
        const uint32_t s_bo = (uint32_t)__shfl((int)bo, src);
        const bool rc = (__shfl((int)info, src) & 0x100) != 0;
        const uint8_t* seq = A.recs.raw + s_off + s_seq;
        uint8_t* dst = const_cast<uint8_t*>(A.out.batch.bases) + s_bo;
        bool bad = false;
        for (int32_t p0 = 4 * lane; p0 < (int32_t)padded4(L); p0 += 256) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int32_t p = p0 + j;
                if (p >= L) break;
                const int32_t q = rc ? L - 1 - p : p;
                const uint32_t byte = seq[q >> 1];
                uint32_t code = (q & 1) ? (byte & 15u) : (byte >> 4);
                if (rc) code = comp_code(code);
                const uint32_t ch = base_ascii(code);
                bad |= ch == 0u;
                word |= ch << (8 * j);
            }
            *reinterpret_cast<uint32_t*>(dst + p0) = word;
        }
        if (__ballot(bad) && lane == src) err = IM_REC_ERR_BASE;
    }
    if (err) {
        if (A.out.rec_class) A.out.rec_class[i] = (uint8_t)err;
        atomicAdd(&A.out.counters[3], 1);
    }
}

}  // namespace

size_t triage_scratch_bytes(int32_t n_records)
{
    const size_t n = (size_t)(n_records > 0 ? n_records : 1);
    const size_t blocks = (n + kTriBlock - 1) / kTriBlock;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(n * 4) + up(n * 4) + up(blocks * 8);
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
    char* s = static_cast<char*>(scratch);
    A.info = reinterpret_cast<uint32_t*>(s); s += up(n * 4);
    A.rmax = reinterpret_cast<int32_t*>(s); s += up(n * 4);
    A.blk = reinterpret_cast<uint2*>(s);
    hipLaunchKernelGGL(triage_classify_kernel, dim3(blocks), dim3(kTriBlock), 0, stream, A);
    hipLaunchKernelGGL(triage_scan_kernel, dim3(1), dim3(1024), 0, stream, A.blk, blocks, out.counters);
    hipLaunchKernelGGL(triage_emit_kernel, dim3(blocks), dim3(kTriBlock), 0, stream, A);
    return hipGetLastError();
}

}  // namespace im
