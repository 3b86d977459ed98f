/*
 * im_oracle_triage.c -- CPU restatement of fetch_func's per-record decisions and of the
 * READCHUNK flush bookkeeping.  TEST INFRASTRUCTURE ONLY (see im_oracle.h).
 *
 * Parity status: PINNED through the reference's own outputs.  fetch_func is a static callback
 * entangled with libbam, so it cannot be called in isolation; the restatement is pinned
 *   (a) against the candidate list the compiled reference prints with -d -l (one "Attempting"
 *       line per attempt_pe_alignment call, src/alignment.c:785-788), committed as
 *       tests/golden/triage_*.json by tests/golden/make_golden.py, and
 *   (b) end to end through every VCF golden (the product's device triage feeds them).
 *
 * All file:line citations are relative to /root/reference/.
 */
#include "im_oracle.h"

#include <stdlib.h>
#include <string.h>

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* one BAM record, decoded (SAM/BAM specification 4.2; the 32-byte core without block_size) */
typedef struct {
    int32_t tid, pos, mtid, mpos, isize, l_seq;
    int l_qname, mapq, n_cigar, flag;
    const uint8_t *cigar, *seq, *aux, *end;
} rec_t;

static int parse(const uint8_t* p, uint32_t len, rec_t* r)
{
    if (len < 32) return 0;
    r->tid = (int32_t)rd32(p); r->pos = (int32_t)rd32(p + 4);
    r->l_qname = p[8]; r->mapq = p[9];
    r->n_cigar = p[12] | (p[13] << 8); r->flag = p[14] | (p[15] << 8);
    r->l_seq = (int32_t)rd32(p + 16); r->mtid = (int32_t)rd32(p + 20); r->mpos = (int32_t)rd32(p + 24); r->isize = (int32_t)rd32(p + 28);
    if (r->l_seq < 0) return 0;
    uint64_t o = 32u + (uint64_t)r->l_qname;
    r->cigar = p + o; o += 4u * (uint64_t)r->n_cigar;
    /* the delivered-record form of include/indelminer_amd.h (im_dev_records): bin = 0xFFFF marks a record that travels
     * WITHOUT its l_seq quality bytes -- the aux area follows the packed bases directly */
    const int no_qual = p[10] == 0xFF && p[11] == 0xFF;
    r->seq = p + o; o += ((uint64_t)r->l_seq + 1) / 2 + (no_qual ? 0u : (uint64_t)r->l_seq);
    if (o > len) return 0;
    r->aux = p + o; r->end = p + len;
    return 1;
}

/* bam_aux_get (src/samtools-0.1.19/bam_aux.c:27-54): first field with the tag; NULL when absent or
 * when the walk meets a type it cannot skip */
static const uint8_t* aux_get(const rec_t* r, char a, char b)
{
    const uint8_t* s = r->aux;
    while (s + 4 <= r->end) {        /* a tail of < 4 bytes is alignment padding (im_dev_records), not a field */
        const int match = s[0] == (uint8_t)a && s[1] == (uint8_t)b;
        const uint8_t* val = s + 2;
        const int type = s[2];
        s += 3;
        if (match) return val;
        switch (type) {
        case 'A': case 'c': case 'C': s += 1; break;
        case 's': case 'S': s += 2; break;
        case 'i': case 'I': case 'f': s += 4; break;
        case 'd': s += 8; break;
        case 'Z': case 'H': while (s < r->end && *s) s++; s++; break;
        case 'B': {
            if (s + 5 > r->end) return NULL;
            int es;
            switch (s[0]) { case 'c': case 'C': case 'A': es = 1; break; case 's': case 'S': es = 2; break;
                            case 'i': case 'I': case 'f': es = 4; break; case 'd': es = 8; break; default: es = 0; }
            const uint64_t adv = 5u + (uint64_t)es * rd32(s + 1);
            if (adv > (uint64_t)(r->end - s)) return NULL;
            s += adv;
            break;
        }
        default: return NULL;
        }
    }
    return NULL;
}

/* bam_aux2i (bam_aux.c:163-174) */
static int32_t aux2i(const rec_t* r, const uint8_t* v)
{
    const long room = (long)(r->end - v) - 1;
    switch (v[0]) {
    case 'c': return room >= 1 ? (int8_t)v[1] : 0;
    case 'C': return room >= 1 ? v[1] : 0;
    case 's': return room >= 2 ? (int16_t)(v[1] | (v[2] << 8)) : 0;
    case 'S': return room >= 2 ? (v[1] | (v[2] << 8)) : 0;
    case 'i': case 'I': return room >= 4 ? (int32_t)rd32(v + 1) : 0;
    default: return 0;
    }
}

/* lookup_hashtable over the insert-length table (src/hashtable.c:62-81): the table has 16 bins
 * (src/indelminer.c:702), a chain lists its entries newest first (add_hashtable prepends, 44-45),
 * a stored name matches when strncmp(stored, query, len(query)) == 0, the LAST match of the chain
 * is returned.  names[] are in insertion order. */
static uint32_t djb2_backwards(const char* s, int len)
{
    uint32_t h = 5381;
    for (int i = len - 1; i >= 0; i--) h = h * 33u + (uint32_t)(int)s[i];      /* src/hashfunc.c:23-30 */
    return h;
}

static int rg_range(int32_t n_rg, const char* const* names, const int32_t* range_max, const char* q, int qlen, int32_t* out)
{
    const uint32_t bin = djb2_backwards(q, qlen) & 15u;
    /* the last match of a newest-first chain is the OLDEST matching entry */
    for (int32_t i = 0; i < n_rg; i++) {
        const int nl = (int)strlen(names[i]);
        if ((djb2_backwards(names[i], nl) & 15u) != bin) continue;
        if (strncmp(names[i], q, (size_t)qlen) == 0) { *out = range_max[i]; return 1; }
    }
    return 0;
}

/* fetch_func for one record (src/indelminer.c:339-521).  cls as IM_REC_* of include/indelminer_amd.h.
 * bases_out (l_seq bytes) receives the read as attempt_pe_alignment would see it for candidates. */
void imo_triage_record(const uint8_t* rec, uint32_t len,
                       int32_t n_rg, const char* const* rg_names, const int32_t* rg_range_max,
                       int32_t qthreshold, uint32_t ethreshold_vcfcheck, uint32_t maxpedelsize,
                       imo_triage* out, char* bases_out)
{
    memset(out, 0, sizeof *out);
    rec_t r;
    if (!parse(rec, len, &r)) { out->cls = 21; return; }
    const int flag = r.flag;
    if (flag & (0x100 | 0x200 | 0x400 | 0x800)) return;                        /* 348-351 */
    const int aligned = !(flag & 0x4), mate_aligned = !(flag & 0x8), se = !(flag & 0x1);
    const int proper = (flag & 0x2) != 0, rc = (flag & 0x10) != 0, mate_rc = (flag & 0x20) != 0;
    if (se) return;                                                            /* 361 */
    if (aligned && mate_aligned && r.tid != r.mtid) return;                    /* 364-366 */
    out->cls = 1;

    const uint8_t* rg = aux_get(&r, 'R', 'G');
    const char* rgname = "generic";                                            /* 369-373 */
    int rglen = 7;
    if (rg) {
        if (rg[0] != 'Z' && rg[0] != 'H') { out->cls = 16; return; }           /* bam_aux2Z -> NULL */
        rgname = (const char*)rg + 1;
        rglen = 0;
        while ((const uint8_t*)rgname + rglen < r.end && rgname[rglen]) rglen++;
    }
    if (n_rg < 0) out->range_max = 0;          /* deferred ranges (im_triage_params.defer_ranges): no look-up, the caller fills the range in */
    else if (!rg_range(n_rg, rg_names, rg_range_max, rgname, rglen, &out->range_max)) { out->cls = 16; return; }
    const uint8_t* pmmq = aux_get(&r, 'M', 'Q');

    char strand = rc ? '-' : '+';
    int want = 0, revcomp = 0;
    if (aligned && !mate_aligned) return;                                      /* 384-385 */
    else if (!aligned && mate_aligned) {                                       /* 386-424 */
        int mmq = r.mapq;
        if (pmmq) {
            if (!strchr("IiCcSs", pmmq[0]) || pmmq[0] == 0) { out->cls = 17; return; }
            mmq = aux2i(&r, pmmq);
        }
        if (mmq < qthreshold) return;
        want = 2; revcomp = !mate_rc; out->qual = mmq; out->want = 2;
    } else if (aligned && mate_aligned && proper) {                            /* 425-515 */
        int ndel = 0, nins = 0, nclip = 0, three = 0;
        /* new_readaln (src/readaln.c:186-240) builds the segment list of EVERY proper pair before anything is decided: op by op,
         * N / H / P and unknown ops are fatal (163-182) and so is a base code other than A C G T N in an op that carries read bases
         * (bit2char, 4-16, through 116-171) -- whichever comes first along the CIGAR.  Read bases are taken where the CIGAR says,
         * also past l_seq (the bytes behind the packed bases); here not past the record. */
        {
            int64_t q = 0;
            const int64_t avail = 2 * (int64_t)(r.end - r.seq);
            for (int i = 0; i < r.n_cigar; i++) {
                const uint32_t w = rd32(r.cigar + 4 * i);
                const int op = (int)(w & 15u);
                const int64_t l = w >> 4;
                if (op == 3 || op == 5 || op == 6 || op > 8) { out->cls = 18; return; }
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) {
                    for (int64_t j = q; j < q + l && j < avail; j++) {
                        const int c = (r.seq[j >> 1] >> ((~j & 1) << 2)) & 15;
                        if (!(c == 1 || c == 2 || c == 4 || c == 8 || c == 15)) { out->cls = 20; return; }
                    }
                    q += l;
                }
            }
        }
        for (int i = 0; i < r.n_cigar; i++) {
            const int op = (int)(rd32(r.cigar + 4 * i) & 15u);
            if (op == 2) ndel++;
            if (op == 1) nins++;
            if (op == 4) { nclip++; if ((strand == '+' && i == r.n_cigar - 1) || (strand == '-' && i == 0)) three = 1; }
        }
        if (ndel + nins + nclip == 0) return;
        if ((nclip == 0 || (nclip == 1 && three)) && ndel == 0 && nins == 0) return;      /* 457-460 */
        const int mmq = pmmq ? aux2i(&r, pmmq) : r.mapq;
        if (mmq < qthreshold) return;
        want = 3; revcomp = rc == mate_rc; out->qual = r.mapq; out->want = 3;
        /* check_variants (285-337) */
        uint32_t tpos = 0, rpos = 0;
        for (int i = 0; i < r.n_cigar; i++) {
            const uint32_t w = rd32(r.cigar + 4 * i);
            const int op = (int)(w & 15u);
            if (op == 7 || op == 8 || op == 0 || op == 1) tpos += w >> 4;
        }
        int32_t refpos = r.pos;
        for (int i = 0; i < r.n_cigar; i++) {
            const uint32_t w = rd32(r.cigar + 4 * i);
            const int op = (int)(w & 15u), l = (int)(w >> 4);
            if (op == 2 || op == 1) {
                if (rpos > ethreshold_vcfcheck && (tpos - rpos) > ethreshold_vcfcheck) {
                    if (out->n_ev >= IMO_MAX_EV) { out->cls = 21; return; }
                    out->ev_cls[out->n_ev] = op == 2 ? 1 : 0;
                    out->ev_b1[out->n_ev] = refpos;
                    out->ev_b2[out->n_ev] = op == 2 ? refpos + l : refpos;
                    out->n_ev++;
                }
            } else if (op == 0 || op == 7 || op == 8) rpos += (uint32_t)l;
            else if (op == 4) { if (!(i == 0 || i == r.n_cigar - 1)) { out->cls = 19; return; } }
            else { out->cls = 18; return; }
            if (op == 0 || op == 7 || op == 8 || op == 2) refpos += l;
        }
    } else if (aligned && mate_aligned && !proper) {                           /* 516-521 */
        const int a = abs(r.isize);
        if (a > out->range_max && (uint32_t)a < maxpedelsize && rc != mate_rc) out->cls = 4;
        return;
    } else return;

    out->cls = want;
    out->revcomp = revcomp;
    if (revcomp) strand = strand == '+' ? '-' : '+';
    out->strand = strand;
    out->tid = r.mtid; out->anchor = r.mpos; out->l_seq = r.l_seq;
    /* new_unaligned_readaln (src/readaln.c:242-267) + reverse_complement_string (src/sequences.c:204-220) */
    static const char dec[16] = { 0, 'A', 'C', 0, 'G', 0, 0, 0, 'T', 0, 0, 0, 0, 0, 0, 'N' };
    for (int32_t i = 0; i < r.l_seq; i++) {
        const int c = (r.seq[i >> 1] >> ((~i & 1) << 2)) & 15;
        const char ch = dec[c];
        if (!ch) { out->cls = 20; return; }
        if (bases_out) {
            if (!revcomp) bases_out[i] = ch;
            else bases_out[r.l_seq - 1 - i] = ch == 'A' ? 'T' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch == 'T' ? 'A' : 'N';
        }
    }
}

/* pileup eligibility and match segments of one record (src/shared.c:160-176, bam_pileup.c:171-172,238-265):
 * adds the record's M/=/X positions to depth[0..clen) when the record belongs to contig tid */
void imo_depth_add(const uint8_t* rec, uint32_t len, int32_t tid, int32_t* depth, int64_t clen)
{
    rec_t r;
    if (!parse(rec, len, &r)) return;
    if (r.tid != tid || r.tid < 0 || (r.flag & (0x4 | 0x100 | 0x200 | 0x400))) return;
    int64_t x = r.pos;
    for (int i = 0; i < r.n_cigar; i++) {
        const uint32_t w = rd32(r.cigar + 4 * i);
        const int op = (int)(w & 15u);
        const int64_t l = w >> 4;
        if (op == 0 || op == 7 || op == 8) {
            for (int64_t p = x < 0 ? 0 : x; p < x + l && p < clen; p++) depth[p]++;
            x += l;
        } else if (op == 2 || op == 3) x += l;
    }
}

/* process_evidence's node selection for one flush (src/indelminer.c:123-146) over pending entries:
 * sort by (b1,b2), walk until the first entry with b2 >= marker.  consumed[i] != 0 or cls[i] < 0
 * = not pending.  Marks consumed[i] = flush_id for the entries that become nodes; returns their count. */
typedef struct { int32_t b1, b2, idx; } fkey;
static int cmp_fkey(const void* a, const void* b)
{
    const fkey* x = a; const fkey* y = b;
    if (x->b1 != y->b1) return x->b1 < y->b1 ? -1 : 1;
    if (x->b2 != y->b2) return x->b2 < y->b2 ? -1 : 1;
    return 0;
}
int32_t imo_flush_cut(int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                      int32_t marker, int32_t flush_id)
{
    fkey* k = malloc(sizeof(fkey) * (size_t)(n > 0 ? n : 1));
    int32_t m = 0;
    for (int32_t i = 0; i < n; i++)
        if (cls[i] >= 0 && consumed[i] == 0) { k[m].b1 = b1[i]; k[m].b2 = b2[i]; k[m].idx = i; m++; }
    qsort(k, (size_t)m, sizeof(fkey), cmp_fkey);
    int32_t taken = 0;
    for (int32_t s = 0; s < m; s++) {
        if (k[s].b2 >= marker) break;
        consumed[k[s].idx] = flush_id;
        taken++;
    }
    free(k);
    return taken;
}

/* The flush list WITHOUT history -- the formulation im_dev_flush_groupby computes (include/indelminer_amd.h), restated
 * sequentially so that it can be held against the step-by-step imo_flush_cut above.
 *
 * find_marker (src/indelminer.c:211-233) is a minimum over pair-table entries that only leave the table or enter it at
 * the current position of a coordinate-sorted walk, and a flush's marker is min(that, current position) (622-623):
 * within a contig the markers never decrease.  What process_evidence (123-146) consumed at flush f' sorted in front of
 * the first entry with b2 >= marker(f'), so it had b2 < marker(f') <= marker(f) for every later f and could not be f's
 * cutting entry.  Hence   cut(f) = min{(b1,b2)(e) : arr(e) <= f, b2(e) >= marker(f)}   over ALL entries of the contig,
 * and   consumed(e) = id of the first f >= arr(e) with (b1,b2)(e) < cut(f).
 *
 * Flush f: marker[f], id[f] (> 0), last[f] = index of the last flush of f's contig.  Entry i: arr[i] = the first flush
 * whose bounds cover it (n_fl: none; it then stays pending).  Returns 0, or -1 if the markers of some contig decrease
 * (the formulation does not apply; consumed[] is untouched). */
int32_t imo_flush_nohistory(int32_t n_fl, const int32_t* marker, const int32_t* id, const int32_t* last,
                            int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2, const int32_t* arr, int32_t* consumed)
{
    for (int32_t f = 0; f + 1 < n_fl; f++)
        if (last[f] > f && marker[f + 1] < marker[f]) return -1;
    uint64_t* cut = malloc(sizeof(uint64_t) * (size_t)(n_fl > 0 ? n_fl : 1));
    for (int32_t f = 0; f < n_fl; f++) cut[f] = ~(uint64_t)0;
    for (int32_t i = 0; i < n; i++) {
        if (cls[i] < 0 || arr[i] >= n_fl) continue;
        const uint64_t key = ((uint64_t)(uint32_t)b1[i] << 32) | (uint32_t)b2[i];
        for (int32_t f = arr[i]; f <= last[arr[i]] && marker[f] <= b2[i]; f++)
            if (key < cut[f]) cut[f] = key;
    }
    for (int32_t i = 0; i < n; i++) {
        consumed[i] = 0;
        if (cls[i] < 0 || arr[i] >= n_fl) continue;
        const uint64_t key = ((uint64_t)(uint32_t)b1[i] << 32) | (uint32_t)b2[i];
        for (int32_t f = arr[i]; f <= last[arr[i]]; f++)
            if (key < cut[f]) { consumed[i] = id[f]; break; }
    }
    free(cut);
    return 0;
}
