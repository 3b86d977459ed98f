/* host_logic.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: the reference's host-side logic restated: hash table, segment lists, evidence, fetch_func's record rules, variants
 * (merge / filter / print), region depth, -o detailed, process_evidence, annotate mode.  Nothing here touches a file or the device
 * pipeline except through the driver struct. */

/* ------------------------------------------------------------------ qhash -- */

static uint32_t djb2_rev(const char* data, int len)
{
    uint32_t result = 5381;
    for (int i = len - 1; i >= 0; i--) result += (result << 5) + (uint32_t)(int)data[i];
    return result;
}

qhash* qhash_new(int po2size)
{
    qhash* h = xcalloc(1, sizeof *h);
    h->po2 = po2size > 24 ? 24 : po2size;
    h->mask = (1u << h->po2) - 1u;
    h->bins = xcalloc((size_t)1 << h->po2, sizeof(qbin*));
    return h;
}

void qhash_add(qhash* h, const char* name, int len, void* val)
{
    const uint32_t idx = djb2_rev(name, len) & h->mask;
    qbin* b = xcalloc(1, sizeof *b);
    b->name = xmalloc((size_t)len + 1);
    memcpy(b->name, name, (size_t)len);
    b->name[len] = 0;
    b->val = val;
    b->next = h->bins[idx];
    h->bins[idx] = b;
}

qbin* qhash_lookup(qhash* h, const char* name, int len)
{
    const uint32_t idx = djb2_rev(name, len) & h->mask;
    qbin* hit = NULL;
    for (qbin* it = h->bins[idx]; it; it = it->next)
        if (strncmp(it->name, name, (size_t)len) == 0) hit = it;      /* LAST match (src/hashtable.c:73-79) */
    return hit;
}

void* qhash_remove(qhash* h, const char* name, int len)
{
    const uint32_t idx = djb2_rev(name, len) & h->mask;
    qbin** pp = &h->bins[idx];
    for (; *pp; pp = &(*pp)->next) {
        if (strncmp((*pp)->name, name, (size_t)len) == 0) {           /* FIRST match (src/hashtable.c:133-141) */
            qbin* b = *pp;
            void* v = b->val;
            *pp = b->next;
            free(b->name); free(b);
            return v;
        }
    }
    return NULL;
}

void qhash_free(qhash* h, void (*free_val)(void*))
{
    if (!h) return;
    for (uint32_t i = 0; i <= h->mask; i++) {
        qbin* it = h->bins[i];
        while (it) { qbin* n = it->next; if (free_val) free_val(it->val); free(it->name); free(it); it = n; }
    }
    free(h->bins); free(h);
}

/* --------------------------------------------------------------- seglists -- */

static const char kRevcomp[256] = {
    ['A'] = 'T', ['C'] = 'G', ['G'] = 'C', ['T'] = 'A', ['N'] = 'N',
    ['a'] = 't', ['c'] = 'g', ['g'] = 'c', ['t'] = 'a', ['n'] = 'n',
};

static char bit2char(int enc)
{
    /* src/readaln.c:4-17 */
    switch (enc & 0xF) {
    case 1: return 'A';
    case 2: return 'C';
    case 4: return 'G';
    case 8: return 'T';
    case 15: return 'N';
    default: fatalf("Unhandled base encoding : %d:%d", enc, enc & 0xF);
    }
    return 'X';
}

/* What new_readaln refuses (src/readaln.c:186-240), in its order: op by op, N / H / P and unknown ops are fatal (163-182) and so
 * is a base code bit2char refuses (4-16) under an op that carries read bases -- whichever comes first along the CIGAR.  Bases are
 * taken where the CIGAR says (also behind l_seq; here not past the record); bases the CIGAR does not reach are never looked at. */
static void check_like_new_readaln(const bam_record* b)
{
    const uint8_t* cig = BAMR_CIGAR(b);
    const uint8_t* seq = BAMR_SEQ(b);
    const int64_t avail = 2 * (int64_t)((b->data + b->l_data) - seq);
    int64_t q = 0;
    for (int i = 0; i < b->n_cigar; i++) {
        const uint32_t w = bamr_cigar_at(cig, i);
        const int op = CIG_OP(w);
        const int64_t l = CIG_LEN(w);
        if (op == OP_N) fatalf("Implement new_readseg_bam:164");
        if (op == OP_H) fatalf("Implement new_readseg_bam:176");
        if (op == OP_P) fatalf("Implement new_readseg_bam:179");
        if (op > OP_X) fatalf("Unhandled cigar operation");
        if (op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X) {
            for (int64_t j = q; j < q + l && j < avail; j++)
                (void)bit2char((seq[j >> 1] >> ((~j & 1) << 2)) & 15);      /* exits with the reference's message on a code it refuses */
            q += l;
        }
    }
}

/* the l_seq bases of a record new_readaln has accepted: a code outside the CIGAR's reach is never decoded there, '?' here */
static char* decode_bases_checked(const bam_record* b)
{
    static const char dec[16] = { '?', 'A', 'C', '?', 'G', '?', '?', '?', 'T', '?', '?', '?', '?', '?', '?', 'N' };
    char* s = xmalloc((size_t)b->l_seq + 1);
    const uint8_t* q = BAMR_SEQ(b);
    for (int i = 0; i < b->l_seq; i++) s[i] = dec[BAMR_SEQI(q, i) & 15];
    s[b->l_seq] = 0;
    return s;
}

static char* decode_bases(const bam_record* b)
{
    char* s = xmalloc((size_t)b->l_seq + 1);
    const uint8_t* q = BAMR_SEQ(b);
    for (int i = 0; i < b->l_seq; i++) s[i] = bit2char(BAMR_SEQI(q, i));
    s[b->l_seq] = 0;
    return s;
}

static void revcomp_inplace(char* s)
{
    /* reverse_complement_string, src/sequences.c:204-220 with the table at 22-26 */
    const size_t n = strlen(s);
    for (size_t i = 0; i < n / 2; i++) { const char t = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = t; }
    for (size_t i = 0; i < n; i++) { const char c = kRevcomp[(unsigned char)s[i]]; s[i] = c ? c : ' '; }
}

/* new_readaln for an aligned record (src/readaln.c:192-239): CIGAR ops verbatim; N/H/P are
 * "Implement" fatals there (new_readseg_bam 163-180) */
static seglist seglist_from_record(const bam_record* b)
{
    seglist s;
    s.ref_start = b->pos;
    s.n = b->n_cigar;
    s.ops = xmalloc(sizeof(uint32_t) * (size_t)(b->n_cigar ? b->n_cigar : 1));
    const uint8_t* cig = BAMR_CIGAR(b);
    check_like_new_readaln(b);
    for (int i = 0; i < b->n_cigar; i++) s.ops[i] = bamr_cigar_at(cig, i);
    s.bases = decode_bases_checked(b);
    return s;
}

static seglist seglist_copy(const seglist* a)
{
    seglist s = *a;
    s.ops = xmalloc(sizeof(uint32_t) * (size_t)(a->n ? a->n : 1));
    memcpy(s.ops, a->ops, sizeof(uint32_t) * (size_t)a->n);
    s.bases = xstrdup(a->bases);
    return s;
}

/* What the VCF output reads of a split-read evidence object is its numbers; the read's name is printed by `-o detailed` only, and
 * its bases are read by `-o detailed` and by voted_consensus (an INSERTED segment's bases, src/variant.c:52-113).  The device
 * pipeline builds tens of millions of these objects per run: there they carry neither where nothing reads them. */
static int g_lean_evidence;
static seglist seglist_copy_for_evidence(const seglist* a, int seg)
{
    if (!g_lean_evidence || CIG_OP(a->ops[seg]) != OP_D) return seglist_copy(a);
    seglist s = *a;
    s.ops = xmalloc(sizeof(uint32_t) * (size_t)(a->n ? a->n : 1));
    memcpy(s.ops, a->ops, sizeof(uint32_t) * (size_t)a->n);
    s.bases = NULL;
    return s;
}

static void seglist_free(seglist* s) { free(s->ops); free(s->bases); s->ops = NULL; s->bases = NULL; s->n = 0; }

static int seglist_first_start(const seglist* s) { return s->ref_start; }

static int seglist_last_end(const seglist* s)
{
    int r = s->ref_start;
    for (int i = 0; i < s->n; i++) {
        const int op = CIG_OP(s->ops[i]);
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) r += CIG_LEN(s->ops[i]);
    }
    return r;   /* end of the last segment: segments that consume no reference end where they start */
}

/* flank / difference reductions of print_variants and print_vcf_output over the segments
 * [from,to) of a list (src/variant.c:217-274 and 704-767) */
static void seg_reduce(const seglist* a, int from, int to, const char* ref,
                       int32_t* flank, int32_t* nd_print, int32_t* nd_filter)
{
    int refpos = a->ref_start, readpos = 0;
    for (int i = 0; i < a->n; i++) {
        const int op = CIG_OP(a->ops[i]), len = CIG_LEN(a->ops[i]);
        if (i >= from && i < to) {
            switch (op) {
            case OP_M:
                *flank += len;
                for (int j = 0; j < len; j++)
                    if (a->bases[readpos + j] != ref[refpos + j]) { *nd_print += 1; *nd_filter += 1; }
                break;
            case OP_EQ: *flank += len; break;
            case OP_X: *flank += len; *nd_print += len; *nd_filter += len; break;
            case OP_I: *flank += len; *nd_print += len; *nd_filter += len; break;
            case OP_D: *nd_print += len; *nd_filter += len; break;
            case OP_S: *nd_filter += len; break;
            default: fatalf("unhandled BAM operation");
            }
        }
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) refpos += len;
        if (op != OP_D) readpos += len;
    }
}

/* ---------------------------------------------------------------- evidence -- */

static evidence_t* evidence_new_sr(const seglist* whole, int seg, int cls, char strand, uint8_t qual,
                                   const char* qname, const char* ref)
{
    /* new_evidence for SPLIT_READ (src/evidence.c:4-34): aln1 = segments before, aln2 = the
     * indel segment, aln3 = the rest; b1/b2 = the segment's start/end */
    evidence_t* e = xcalloc(1, sizeof *e);
    e->type = EV_SPLIT_READ; e->cls = cls; e->strand = strand; e->qual = qual;
    e->qname = g_lean_evidence ? NULL : xstrdup(qname);
    e->aln = seglist_copy_for_evidence(whole, seg);
    e->seg = seg;
    int refpos = whole->ref_start;
    for (int i = 0; i < seg; i++) {
        const int op = CIG_OP(whole->ops[i]);
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) refpos += CIG_LEN(whole->ops[i]);
    }
    e->b1 = refpos;
    e->b2 = (CIG_OP(whole->ops[seg]) == OP_D) ? refpos + CIG_LEN(whole->ops[seg]) : refpos;
    seg_reduce(whole, 0, seg, ref, &e->lflank, &e->nd_print, &e->nd_filter);
    seg_reduce(whole, seg + 1, whole->n, ref, &e->rflank, &e->nd_print, &e->nd_filter);
    return e;
}

static void evidence_free(evidence_t* e)
{
    if (!e) return;
    free(e->qname);
    seglist_free(&e->aln);
    if (e->aln3.ops || e->aln3.bases) seglist_free(&e->aln3);
    free(e);
}

/* check_variants (src/indelminer.c:285-337): evidence from the aligner's own CIGAR.
 * Returned in segment order (left to right); out[] must hold rln->n entries. */
static int check_variants(const seglist* rln, char strand, uint8_t qual, const char* qname, const char* ref,
                          evidence_t** out)
{
    uint32_t rpos = 0, tpos = 0;
    for (int i = 0; i < rln->n; i++) {
        const int op = CIG_OP(rln->ops[i]);
        if (op == OP_EQ || op == OP_X || op == OP_M || op == OP_I) tpos += (uint32_t)CIG_LEN(rln->ops[i]);
    }
    int n = 0;
    for (int i = 0; i < rln->n; i++) {
        const int op = CIG_OP(rln->ops[i]);
        if (op == OP_D || op == OP_I) {
            if (rpos > O.ethreshold_vcfcheck && (tpos - rpos) > O.ethreshold_vcfcheck)
                out[n++] = evidence_new_sr(rln, i, op == OP_D ? CLS_DELETION : CLS_INSERTION, strand, qual, qname, ref);
        } else if (op == OP_M || op == OP_EQ || op == OP_X) {
            rpos += (uint32_t)CIG_LEN(rln->ops[i]);
        } else if (op == OP_S) {
            forceassert(i == 0 || i == rln->n - 1);
        } else fatalf("unknown cigar op");
    }
    return n;
}

/* ------------------------------------------------------------------ pass A -- */

enum { ITEM_CAND = 1, ITEM_PE = 2 };
#define EV_PHANTOM (-1)         /* a paired-read entry of a stage that stands for the entries waiting for the contig's end (stage_leftovers) */

typedef struct {
    int kind;
    int cand;                   /* ITEM_CAND: index into the candidate batch */
    evidence_t** bwa; int nbwa; /* ITEM_CAND: CIGAR-derived fallback (src/indelminer.c:504-510) */
    evidence_t* pe;             /* ITEM_PE */
} item_t;

typedef struct { int64_t n_items; int marker; int32_t tid; } flush_t;

typedef struct {
    /* candidate batch of the contig, struct of arrays for im_realign_batch */
    int32_t n, cap;
    uint8_t* bases; int64_t bases_len, bases_cap;
    int64_t* base_off;
    int32_t *tid, *anchor, *range_max;
    char** qname; char* strand; uint8_t* qual;
} cand_batch;

typedef struct {
    im_ctx* gpu;
    bam_header* hdr;
    char** sequences; int64_t* seqlen;
    qhash* insertlengths;
    char rg_last_name[256]; const int32_t* rg_last_val; const int32_t* rg_tmp_val;    /* one-entry cache of the lookup above */
    /* the GPU context is opened and the reference uploaded by a helper thread while the main
     * thread decodes the BAM (pass A needs no GPU); gpu_wait() joins it before the first GPU call */
    pthread_t gpu_thread;
    int gpu_pending, gpu_rc;
    pthread_mutex_t gpu_mu; pthread_cond_t gpu_cv; int seq_ready; int ctx_ready, ctx_rc;   /* ctx_ready: im_ctx_create has returned (gpu_rc says how) */      /* the helper opens the context at once and uploads the reference when the FASTA is in */
    char gpu_err[512];
    qhash* readpairs;
    const char* bam_name;
    bai_index* idx;
    int64_t numread;
    item_t* items; int64_t n_items, cap_items;
    flush_t* flushes; int n_flushes, cap_flushes;
    cand_batch cb;
    evidence_t** pending; int64_t n_pending, cap_pending;
    int64_t arrival;
    /* match segments of the contig's pileup-eligible records, for the device depth array */
    int32_t *seg_start, *seg_len; int64_t n_seg, cap_seg;
    int depth_tid;              /* contig whose depth array is resident on the device, -1 = none */
    int pipe_mode;              /* device pipeline: depth queries go to the genome-wide array */
    int marker_floor;           /* multi-GPU: smallest start of a stale pair-table entry of an earlier contig on another rank */
    /* live entries of the pair table (find_marker walks these) */
    evidence_t** live; int32_t n_live, cap_live;
    int live_changed;           /* set by live_add / live_del: the walk logs the list's minimum when it moves */
} driver;

static void gpu_wait(driver* d);
static void print_vcf_preamble(void);
static int g_mg_rank = 0, g_mg_local = -1;
static int g_mg_parts = 0;              /* a multi-GPU run: output goes to per-contig parts that rank 0 puts together */
static char g_mg_header_path[512] = "";

static void cb_push(cand_batch* cb, const char* bases, int32_t tid, int32_t anchor, int32_t range_max,
                    const char* qname, char strand, uint8_t qual)
{
    if (cb->n == cb->cap) {
        cb->cap = cb->cap ? cb->cap * 2 : 4096;
        cb->base_off = xrealloc(cb->base_off, sizeof(int64_t) * ((size_t)cb->cap + 1));
        cb->tid = xrealloc(cb->tid, sizeof(int32_t) * (size_t)cb->cap);
        cb->anchor = xrealloc(cb->anchor, sizeof(int32_t) * (size_t)cb->cap);
        cb->range_max = xrealloc(cb->range_max, sizeof(int32_t) * (size_t)cb->cap);
        cb->qname = xrealloc(cb->qname, sizeof(char*) * (size_t)cb->cap);
        cb->strand = xrealloc(cb->strand, (size_t)cb->cap);
        cb->qual = xrealloc(cb->qual, (size_t)cb->cap);
    }
    const size_t l = strlen(bases);
    if (cb->bases_len + (int64_t)l + 16 > cb->bases_cap) {
        cb->bases_cap = (cb->bases_cap ? cb->bases_cap * 2 : (1 << 20)) + (int64_t)l;
        cb->bases = xrealloc(cb->bases, (size_t)cb->bases_cap);
    }
    memcpy(cb->bases + cb->bases_len, bases, l);
    cb->base_off[cb->n] = cb->bases_len;
    cb->bases_len += (int64_t)l;
    cb->base_off[cb->n + 1] = cb->bases_len;
    cb->tid[cb->n] = tid; cb->anchor[cb->n] = anchor; cb->range_max[cb->n] = range_max;
    cb->qname[cb->n] = xstrdup(qname); cb->strand[cb->n] = strand; cb->qual[cb->n] = qual;
    cb->n++;
}

static void cb_reset(cand_batch* cb)
{
    for (int32_t i = 0; i < cb->n; i++) free(cb->qname[i]);
    cb->n = 0; cb->bases_len = 0;
}

static item_t* push_item(driver* d)
{
    if (d->n_items == d->cap_items) {
        d->cap_items = d->cap_items ? d->cap_items * 2 : 4096;
        d->items = xrealloc(d->items, sizeof(item_t) * (size_t)d->cap_items);
    }
    item_t* it = &d->items[d->n_items++];
    memset(it, 0, sizeof *it);
    return it;
}

/* find_marker (src/indelminer.c:211-233): smallest aln1->start among the pairs still waiting for a mate.
 * The reference walks all 2^20 bins of the pair table; the live entries are kept in a list here. */
static int find_marker_live(const driver* d)
{
    int m = INT_MAX;
    for (int32_t i = 0; i < d->n_live; i++)
        if (seglist_first_start(&d->live[i]->aln) < m) m = seglist_first_start(&d->live[i]->aln);
    return m;
}

/* find_mate_rln (src/indelminer.c:256-280): look the mate up in the BAM when it is not in the
 * pair table (region runs).  Returns 1 and fills *out when found. */
static int find_mate(driver* d, int32_t tid, int32_t pos, char want_index, const char* qname, seglist* out, char* strand)
{
    bgzf_reader* r = bgzf_open(d->bam_name);
    if (!r) return 0;
    bam_header* h = bam_header_load(r);
    bam_region_iter it;
    bam_record b; memset(&b, 0, sizeof b);
    int found = 0;
    if (h && bam_region_begin(&it, r, d->idx, tid, pos, pos + 1) == 0) {
        while (bam_region_next(&it, &b) == 1) {
            if (strcmp(BAMR_QNAME(&b), qname) != 0) continue;
            const char index = (b.flag & 0x40) ? '1' : '2';
            if (index != want_index) continue;
            if (b.flag & 0x4) continue;     /* check_for_mate goes through new_readaln: unaligned mates leave segments NULL-start; treated as not found */
            if (found) seglist_free(out);   /* a later hit overwrites (src/indelminer.c:243-251) */
            *out = seglist_from_record(&b);
            *strand = (b.flag & 0x10) ? '-' : '+';
            found = 1;
        }
    }
    free(b.data);
    bam_header_free(h);
    bgzf_close(r);
    return found;
}

static int mate_mapq(const bam_record* b, int strict)
{
    /* MQ tag if present, else the read's own MAPQ (src/indelminer.c:388-400,463-472,592-601) */
    const uint8_t* p = bam_aux_find(b, "MQ");
    if (!p) return b->mapq;
    if (strict) forceassert(p[0] == 'I' || p[0] == 'i' || p[0] == 'C' || p[0] == 'c' || p[0] == 'S' || p[0] == 's');
    return bam_aux_int(p);
}

static void live_add(driver* d, evidence_t* e)
{
    if (d->n_live == d->cap_live) { d->cap_live = d->cap_live ? d->cap_live * 2 : 1024; d->live = xrealloc(d->live, sizeof(evidence_t*) * (size_t)d->cap_live); }
    e->live_slot = d->n_live;
    d->live[d->n_live++] = e;
    d->live_changed = 1;
}
static void live_del(driver* d, evidence_t* e)
{
    if (!e) return;
    const int32_t s = e->live_slot;
    if (s < 0 || s >= d->n_live || d->live[s] != e) return;
    d->live[s] = d->live[--d->n_live];
    d->live[s]->live_slot = s;
    e->live_slot = -1;
    d->live_changed = 1;
}

/* the discordant-pair branch of fetch_func (src/indelminer.c:516-615): the first mate waits in the pair
 * table, the second completes the evidence.  Returns the completed evidence or NULL. */
static evidence_t* discordant_pair(driver* d, const bam_record* b, const int32_t* range)
{
    const int flag = b->flag;
    const int is_rc = (flag & 0x10) == 0x10, is_mate_rc = (flag & 0x20) == 0x20;
    const char* qname = BAMR_QNAME(b);
    evidence_t* done = NULL;
    if (abs(b->isize) > range[1] && (uint32_t)abs(b->isize) < O.maxpedelsize && is_rc != is_mate_rc) {
        if (b->pos < b->mpos) {
            evidence_t* e = xcalloc(1, sizeof *e);
            e->type = EV_PAIRED_READ; e->cls = CLS_DELETION;
            e->qual = b->mapq; e->strand = is_rc ? '-' : '+';
            e->qname = xstrdup(qname);
            e->aln = seglist_from_record(b);
            qhash_add(d->readpairs, qname, b->l_qname, e);
            live_add(d, e);
        } else {
            qbin* hb = qhash_lookup(d->readpairs, qname, b->l_qname);
            evidence_t* e = hb ? hb->val : NULL;
            int skip = 0;
            evidence_t* dropped = NULL;         /* completed, but neither mate passes -q: freed once it has left the table */
            if (!e) {
                seglist m; char mstrand = '+';
                const char want = (flag & 0x40) ? '2' : '1';
                if (!find_mate(d, b->mtid, b->mpos, want, qname, &m, &mstrand)) skip = 1;
                else {
                    e = xcalloc(1, sizeof *e);
                    e->type = EV_PAIRED_READ; e->cls = CLS_DELETION;
                    e->qual = 0;            /* find_mate_rln never copies the mate's MAPQ (src/indelminer.c:243-251) */
                    e->strand = mstrand;
                    e->qname = xstrdup(qname);
                    e->aln = m;
                    if (b->mapq < e->qual) e->qual = b->mapq;
                    qhash_add(d->readpairs, qname, b->l_qname, e);
                    live_add(d, e);
                }
            }
            if (!skip) {
                e->aln3 = seglist_from_record(b);
                e->b1 = seglist_last_end(&e->aln);
                e->b2 = seglist_first_start(&e->aln3);
                e->mindelsize = abs(b->isize) - range[1];
                e->max = range[1];
                const int smq = b->mapq, mmq = mate_mapq(b, 0);
                if (smq >= O.qthreshold || mmq >= O.qthreshold) {
                    const char* r = d->sequences[b->tid];
                    seg_reduce(&e->aln, 0, e->aln.n, r, &e->lflank, &e->nd_print, &e->nd_filter);
                    seg_reduce(&e->aln3, 0, e->aln3.n, r, &e->rflank, &e->nd_print, &e->nd_filter);
                    done = e;
                } else dropped = e;
            }
            live_del(d, qhash_remove(d->readpairs, qname, b->l_qname));
            if (dropped) evidence_free(dropped);
        }
    }
    return done;
}

/* must_find_hashtable(insertlengths, rgname) (src/indelminer.c:369-376) with a one-entry cache */
static const int32_t* record_range(driver* d, const bam_record* b)
{
    const uint8_t* rg = bam_aux_find(b, "RG");
    const char* rgname = "generic";
    if (rg) rgname = bam_aux_str(rg);
    if (d->rg_last_val == NULL || strcmp(rgname, d->rg_last_name) != 0) {
        qbin* rb = qhash_lookup(d->insertlengths, rgname, (int)strlen(rgname));
        if (!rb) fatalf("did not find %s in the hash", rgname);
        snprintf(d->rg_last_name, sizeof d->rg_last_name, "%s", rgname);
        d->rg_last_val = strlen(rgname) < sizeof d->rg_last_name ? rb->val : NULL;     /* over-long names are not cached */
        d->rg_tmp_val = rb->val;
    } else d->rg_tmp_val = d->rg_last_val;
    return d->rg_tmp_val;
}

/* fetch_func (src/indelminer.c:339-673) for one record, pass A part */
static void dispatch_record(driver* d, const bam_record* b)
{
    const int flag = b->flag;
    if (flag & 0x100) return;
    if (flag & 0x200) return;
    if (flag & 0x400) return;
    if (flag & 0x800) return;
    const int is_aligned = (flag & 0x4) == 0, is_mate_aligned = (flag & 0x8) == 0;
    const int is_se = (flag & 0x1) == 0, is_proper_pair = (flag & 0x2) == 0x2;
    const int is_rc = (flag & 0x10) == 0x10, is_mate_rc = (flag & 0x20) == 0x20;
    if (is_se) return;
    if (is_aligned && is_mate_aligned && b->tid != b->mtid) return;

    const int32_t* range = record_range(d, b);
    const char* qname = BAMR_QNAME(b);

    if (is_aligned && !is_mate_aligned) {
        /* dealt with at the mate */
    } else if (!is_aligned && is_mate_aligned) {
        const int mmq = mate_mapq(b, 1);
        if (mmq >= O.qthreshold) {
            char* bases = decode_bases(b);
            char strand = is_rc ? '-' : '+';
            if (!is_mate_rc) { revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
            item_t* it = push_item(d);
            it->kind = ITEM_CAND; it->cand = d->cb.n;
            cb_push(&d->cb, bases, b->mtid, b->mpos, range[1], qname, strand, (uint8_t)mmq);
            free(bases);
        }
    } else if (is_aligned && is_mate_aligned && is_proper_pair) {
        /* the CIGAR is judged on the record itself; the segment list (two allocations and the base
         * decode) is only built for the few reads that go on.  The reference builds it for every
         * proper pair (new_unaligned_readaln, src/indelminer.c:430) and would stop on N / H / P
         * there, so those checks stay in front. */
        const char strand0 = is_rc ? '-' : '+';
        uint32_t numcdels = 0, numcins = 0, numcsclip = 0;
        int is_threeprime_clip = 0;
        const uint8_t* cig = BAMR_CIGAR(b);
        const int ncig = b->n_cigar;
        check_like_new_readaln(b);
        for (int i = 0; i < ncig; i++) {
            const int op = CIG_OP(bamr_cigar_at(cig, i));
            if (op == OP_D) numcdels++;
            if (op == OP_I) numcins++;
            if (op == OP_S) numcsclip++;
            if (((strand0 == '+' && i == ncig - 1) || (strand0 == '-' && i == 0)) && op == OP_S) is_threeprime_clip = 1;
        }
        const uint32_t numinteresting = numcdels + numcins + numcsclip;
        seglist rln; rln.ops = NULL; rln.bases = NULL; rln.n = 0; rln.ref_start = 0;
        if (numinteresting > 0) {
            if (((numcsclip == 0) || (numcsclip == 1 && is_threeprime_clip)) && numcdels == 0 && numcins == 0) {
                /* nothing to do (src/indelminer.c:457-460) */
            } else {
                const int mmq = mate_mapq(b, 0);
                if (mmq >= O.qthreshold) {
                    rln = seglist_from_record(b);
                    const char* own_ref = d->sequences[b->tid];
                    evidence_t** bwa = xmalloc(sizeof(evidence_t*) * (size_t)(rln.n ? rln.n : 1));
                    const int nbwa = check_variants(&rln, strand0, b->mapq, qname, own_ref, bwa);
                    char* bases = decode_bases(b);
                    char strand = strand0;
                    if ((is_rc && is_mate_rc) || (!is_rc && !is_mate_rc)) { revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
                    item_t* it = push_item(d);
                    it->kind = ITEM_CAND; it->cand = d->cb.n; it->bwa = bwa; it->nbwa = nbwa;
                    cb_push(&d->cb, bases, b->mtid, b->mpos, range[1], qname, strand, b->mapq);
                    free(bases);
                }
            }
        }
        seglist_free(&rln);
    } else if (is_aligned && is_mate_aligned && !is_proper_pair) {
        evidence_t* e = discordant_pair(d, b, range);
        if (e) { item_t* it = push_item(d); it->kind = ITEM_PE; it->pe = e; }
    }

    if ((++d->numread % READCHUNK) == 0) {
        timestamp("Read %ld reads", (long)d->numread);
        int marker = find_marker_live(d);
        if (b->pos < marker) marker = b->pos;
        if (d->n_flushes == d->cap_flushes) {
            d->cap_flushes = d->cap_flushes ? d->cap_flushes * 2 : 64;
            d->flushes = xrealloc(d->flushes, sizeof(flush_t) * (size_t)d->cap_flushes);
        }
        d->flushes[d->n_flushes].n_items = d->n_items;
        d->flushes[d->n_flushes].marker = marker;
        d->flushes[d->n_flushes].tid = b->tid;
        d->n_flushes++;
    }
}

/* ------------------------------------------------------- variants (host) -- */

static void vl_push(variant_list* l, variant_t* v)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 64; l->v = xrealloc(l->v, sizeof(variant_t*) * (size_t)l->cap); }
    l->v[l->n++] = v;
}

static void variant_free(variant_t* v) { if (v) { free(v->evidence); free(v); } }

/* stable insertion of sort_by_position (src/variant.c:15-25,40-44): glibc qsort is a stable
 * merge sort for these sizes, so equal (start,stop) keep their list order */
static int cmp_variant_pos(const variant_t* a, const variant_t* b)
{
    if (a->start == b->start) return (int)a->stop - (int)b->stop;
    return (int)a->start - (int)b->start;
}
static void sort_variants(variant_list* l)
{
    /* merge sort on pointers, stable */
    if (l->n < 2) return;
    variant_t** tmp = xmalloc(sizeof(variant_t*) * (size_t)l->n);
    for (int w = 1; w < l->n; w *= 2) {
        for (int lo = 0; lo < l->n; lo += 2 * w) {
            int mid = lo + w < l->n ? lo + w : l->n, hi = lo + 2 * w < l->n ? lo + 2 * w : l->n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (cmp_variant_pos(l->v[j], l->v[i]) < 0) ? l->v[j++] : l->v[i++];
            while (i < mid) tmp[k++] = l->v[i++];
            while (j < hi) tmp[k++] = l->v[j++];
        }
        memcpy(l->v, tmp, sizeof(variant_t*) * (size_t)l->n);
    }
    free(tmp);
}

/* voted_consensus (src/variant.c:52-113) with its 16-bin table walk */
static char* voted_consensus(evidence_t** ev, uint32_t nsupport, int* maxsize)
{
    qhash* counts = qhash_new(4);
    uint32_t* intcounts = xcalloc(nsupport, sizeof(uint32_t));
    uint32_t indx = 0;
    int size = 0;
    char** keep = xcalloc(nsupport, sizeof(char*));
    for (uint32_t i = 0; i < nsupport; i++) {
        forceassert(ev[i]->type == EV_SPLIT_READ);
        const seglist* a = &ev[i]->aln;
        /* aln2->sequence: the segment's read bases, '-' for a deletion (src/readaln.c:58-73) */
        int readpos = 0;
        for (int s = 0; s < ev[i]->seg; s++) if (CIG_OP(a->ops[s]) != OP_D) readpos += CIG_LEN(a->ops[s]);
        const int op = CIG_OP(a->ops[ev[i]->seg]), len = CIG_LEN(a->ops[ev[i]->seg]);
        char* seq = xmalloc((size_t)len + 1);
        if (op == OP_D) memset(seq, '-', (size_t)len); else memcpy(seq, a->bases + readpos, (size_t)len);
        seq[len] = 0;
        keep[i] = seq;
        const int inslen = (int)strlen(seq);
        if (inslen > size) size = inslen;
        if (qhash_lookup(counts, seq, inslen) == NULL) { qhash_add(counts, seq, inslen, intcounts + indx); indx++; }
        qbin* b = qhash_lookup(counts, seq, inslen);
        *(uint32_t*)b->val += 1;
    }
    *maxsize = size;
    uint32_t maximumcount = 0;
    const char* consensus = NULL;
    for (uint32_t j = 0; j <= counts->mask; j++)
        for (qbin* it = counts->bins[j]; it; it = it->next)
            if (*(uint32_t*)it->val > maximumcount) { maximumcount = *(uint32_t*)it->val; consensus = it->name; }
    forceassert(consensus != NULL);
    char* rt = xstrdup(consensus);
    for (uint32_t i = 0; i < nsupport; i++) free(keep[i]);
    free(keep); free(intcounts);
    qhash_free(counts, NULL);
    return rt;
}

/* move_boundaries (src/variant.c:923-991).  The reference indexes the contig without bounds
 * checks; reads past either end are stopped here (they are out-of-bounds reads there). */
static void move_boundaries(variant_t* vs, const char* reference, int64_t reflen)
{
    uint32_t lw = 0, rw = 0;
    if (vs->type == CLS_INSERTION) {
        int maxinsertsize;
        char* consensus = voted_consensus(vs->evidence, vs->support, &maxinsertsize);
        const size_t cl = strlen(consensus);
        while ((int64_t)vs->start - (int64_t)cl - (int64_t)lw >= 0 && cl > 0 &&
               strncmp(consensus, reference + vs->start - cl - lw, cl) == 0) lw += (uint32_t)cl;
        uint32_t shift = 0;
        while (shift < cl && (int64_t)vs->start - 1 - (int64_t)lw >= 0 &&
               consensus[cl - shift - 1] == reference[vs->start - 1 - lw]) { lw++; shift++; }
        while (cl > 0 && (int64_t)vs->stop + rw < reflen && strncmp(consensus, reference + vs->stop + rw, cl) == 0) rw += (uint32_t)cl;
        shift = 0;
        while (shift < cl && (int64_t)vs->stop + rw < reflen && consensus[shift] == reference[vs->stop + rw]) { rw++; shift++; }
        free(consensus);
    } else if (vs->type == CLS_DELETION) {
        while ((int64_t)vs->start - 1 - (int64_t)lw >= 0 && reference[vs->start - 1 - lw] == reference[vs->stop - 1 - lw]) lw++;
        while ((int64_t)vs->stop + rw < reflen && reference[vs->start + rw] == reference[vs->stop + rw]) rw++;
    }
    vs->lw = lw; vs->rw = rw;
}

/* merge_variants (src/variant.c:1029-1225) over arrays.  in: sorted list; out: sorted list. */
static void merge_variants(variant_list* pvs, const char* reference, int64_t reflen, int join_sr_pe)
{
    if (pvs->n == 0) return;
    for (int i = 0; i < pvs->n; i++)
        if (pvs->v[i]->evdnctype == EV_SPLIT_READ) move_boundaries(pvs->v[i], reference, reflen);

    variant_list vs = {0}, pe = {0};
    for (int i = 0; i < pvs->n; i++) vl_push(pvs->v[i]->evdnctype == EV_PAIRED_READ ? &pe : &vs, pvs->v[i]);
    sort_variants(&vs);

    /* SR variants with the same type and the same shifted boundaries collapse into the first
     * (1079-1123).  Each iter1 scans forward while start <= iter1.stop + iter1.rw. */
    for (int i = 0; i < vs.n; i++) {
        variant_t* a = vs.v[i];
        int j = i + 1;
        while (j < vs.n && vs.v[j]->start <= a->stop + a->rw) {
            variant_t* b = vs.v[j];
            forceassert(a->evdnctype == EV_SPLIT_READ);
            forceassert(b->evdnctype == EV_SPLIT_READ);
            if (b->type == a->type && (a->start - a->lw) == (b->start - b->lw) && (a->stop + a->rw) == (b->stop + b->rw)) {
                /* mergeSRvariants (993-1023): a's coordinates, evidence of a then of b */
                a->evidence = xrealloc(a->evidence, sizeof(evidence_t*) * (size_t)(a->support + b->support));
                memcpy(a->evidence + a->support, b->evidence, sizeof(evidence_t*) * (size_t)b->support);
                a->support += b->support;
                variant_free(b);
                memmove(&vs.v[j], &vs.v[j + 1], sizeof(variant_t*) * (size_t)(vs.n - j - 1));
                vs.n--;
                j = i + 1;          /* the reference restarts its scan behind the merged node (1113-1118) */
                continue;
            }
            j++;
        }
    }

    if (!join_sr_pe) {
        for (int i = 0; i < pe.n; i++) vl_push(&vs, pe.v[i]);
        sort_variants(&vs);
        free(pvs->v); free(pe.v);
        *pvs = vs;
        return;
    }

    /* paired-read variants join the best-overlapping split-read variant (1143-1217).  vs is a
     * list whose HEAD receives every PE variant that did not merge; the candidate scan walks
     * that list from the head and stops at the first start > stop, prepended PE variants
     * included -- kept as is. */
    variant_t** lst = xmalloc(sizeof(variant_t*) * (size_t)(vs.n + pe.n + 1));
    int nl = vs.n;
    memcpy(lst, vs.v, sizeof(variant_t*) * (size_t)vs.n);
    for (int p = 0; p < pe.n; p++) {
        variant_t* it1 = pe.v[p];
        uint32_t overlap = 0;
        variant_t* cand = NULL;
        for (int q = 0; q < nl; q++) {
            variant_t* it2 = lst[q];
            if (it2->start > it1->stop) break;
            if (it2->evdnctype == EV_PAIRED_READ) continue;
            uint32_t olap = 0;
            if (it1->start >= it2->start && it1->start < it2->stop)
                olap = (it1->stop < it2->stop ? it1->stop : it2->stop) - it1->start;
            else if (it2->start >= it1->start && it2->start < it1->stop)
                olap = (it1->stop < it2->stop ? it1->stop : it2->stop) - it2->start;
            const double f = (olap * 100.0 / (double)(it1->stop - it1->start)) + (olap * 100.0 / (double)(it2->stop - it2->start));
            const uint32_t olapf = isfinite(f) ? (uint32_t)f : 0u;     /* NaN/inf convert to 0 on x86-64 */
            if (olapf > overlap) { overlap = olapf; cand = it2; }
        }
        int tomerge = 1;
        if (cand) {
            const int size = (int)cand->stop - (int)cand->start;
            for (uint32_t i = 0; i < it1->support; i++) if (size < it1->evidence[i]->mindelsize) { tomerge = 0; break; }
        }
        if (cand && tomerge) {
            cand->evidence = xrealloc(cand->evidence, sizeof(evidence_t*) * (size_t)(cand->support + it1->support));
            memcpy(cand->evidence + cand->support, it1->evidence, sizeof(evidence_t*) * (size_t)it1->support);
            cand->support += it1->support;
            if ((cand->evdnctype == EV_PAIRED_READ && it1->evdnctype == EV_SPLIT_READ) ||
                (cand->evdnctype == EV_SPLIT_READ && it1->evdnctype == EV_PAIRED_READ)) cand->evdnctype = EV_COMPOSITE;
            variant_free(it1);
        } else {
            memmove(lst + 1, lst, sizeof(variant_t*) * (size_t)nl);
            lst[0] = it1;
            nl++;
        }
    }
    free(pvs->v); free(vs.v); free(pe.v);
    pvs->v = lst; pvs->n = nl; pvs->cap = nl;
    sort_variants(pvs);
}

/* ---- region depth, calculate_cov_params (src/shared.c:178-212) ---- */
/* Loci at which samtools' pileup, the reference's source of DP= (src/shared.c:178-212), may have stopped taking records: a record
 * is left out when it starts at the position the iterator stands on and more than 8000 nodes are allocated
 * (src/samtools-0.1.19/bam_pileup.c:172,244).  The device's depth array counts every record, so a query whose deepest position
 * reaches this bound is answered from the file with that rule (region_depth_from_bam). */
#define DP_DEEP_LOCUS 4000

static uint32_t region_depth_from_bam(driver* d, int32_t tid, int32_t start, int32_t stop);

static uint32_t region_depth(driver* d, int32_t tid, int32_t start, int32_t stop)
{
    if (stop <= start) return 0;
    if (d->depth_tid == tid) {
        /* the device holds the contig's depth array (im_depth_build in run_contig) */
        uint32_t sum = 0;
        gpu_wait(d);
        pthread_mutex_lock(&g_query_mu);
        uint32_t deepest = 0;
        const int qrc = d->pipe_mode ? im_depth_query_max_tid(d->gpu, tid, 1, &start, &stop, &sum, &deepest) : im_depth_query(d->gpu, 1, &start, &stop, &sum);
        pthread_mutex_unlock(&g_query_mu);
        if (qrc != IM_OK)
            fatalf("im_depth_query: %s", im_last_error(d->gpu));
        if (deepest >= DP_DEEP_LOCUS) return region_depth_from_bam(d, tid, start, stop);      /* the pileup's cap may have applied */
        return (uint32_t)floor(sum * 1.0 / (uint32_t)(stop - start));
    }
    /* region runs (-c): the reference pileups the whole BAM around the variant, which can reach
     * outside the analysed region -- go to the file like it does */
    return region_depth_from_bam(d, tid, start, stop);
}

static uint32_t region_depth_from_bam(driver* d, int32_t tid, int32_t start, int32_t stop)
{
    /* pileup semantics (bam_pileup.c:67-143,238-265): records with flag & (0x4|0x100|0x200|0x400)
     * or tid < 0 are skipped; a position counts a read iff its covering op is M/=/X.
     * The iterator's buffer, restated: after a record starting at p has been taken, every position in front of p has been
     * served and the records that ended there are gone -- the buffer holds the taken records with end >= p.  The next record
     * is refused iff it starts at that same p and the pool counts more than 8000 nodes (the buffered records + the list's open
     * tail node + the dummy node, bam_pileup.c:164-166,201): its bases are not counted anywhere. */
    if (stop <= start) return 0;
    uint32_t* cov = xcalloc((size_t)(stop - start), sizeof(uint32_t));
    bgzf_reader* r = bgzf_open(d->bam_name);
    if (!r) fatalf("error in opening the file %s", d->bam_name);
    bam_header* h = bam_header_load(r);
    bam_region_iter it;
    bam_record b; memset(&b, 0, sizeof b);
    int32_t* heap = NULL; int64_t nh = 0, caph = 0;          /* ends of the buffered records, smallest on top */
    int32_t it_tid = 0, it_pos = 0;                          /* where the iterator stands (calloc'ed: contig 0, position 0) */
    if (h && bam_region_begin(&it, r, d->idx, tid, start, stop) == 0) {
        while (bam_region_next(&it, &b) == 1) {
            if (b.tid < 0 || (b.flag & (0x4 | 0x100 | 0x200 | 0x400))) continue;
            if (it_tid == b.tid && it_pos == b.pos && nh + 2 > 8000) continue;          /* bam_plp_push: not taken */
            const int32_t rend = bam_record_end(&b);
            const uint8_t* cig = BAMR_CIGAR(&b);
            int32_t x = b.pos;
            for (int k = 0; k < b.n_cigar; k++) {
                const int op = CIG_OP(bamr_cigar_at(cig, k)), len = CIG_LEN(bamr_cigar_at(cig, k));
                if (op == OP_M || op == OP_EQ || op == OP_X) {
                    int32_t lo = x < start ? start : x, hi = x + len > stop ? stop : x + len;
                    for (int32_t p = lo; p < hi; p++) cov[p - start]++;
                    x += len;
                } else if (op == OP_D || op == OP_N) x += len;
            }
            /* the node stays in the list when the record reaches beyond the iterator's position (bam_pileup.c:260-263) */
            if (rend > it_pos || b.tid > it_tid) {
                if (nh == caph) { caph = caph ? caph * 2 : 1024; heap = xrealloc(heap, sizeof(int32_t) * (size_t)caph); }
                int64_t i = nh++;
                while (i > 0 && heap[(i - 1) / 2] > rend) { heap[i] = heap[(i - 1) / 2]; i = (i - 1) / 2; }
                heap[i] = rend;
            }
            /* bam_plbuf_push drains: every position in front of this record's start is served, what ended there leaves */
            if (b.tid > it_tid || b.pos > it_pos) { it_tid = b.tid; it_pos = b.pos; }
            while (nh > 0 && heap[0] < it_pos) {
                const int32_t last = heap[--nh];
                int64_t i = 0;
                for (;;) {
                    int64_t c = 2 * i + 1;
                    if (c >= nh) break;
                    if (c + 1 < nh && heap[c + 1] < heap[c]) c++;
                    if (heap[c] >= last) break;
                    heap[i] = heap[c]; i = c;
                }
                if (nh > 0) heap[i] = last;
            }
        }
    }
    free(heap);
    free(b.data);
    bam_header_free(h);
    bgzf_close(r);
    uint64_t covsum = 0;
    for (int32_t i = 0; i < stop - start; i++) covsum += cov[i];
    free(cov);
    return (uint32_t)floor((uint32_t)covsum * 1.0 / (uint32_t)(stop - start));
}

/* print_vcf_output (src/variant.c:115-311) */
static void print_vcf_output(driver* d, const variant_t* v)
{
    const char* seq = d->sequences[v->tid];
    printf("%s\t%d\t.\t", d->hdr->target_name[v->tid], (int)(v->start - v->lw));
    int endpos = -1;
    if (v->type == CLS_DELETION) {
        const int reflength = (int)(v->stop + v->rw) - (int)(v->start - v->lw - 1);
        forceassert(reflength >= 1);
        const int altlength = (int)(v->start + v->rw) - (int)(v->start - v->lw - 1);
        forceassert(altlength >= 1);
        endpos = (int)(v->start - v->lw) + reflength - altlength + 1;
        for (int i = 0; i < (reflength - altlength + 1); i++) printf("%c", seq[v->start - v->lw - 1 + (uint32_t)i]);
        printf("\t");
        printf("%c\t", seq[v->start - v->lw - 1]);
    } else if (v->type == CLS_INSERTION) {
        const int reflength = (int)(v->stop + v->rw) - (int)(v->start - v->lw - 1);
        forceassert(reflength >= 1);
        int maxinsertsize = 0;
        char* consensus = voted_consensus(v->evidence, v->support, &maxinsertsize);
        const int altlength = reflength + (int)strlen(consensus) + (int)(v->stop + v->rw) - (int)v->start;
        forceassert(altlength >= 1);
        endpos = (int)(v->start - v->lw) + 1;
        printf("%c\t", seq[v->start - v->lw - 1]);
        printf("%c", seq[v->start - v->lw - 1]);
        printf("%s\t", consensus);
        free(consensus);
    } else fatalf("unhandled variant type");

    printf(".\t.\t%s;", v->type == CLS_DELETION ? "DELETION" : "INSERTION");
    if (v->evdnctype == EV_SPLIT_READ) printf("SPLIT_READ;");
    else if (v->evdnctype == EV_PAIRED_READ) printf("PAIRED_READ;");
    else if (v->evdnctype == EV_COMPOSITE) printf("COMPOSITE;");
    else fatalf("unknown evidence type for this variant");
    forceassert(endpos != -1);
    printf("NS=%u;END=%d;BP_END=%d", v->support, endpos, (int)(v->stop + v->rw + 1));

    uint32_t nf = 0, nr = 0;
    for (uint32_t i = 0; i < v->support; i++) {
        if (v->evidence[i]->strand == '+') nf++;
        else if (v->evidence[i]->strand == '-') nr++;
        else fatalf("unknown strand");
    }
    printf(";NFS=%u;NRS=%u", nf, nr);

    uint32_t maxtaild = 100;
    char* taildistances = xcalloc(maxtaild + 1, 1);
    uint32_t nut = 0, num_pe = 0, mq = 0, mq30 = 0, numdiffs = 0;
    int balance = INT_MAX, lflank = -1, rflank = -1;
    for (uint32_t i = 0; i < v->support; i++) {
        const evidence_t* e = v->evidence[i];
        mq += e->qual;
        if (e->qual >= 30) mq30++;
        const uint32_t ltmp = (uint32_t)e->lflank, rtmp = (uint32_t)e->rflank;
        numdiffs += (uint32_t)e->nd_print;
        const uint32_t taild = rtmp < ltmp ? rtmp : ltmp;
        if (taild > maxtaild) {
            taildistances = xrealloc(taildistances, taild + 1);
            memset(taildistances + maxtaild + 1, 0, taild - maxtaild);
            maxtaild = taild;
        }
        taildistances[taild] = '1';
        if (e->type == EV_PAIRED_READ) num_pe++;
        if (abs((int)(rtmp - ltmp)) < balance) { balance = abs((int)(rtmp - ltmp)); lflank = (int)ltmp; rflank = (int)rtmp; }
    }
    for (uint32_t i = 0; i < maxtaild; i++) if (taildistances[i] == '1') nut++;     /* i < maxtaild: src/variant.c:293-295 */
    nut += num_pe;
    printf(";UTAILS=%d;MQ=%d;MQ30=%d;DF=%d;DP=%d", (int)nut, (int)(mq * 1.0 / v->support), (int)mq30,
           (int)((numdiffs * 1.0 / v->support) + 0.5),
           v->dp_valid ? (int)v->dp_cached
                       : (int)region_depth(d, v->tid, (int32_t)(v->start - v->lw - 1), (int32_t)(v->stop + v->rw + 1)));
    printf(";BF=%d,%d", lflank, rflank);
    printf("\n");
    free(taildistances);
}

/* ---- -o detailed: print_det_output and friends (src/variant.c:313-675) ---- */

typedef struct { int op, len, start, end; const char* seq; } segview;

/* the readseg list of a seglist: start/end per new_readseg (src/readaln.c:24-99), bases sliced
 * from the read ('-' runs for deletions are implied) */
static int seg_views(const seglist* a, int from, int to, segview* out)
{
    int refpos = a->ref_start, readpos = 0, n = 0;
    for (int i = 0; i < a->n; i++) {
        const int op = CIG_OP(a->ops[i]), len = CIG_LEN(a->ops[i]);
        const int start = refpos;
        if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) refpos += len;
        if (i >= from && i < to) { out[n].op = op; out[n].len = len; out[n].start = start; out[n].end = refpos; out[n].seq = a->bases + readpos; n++; }
        if (op != OP_D) readpos += len;
    }
    return n;
}

static void det_print_left(const segview* v, int n, int lpos)
{
    /* segments of aln1 from the first one that reaches lpos (src/variant.c:549-591) */
    int k = 0;
    while (k < n && v[k].end < lpos) k++;
    forceassert(k < n);
    int rstart = v[k].start > lpos ? 0 : lpos - v[k].start;
    if (v[k].start > lpos) for (int i = lpos; i < v[k].start; i++) printf(" ");
    for (; k < n; k++) {
        switch (v[k].op) {
        case OP_EQ: case OP_X: case OP_M:
            if (rstart < v[k].len) printf("%.*s", v[k].len - rstart, v[k].seq + rstart);
            break;
        case OP_I: break;
        case OP_D: for (int i = rstart; i < v[k].len; i++) printf("-"); break;
        case OP_S: break;
        default: fatalf("Unknown CIGAR operation: %d", v[k].op);
        }
        rstart = 0;
    }
}

static void det_print_right(const segview* v, int n, int idx3, int idx4)
{
    int i = idx3;
    for (int k = 0; k < n && i < idx4; k++) {
        switch (v[k].op) {
        case OP_EQ: case OP_X: case OP_M:
            for (int j = 0; j < v[k].len && i < idx4; j++, i++) printf("%c", v[k].seq[j]);
            break;
        case OP_I: break;
        case OP_D: for (int j = 0; j < v[k].len && i < idx4; j++, i++) printf("-"); break;
        case OP_S: break;
        default: fatalf("Unknown CIGAR operation: %d", v[k].op);
        }
    }
    for (; i < idx4; i++) printf(" ");
}

static void print_deletion_output(driver* d, const variant_t* v)
{
    const char* ref = d->sequences[v->tid];
    const uint32_t sequencelen = (uint32_t)d->hdr->target_len[v->tid];
    const uint32_t neighborhood = 80;
    int i, idx2, idx3, idx4, lpos, rpos;
    char buffer[64];
    lpos = v->start < neighborhood ? 0 : (int)(v->start - neighborhood);
    for (i = lpos, idx2 = 0; i < (int)v->start; i++, idx2++) printf("%c", toupper(ref[i]));
    if ((v->stop - v->start) < 10) {
        for (idx3 = idx2; i < (int)v->stop; i++, idx3++) printf("%c", tolower(ref[i]));
    } else {
        for (idx3 = idx2; i < (int)(v->start + 5); i++, idx3++) printf("%c", tolower(ref[i]));
        if ((v->stop - v->start - 10) > 0) {
            printf("<%d>", (int)(v->stop - v->start - 10));
            sprintf(buffer, "<%d>", (int)(v->stop - v->start - 10));
            idx3 += (int)strlen(buffer);
        }
        for (i = (int)v->stop - 5; i < (int)v->stop; i++, idx3++) printf("%c", tolower(ref[i]));
    }
    rpos = (v->stop + neighborhood) > sequencelen ? (int)sequencelen : (int)(v->stop + neighborhood);
    for (idx4 = idx3; i < rpos; i++, idx4++) printf("%c", toupper(ref[i]));
    printf("\n");
    for (uint32_t s = 0; s < v->support; s++) {
        const evidence_t* e = v->evidence[s];
        if (e->type == EV_PAIRED_READ) { printf("%s\n", e->qname); continue; }
        segview* sv = xmalloc(sizeof(segview) * (size_t)(e->aln.n + 1));
        int n1 = seg_views(&e->aln, 0, e->seg, sv);
        det_print_left(sv, n1, lpos);
        if (CIG_OP(e->aln.ops[e->seg]) != OP_D) fatalf("This segment should only contain the variation");
        for (i = idx2; i < idx3; i++) printf("-");
        int n3 = seg_views(&e->aln, e->seg + 1, e->aln.n, sv);
        det_print_right(sv, n3, idx3, idx4);
        printf("%s\n", e->qname);
        free(sv);
    }
}

static void print_insertion_output(driver* d, const variant_t* v)
{
    int maxinsertsize = 0;
    char* consensus = voted_consensus(v->evidence, v->support, &maxinsertsize);
    free(consensus);
    const char* ref = d->sequences[v->tid];
    const uint32_t sequencelen = (uint32_t)d->hdr->target_len[v->tid];
    const uint32_t neighborhood = 80;
    int i, j, idx2, idx3, idx4, lpos, rpos;
    lpos = v->start < neighborhood ? 0 : (int)(v->start - neighborhood);
    for (i = lpos, idx2 = 0; i < (int)v->start; i++, idx2++) printf("%c", toupper(ref[i]));
    for (idx3 = idx2, j = 0; j < maxinsertsize; j++, idx3++) printf("-");
    rpos = (v->stop + neighborhood) > sequencelen ? (int)sequencelen : (int)(v->stop + neighborhood);
    for (idx4 = idx3; i < rpos; i++, idx4++) printf("%c", toupper(ref[i]));
    printf("\n");
    for (uint32_t s = 0; s < v->support; s++) {
        const evidence_t* e = v->evidence[s];
        segview* sv = xmalloc(sizeof(segview) * (size_t)(e->aln.n + 1));
        int n1 = seg_views(&e->aln, 0, e->seg, sv);
        int k = 0;
        while (k < n1 && sv[k].end < lpos) k++;
        if (k == n1) { free(sv); continue; }          /* src/variant.c:360-361 */
        det_print_left(sv, n1, lpos);
        segview one;
        seg_views(&e->aln, e->seg, e->seg + 1, &one);
        if (one.op != OP_I) fatalf("This segment should only contain the variation");
        for (j = 0; j < one.len; j++) printf("%c", tolower(one.seq[j]));
        for (; j < maxinsertsize; j++) printf("-");
        int n3 = seg_views(&e->aln, e->seg + 1, e->aln.n, sv);
        det_print_right(sv, n3, idx3, idx4);
        printf("%s\n", e->qname);
        free(sv);
    }
}

static void print_det_output(driver* d, const variant_t* v)
{
    static int indel_index = 1;
    printf("###########################################################\n");
    /* the blocks are numbered across the run: a rank of a multi-GPU run does not know how many the contigs in front of its own
     * print, so it leaves a mark where the number goes and rank 0 counts while it puts the parts together (mg_finish) */
    if (g_mg_parts) printf("\001"); else printf("%d", indel_index++);
    printf("\t%s\t%d\t%d\t%s\t%d\t%d\t%d\n", d->hdr->target_name[v->tid], (int)v->start, (int)v->stop,
           v->type == CLS_DELETION ? "Deletion" : "Insertion", (int)v->start, (int)(v->stop + v->rw + 1), (int)v->support);
    if (v->type == CLS_DELETION) print_deletion_output(d, v);
    else if (v->type == CLS_INSERTION) print_insertion_output(d, v);
}

static void emit_variant(driver* d, const variant_t* v)
{
    if (strcmp(O.outputformat, "vcf") == 0) print_vcf_output(d, v);
    else if (strcmp(O.outputformat, "detailed") == 0) print_det_output(d, v);
}

/* print_variants (src/variant.c:678-838) */
static void print_variants(driver* d, variant_list* vs)
{
    variant_list sel = {0};
    for (int x = 0; x < vs->n; x++) {
        variant_t* it = vs->v[x];
        int left = 0, right = 0, balance = INT_MAX, lflank = -1, rflank = -1;
        uint32_t numdiffs = 0;
        for (uint32_t i = 0; i < it->support; i++) {
            const evidence_t* e = it->evidence[i];
            const uint32_t ltmp = (uint32_t)e->lflank, rtmp = (uint32_t)e->rflank;
            numdiffs += (uint32_t)e->nd_filter;
            if (ltmp >= O.minbalance) left = 1;
            if (rtmp >= O.minbalance) right = 1;
            if (abs((int)(rtmp - ltmp)) < balance) { balance = abs((int)(rtmp - ltmp)); lflank = (int)ltmp; rflank = (int)rtmp; }
        }
        const uint32_t xnumdiffs = (uint32_t)((int)(numdiffs * 1.0 / it->support) + 0.5);
        const int ok_flanks = (it->type == CLS_DELETION && (uint32_t)lflank >= O.minbalance && (uint32_t)rflank >= O.minbalance) ||
                              (it->type == CLS_INSERTION && ((uint32_t)lflank >= O.minbalance || (uint32_t)rflank >= O.minbalance));
        if (ok_flanks && it->support >= O.minsupport && xnumdiffs <= O.maxdiffsallowed && left && right) vl_push(&sel, it);
    }
    /* who gets printed, in print order ... */
    variant_list out = {0};
    if (O.call_all_indels) {
        for (int i = 0; i < sel.n; i++) vl_push(&out, sel.v[i]);
    } else {
        int i = 0;
        while (i < sel.n) {
            variant_t* it = sel.v[i];
            int j = i + 1;
            while (j < sel.n && (sel.v[j]->start - sel.v[j]->lw) <= (it->stop + it->rw)) j++;
            uint32_t maxsupport = 0;
            variant_t* chosen = it;
            for (int t = i; t < j; t++) if (sel.v[t]->support > maxsupport) { maxsupport = sel.v[t]->support; chosen = sel.v[t]; }
            vl_push(&out, chosen);
            i = j;
        }
    }
    /* ... their DP= values in ONE device query instead of one launch + copy + wait per variant
     * (calculate_cov_params is called per printed variant, src/variant.c:303-306) ... */
    if (strcmp(O.outputformat, "vcf") == 0 && out.n > 0) {
        int32_t* beg = xmalloc(sizeof(int32_t) * (size_t)out.n);
        int32_t* end = xmalloc(sizeof(int32_t) * (size_t)out.n);
        uint32_t* sum = xmalloc(sizeof(uint32_t) * (size_t)out.n);
        uint32_t* deepest = xmalloc(sizeof(uint32_t) * (size_t)out.n);
        int* who = xmalloc(sizeof(int) * (size_t)out.n);
        int m = 0;
        for (int i = 0; i < out.n; i++) {
            variant_t* v = out.v[i];
            const int32_t start = (int32_t)(v->start - v->lw - 1), stop = (int32_t)(v->stop + v->rw + 1);
            v->dp_valid = 0;
            if (d->depth_tid != v->tid) continue;          /* region runs go to the BAM (region_depth) */
            if (stop <= start) { v->dp_cached = 0; v->dp_valid = 1; continue; }
            beg[m] = start; end[m] = stop; who[m] = i; m++;
        }
        if (m > 0) {
            gpu_wait(d);
            pthread_mutex_lock(&g_query_mu);
            for (int q = 0; q < m; q++) deepest[q] = 0;
            const int qrc = d->pipe_mode ? im_depth_query_max_tid(d->gpu, d->depth_tid, m, beg, end, sum, deepest) : im_depth_query(d->gpu, m, beg, end, sum);
            pthread_mutex_unlock(&g_query_mu);
            if (qrc != IM_OK)
                fatalf("im_depth_query: %s", im_last_error(d->gpu));
            for (int q = 0; q < m; q++) {
                variant_t* v = out.v[who[q]];
                v->dp_cached = deepest[q] >= DP_DEEP_LOCUS ? (int32_t)region_depth_from_bam(d, d->depth_tid, beg[q], end[q])      /* the pileup's cap may have applied */
                                                           : (int32_t)(uint32_t)floor(sum[q] * 1.0 / (uint32_t)(end[q] - beg[q]));
                v->dp_valid = 1;
            }
        }
        free(beg); free(end); free(sum); free(deepest); free(who);
    }
    /* ... and out they go */
    for (int i = 0; i < out.n; i++) emit_variant(d, out.v[i]);
    free(out.v);
    free(sel.v);
}

/* --------------------------------------------------------- process_evidence -- */

typedef struct { int32_t b1, b2; int64_t arrival; int64_t idx; } skey;
static int g_tie_desc;
static int cmp_skey(const void* x, const void* y)
{
    const skey* a = x; const skey* b = y;
    if (a->b1 != b->b1) return a->b1 < b->b1 ? -1 : 1;
    if (a->b2 != b->b2) return a->b2 < b->b2 ? -1 : 1;
    /* prepend list + stable merge sort: ties newest first; the expected.vcf
     * order is the opposite */
    if (a->arrival == b->arrival) return 0;
    if (g_tie_desc) return a->arrival < b->arrival ? -1 : 1;
    return a->arrival > b->arrival ? -1 : 1;
}

static int uf_find(int* p, int x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

/* process_evidence (src/indelminer.c:117-209): consumes the pending evidence list, returns the
 * variants sorted by position (sort_variants applied). */
static void process_evidence(driver* d, int32_t tid, int marker, variant_list* out)
{
    const int64_t n = d->n_pending;
    out->n = 0;
    if (n == 0) return;
    skey* keys = xmalloc(sizeof(skey) * (size_t)n);
    for (int64_t i = 0; i < n; i++) { keys[i].b1 = d->pending[i]->b1; keys[i].b2 = d->pending[i]->b2; keys[i].arrival = d->pending[i]->arrival; keys[i].idx = i; }
    g_tie_desc = O.tie_desc;
    qsort(keys, (size_t)n, sizeof(skey), cmp_skey);
    int64_t m = 0;
    while (m < n && keys[m].b2 < marker) m++;           /* nodes for the sorted prefix (137-146) */

    /* split-read nodes: the GPU groups them (identical class,b1,b2; src/graph.c:122-127) */
    int32_t nsr = 0, npe = 0;
    int64_t* sr_idx = xmalloc(sizeof(int64_t) * (size_t)(m ? m : 1));
    int64_t* pe_pos = xmalloc(sizeof(int64_t) * (size_t)(m ? m : 1));    /* sorted positions of PE nodes */
    uint8_t* is_node = xcalloc((size_t)n, 1);
    int64_t* sorted_pos = xmalloc(sizeof(int64_t) * (size_t)n);
    for (int64_t s = 0; s < m; s++) { is_node[keys[s].idx] = 1; sorted_pos[keys[s].idx] = s; if (d->pending[keys[s].idx]->type == EV_PAIRED_READ) pe_pos[npe++] = s; }
    for (int64_t i = 0; i < n; i++) if (is_node[i] && d->pending[i]->type == EV_SPLIT_READ) sr_idx[nsr++] = i;   /* arrival order */

    variant_list vars = {0};
    if (nsr > 0) {
        int32_t *cls = xmalloc(4 * (size_t)nsr), *b1 = xmalloc(4 * (size_t)nsr), *b2 = xmalloc(4 * (size_t)nsr);
        int32_t *order = xmalloc(4 * (size_t)nsr), *first = xmalloc(4 * (size_t)nsr), *count = xmalloc(4 * (size_t)nsr);
        uint8_t* used = xmalloc((size_t)nsr);
        for (int32_t i = 0; i < nsr; i++) { const evidence_t* e = d->pending[sr_idx[i]]; cls[i] = e->cls; b1[i] = e->b1; b2[i] = e->b2; }
        int32_t ncl = 0;
        gpu_wait(d);
        const int rc = im_cluster_sr(d->gpu, nsr, cls, b1, b2, INT_MAX, O.tie_desc, order, first, count, used, &ncl);
        if (rc != IM_OK) fatalf("im_cluster_sr: %s", im_last_error(d->gpu));
        for (int32_t c = 0; c < ncl; c++) {
            variant_t* v = xcalloc(1, sizeof *v);
            const evidence_t* e0 = d->pending[sr_idx[order[first[c]]]];
            v->type = e0->cls; v->evdnctype = EV_SPLIT_READ; v->tid = tid;
            v->start = (uint32_t)e0->b1; v->stop = (uint32_t)e0->b2; v->support = (uint32_t)count[c];
            v->evidence = xmalloc(sizeof(evidence_t*) * (size_t)count[c]);
            int64_t rep = -1;
            for (int32_t k = 0; k < count[c]; k++) {
                const int64_t pi = sr_idx[order[first[c] + k]];
                v->evidence[k] = d->pending[pi];
                if (sorted_pos[pi] > rep) rep = sorted_pos[pi];
            }
            v->rep_arrival = rep;       /* largest sorted position of a member: decides the component id order */
            if (v->start <= v->stop) vl_push(&vars, v); else variant_free(v);
        }
        free(cls); free(b1); free(b2); free(order); free(first); free(count); free(used);
    }
    if (npe > 0) {
        /* paired-read nodes: add_node's O(N^2) rule (src/graph.c:100-121), union-find for the components */
        int* parent = xmalloc(sizeof(int) * (size_t)npe);
        for (int i = 0; i < npe; i++) parent[i] = i;
        /* the partners of e1 lie within the largest insert-length bound below it in (b1)-sorted order (see group_process_flush) */
        int32_t widest = 0;
        for (int j = 0; j < npe; j++) if (d->pending[keys[pe_pos[j]].idx]->max > widest) widest = d->pending[keys[pe_pos[j]].idx]->max;
        for (int j = 0; j < npe; j++) {
            const evidence_t* e1 = d->pending[keys[pe_pos[j]].idx];
            for (int i = j - 1; i >= 0; i--) {
                const evidence_t* e2 = d->pending[keys[pe_pos[i]].idx];
                forceassert(e2->b1 <= e1->b1);
                if (e1->b1 - e2->b1 >= widest) break;
                if (e2->b1 < e1->b2 && e1->cls == e2->cls) {
                    const int32_t bb1 = e1->b1 > e2->b1 ? e1->b1 : e2->b1;
                    const int32_t bb2 = e1->b2 < e2->b2 ? e1->b2 : e2->b2;
                    const int32_t d1 = bb1 - seglist_first_start(&e1->aln) + seglist_last_end(&e1->aln3) - bb2;
                    const int32_t d2 = bb1 - seglist_first_start(&e2->aln) + seglist_last_end(&e2->aln3) - bb2;
                    if (d1 < e1->max && d2 < e2->max) { int a = uf_find(parent, i), c = uf_find(parent, j); if (a != c) parent[a] = c; }
                }
            }
        }
        /* components; members in descending sorted position (node list is prepend order) */
        uint8_t* done = xcalloc((size_t)npe, 1);
        for (int j = npe - 1; j >= 0; j--) {
            if (done[j]) continue;
            const int root = uf_find(parent, j);
            variant_t* v = xcalloc(1, sizeof *v);
            v->evidence = xmalloc(sizeof(evidence_t*) * (size_t)npe);
            int left = -1, right = -1;
            for (int t = j; t >= 0; t--) {
                if (done[t] || uf_find(parent, t) != root) continue;
                done[t] = 1;
                evidence_t* e = d->pending[keys[pe_pos[t]].idx];
                v->evidence[v->support++] = e;
                if (left == -1 || e->b1 > left) left = e->b1;
                if (right == -1 || e->b2 < right) right = e->b2;
            }
            const evidence_t* e0 = v->evidence[0];
            v->type = e0->cls; v->evdnctype = e0->type; v->tid = tid;
            v->start = (uint32_t)left; v->stop = (uint32_t)right;
            v->rep_arrival = pe_pos[j];
            if (v->start <= v->stop) vl_push(&vars, v); else variant_free(v);
        }
        free(parent); free(done);
    }
    /* components are numbered from the largest sorted position down, the variant list is built by
     * prepending, and sort_variants is stable: equal (start,stop) come out in ascending order of
     * the component's largest sorted position */
    for (int i = 1; i < vars.n; i++) {
        variant_t* v = vars.v[i]; int j = i - 1;
        while (j >= 0 && vars.v[j]->rep_arrival > v->rep_arrival) { vars.v[j + 1] = vars.v[j]; j--; }
        vars.v[j + 1] = v;
    }
    sort_variants(&vars);

    /* every node is used up, whether or not its component made a variant (189,199) */
    for (int64_t i = 0; i < n; i++) if (is_node[i]) d->pending[i]->used = 1;
    free(keys); free(sr_idx); free(pe_pos); free(is_node); free(sorted_pos);
    *out = vars;
}

static void free_used_evidence(driver* d)
{
    int64_t k = 0;
    for (int64_t i = 0; i < d->n_pending; i++) {
        if (d->pending[i]->used) evidence_free(d->pending[i]);
        else d->pending[k++] = d->pending[i];
    }
    d->n_pending = k;
}

static void pending_push(driver* d, evidence_t* e)
{
    if (d->n_pending == d->cap_pending) {
        d->cap_pending = d->cap_pending ? d->cap_pending * 2 : 4096;
        d->pending = xrealloc(d->pending, sizeof(evidence_t*) * (size_t)d->cap_pending);
    }
    e->arrival = d->arrival++;
    d->pending[d->n_pending++] = e;
}

/* ------------------------------------------------------------ annotate mode -- */

typedef struct {
    int32_t  tid;
    uint32_t start;                 /* VCF POS */
    char*    reference;
    char*    alternate;
    int      type, evdnctype;
    uint32_t support, stop, bpstop;
    char*    addntlinfo;
    int      diffsample_support;
} knownvariant_t;

typedef struct { knownvariant_t** v; int n, cap; int next; } known_list;

static const char* g_vcfname = NULL;
static const char* g_sample_name = NULL;

/* read_variants (src/variant.c:841-921): the whole VCF is parsed again for every contig */
static void read_variants(const char* vcfname, int32_t tid, const char* chromname, known_list* out)
{
    size_t cap = 2;
    char* line = xmalloc(cap);
    out->n = 0; out->next = 0;
    FILE* fp = fopen(vcfname, "r");
    if (!fp) fatalf("error in opening the file %s", vcfname);
    const size_t big = (size_t)O.maxpedelsize + 16;
    char* reference = xmalloc(big);
    char* alternate = xmalloc(big);
    int numread = 0;
    while (im_getline(&line, &cap, fp) != -1) {
        if (line[0] == '#') continue;
        char chrom[128], type[128], evd[128], sup[128], stp[128], bps[128], info[1024];
        unsigned start;
        if (sscanf(line, "%127s %u %*c %s %s %*c %*c %127[^;];%127[^;];NS=%127[^;];END=%127[^;];BP_END=%127[^;];%1023s\n",
                   chrom, &start, reference, alternate, type, evd, sup, stp, bps, info) != 10)
            fatalf("Error in reading the variant : %s", line);
        if (strcmp(chrom, chromname) != 0) continue;
        numread++;
        knownvariant_t* k = xcalloc(1, sizeof *k);
        k->tid = tid; k->start = start;
        k->addntlinfo = xstrdup(info);
        k->reference = xstrdup(reference); k->alternate = xstrdup(alternate);
        k->type = strncmp(type, "DELETION", 8) == 0 ? CLS_DELETION : CLS_INSERTION;
        if (strcmp(evd, "SPLIT_READ") == 0) k->evdnctype = EV_SPLIT_READ;
        else if (strcmp(evd, "PAIRED_READ") == 0) k->evdnctype = EV_PAIRED_READ;
        else if (strcmp(evd, "COMPOSITE") == 0) k->evdnctype = EV_COMPOSITE;
        else fatalf("unknown evidence type for this variant");
        k->support = (uint32_t)atoi(sup); k->stop = (uint32_t)atoi(stp); k->bpstop = (uint32_t)atoi(bps);
        if (out->n == out->cap) { out->cap = out->cap ? out->cap * 2 : 64; out->v = xrealloc(out->v, sizeof(knownvariant_t*) * (size_t)out->cap); }
        out->v[out->n++] = k;
    }
    fclose(fp);
    free(line); free(reference); free(alternate);
    /* list built by prepending, then the stable sort_by_knownposition (27-37,915): ties come out in
     * reverse file order */
    for (int i = 0; i < out->n / 2; i++) { knownvariant_t* t = out->v[i]; out->v[i] = out->v[out->n - 1 - i]; out->v[out->n - 1 - i] = t; }
    for (int i = 1; i < out->n; i++) {
        knownvariant_t* k = out->v[i]; int j = i - 1;
        while (j >= 0 && (out->v[j]->start > k->start || (out->v[j]->start == k->start && (int)out->v[j]->bpstop - (int)k->bpstop > 0))) { out->v[j + 1] = out->v[j]; j--; }
        out->v[j + 1] = k;
    }
    fprintf(stderr, "Read %d variants for %s\n", numread, chromname);
}

static void known_free(known_list* l)
{
    for (int i = 0; i < l->n; i++) { free(l->v[i]->reference); free(l->v[i]->alternate); free(l->v[i]->addntlinfo); free(l->v[i]); }
    l->n = 0; l->next = 0;
}

static void print_vcf_line(const driver* d, const knownvariant_t* k)
{
    /* src/variant.c:1227-1244 */
    printf("%s\t%d\t.\t%s\t%s\t.\t.\t%s;", d->hdr->target_name[k->tid], (int)k->start, k->reference, k->alternate,
           k->type == CLS_DELETION ? "DELETION" : "INSERTION");
    if (k->evdnctype == EV_SPLIT_READ) printf("SPLIT_READ;");
    else if (k->evdnctype == EV_PAIRED_READ) printf("PAIRED_READ;");
    else if (k->evdnctype == EV_COMPOSITE) printf("COMPOSITE;");
    printf("NS=%d;END=%d;BP_END=%d;%s", (int)k->support, (int)k->stop, (int)k->bpstop, k->addntlinfo);
}

/* is_indel_supported (src/variant.c:1561-1573) = check_for_indel (1427-1556) over the reads that
 * overlap [start, stop).  The CIGAR bookkeeping is per read on the host; the Smith-Waterman of
 * every read that needs one goes to the GPU as one im_support_batch. */
static int is_indel_supported(driver* d, knownvariant_t* k)
{
    const char* seq = d->sequences[k->tid];
    const int64_t seqlen = d->seqlen[k->tid];
    bgzf_reader* r = bgzf_open(d->bam_name);
    if (!r) fatalf("error in opening the file %s", d->bam_name);
    bam_header* h = bam_header_load(r);
    bam_region_iter it;
    bam_record b; memset(&b, 0, sizeof b);
    /* SW tasks in read order with the read's own counts */
    uint8_t* tg = NULL; size_t tg_len = 0, tg_cap = 0;
    uint8_t* qs = NULL; size_t qs_len = 0, qs_cap = 0;
    int64_t *to = NULL, *qo = NULL; int32_t* own = NULL; int nt = 0, capt = 0;
    if (h && bam_region_begin(&it, r, d->idx, k->tid, (int32_t)k->start, (int32_t)k->stop) == 0) {
        while (!k->diffsample_support && bam_region_next(&it, &b) == 1) {
            if (b.flag & 0x4) continue;
            if (b.flag & (0x100 | 0x200 | 0x400 | 0x800)) continue;
            seglist rln = seglist_from_record(&b);
            int aln1subs = 0, aln1indels = 0, aln1aligned = 0, overlaps = 0, qstart = -1, qstop = -1, readindx = 0;
            int refpos = rln.ref_start, done = 0;
            for (int sgi = 0; sgi < rln.n && !done; sgi++) {
                const int op = CIG_OP(rln.ops[sgi]), len = CIG_LEN(rln.ops[sgi]);
                const int sstart = refpos;
                const int send = (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D) ? refpos + len : refpos;
                if (!(send < (int)k->start || sstart > (int)k->stop)) overlaps = 1;
                switch (op) {
                case OP_S:
                    if (sgi == rln.n - 1) qstop = readindx;
                    readindx += len;
                    break;
                case OP_I:
                    if (qstart == -1) qstart = readindx;
                    if (k->type == CLS_INSERTION && sstart == (int)k->start) { k->diffsample_support = 1; done = 1; break; }
                    readindx += len; aln1indels += len; aln1aligned += len;
                    break;
                case OP_D:
                    if (k->type == CLS_DELETION && sstart == (int)k->start && send == (int)k->stop - 1) { k->diffsample_support = 1; done = 1; break; }
                    aln1indels += len;
                    break;
                case OP_M:
                    if (qstart == -1) qstart = readindx;
                    for (int i = 0, j = sstart; i < len; i++, j++) if (rln.bases[readindx + i] != seq[j]) aln1subs++;
                    readindx += len; aln1aligned += len;
                    break;
                default:
                    fatalf("Unhandled CIGAR op: %d", op);
                }
                refpos = send;
            }
            if (done) { seglist_free(&rln); break; }
            if (qstop == -1) qstop = aln1aligned + qstart;
            forceassert(aln1aligned == (qstop - qstart));
            if (!overlaps) { seglist_free(&rln); continue; }
            const int indelsize = abs((int)strlen(k->alternate) - (int)strlen(k->reference));
            int rstart = b.pos, rstop = bam_record_end(&b);
            if ((uint32_t)rstop < k->bpstop) { seglist_free(&rln); continue; }
            forceassert(qstart != -1 && qstop != -1);
            rstart -= indelsize; rstop += indelsize;
            /* the fake reference with the variant in it (1259-1272); reads beyond the contig's ends
             * stop at its terminator there, here they are clipped */
            if (rstart < 0) rstart = 0;
            if (rstop > seqlen) rstop = (int)seqlen;
            const size_t alen = strlen(k->alternate);
            size_t need = (size_t)(rstop - rstart) + alen + 8;
            if (tg_len + need > tg_cap) { tg_cap = (tg_cap + need) * 2; tg = xrealloc(tg, tg_cap); }
            uint8_t* t = tg + tg_len;
            size_t tl = 0;
            if (k->type == CLS_DELETION) {
                /* ref[rstart, start) + ref[stop-1, rstop) */
                const int a_end = (int)k->start < rstop ? (int)k->start : rstop;
                if (a_end > rstart) { memcpy(t, seq + rstart, (size_t)(a_end - rstart)); tl = (size_t)(a_end - rstart); }
                const int b_beg = (int)k->stop - 1;
                if (rstop > b_beg && b_beg >= 0) { memcpy(t + tl, seq + b_beg, (size_t)(rstop - b_beg)); tl += (size_t)(rstop - b_beg); }
            } else {
                /* ref[rstart, start) + alternate[1..] + ref[start, rstop) */
                const int a_end = (int)k->start < rstop ? (int)k->start : rstop;
                if (a_end > rstart) { memcpy(t, seq + rstart, (size_t)(a_end - rstart)); tl = (size_t)(a_end - rstart); }
                if (alen > 1) { memcpy(t + tl, k->alternate + 1, alen - 1); tl += alen - 1; }
                if (rstop > a_end) { memcpy(t + tl, seq + a_end, (size_t)(rstop - a_end)); tl += (size_t)(rstop - a_end); }
            }
            /* query = read[qstart, qstop) of the record's stored bases */
            int qlen = qstop - qstart;
            if ((int)strlen(rln.bases + qstart) < qlen) qlen = (int)strlen(rln.bases + qstart);
            if (qs_len + (size_t)qlen + 8 > qs_cap) { qs_cap = (qs_cap + (size_t)qlen + 8) * 2; qs = xrealloc(qs, qs_cap); }
            memcpy(qs + qs_len, rln.bases + qstart, (size_t)qlen);
            if (nt + 2 > capt) { capt = capt ? capt * 2 : 64; to = xrealloc(to, sizeof(int64_t) * (size_t)(capt + 1)); qo = xrealloc(qo, sizeof(int64_t) * (size_t)(capt + 1)); own = xrealloc(own, sizeof(int32_t) * 3 * (size_t)capt); }
            to[nt] = (int64_t)tg_len; qo[nt] = (int64_t)qs_len;
            own[3 * nt] = aln1subs; own[3 * nt + 1] = aln1indels; own[3 * nt + 2] = aln1aligned;
            tg_len += tl; qs_len += (size_t)qlen; nt++;
            to[nt] = (int64_t)tg_len; qo[nt] = (int64_t)qs_len;
            seglist_free(&rln);
        }
    }
    free(b.data);
    bam_header_free(h);
    bgzf_close(r);
    if (!k->diffsample_support && nt > 0) {
        int32_t* res = xmalloc(sizeof(int32_t) * 4 * (size_t)nt);
        gpu_wait(d);
        const int64_t t0sw = wall_ns();
        if (im_support_batch(d->gpu, nt, tg, to, qs, qo, res) != IM_OK) fatalf("im_support_batch: %s", im_last_error(d->gpu));
        if (g_timing) {         /* annotate mode's Smith-Waterman work (src/variant.c:1288-1424: (query + 1) x (target + 1) cells per task) */
            g_sw_ns += wall_ns() - t0sw; g_sw_calls++; g_sw_tasks += nt;
            for (int i = 0; i < nt; i++) g_sw_cells += (to[i + 1] - to[i] + 1) * (qo[i + 1] - qo[i] + 1);
        }
        for (int i = 0; i < nt; i++)
            if (res[4 * i] <= own[3 * i] && res[4 * i + 1] <= own[3 * i + 1] && res[4 * i + 2] >= own[3 * i + 2]) { k->diffsample_support = 1; break; }
        free(res);
    }
    free(tg); free(qs); free(to); free(qo); free(own);
    return k->diffsample_support;
}

/* print_knownvariants (src/variant.c:1577-1692): known variants from kl->next on; stops at the
 * first known variant that lies behind the last discovered one (1661-1666) */
static void print_knownvariants(driver* d, known_list* kl, const variant_list* vars)
{
    if (vars->n == 0) return;
    int ki = kl->next;
    for (; ki < kl->n; ki++) {
        knownvariant_t* k = kl->v[ki];
        int is_found = 0;
        const uint32_t kstart = k->start, kstop = k->stop;
        int ui;
        for (ui = 0; ui < vars->n; ui++) {
            const variant_t* u = vars->v[ui];
            const uint32_t ustart = u->start - u->lw;
            uint32_t ustop = 0;
            const int reflength = (int)(u->stop + u->rw) - (int)(u->start - u->lw - 1);
            forceassert(reflength >= 1);
            if (u->type == CLS_DELETION) {
                const int altlength = (int)(u->start + u->rw) - (int)(u->start - u->lw - 1);
                forceassert(altlength >= 1);
                ustop = u->start - u->lw + (uint32_t)reflength - (uint32_t)altlength + 1;
            } else if (u->type == CLS_INSERTION) ustop = ustart + 1;
            forceassert(ustop != 0);
            if (kstart >= ustop) { }
            else if (ustart >= kstop) { }
            else {
                if ((k->evdnctype == EV_SPLIT_READ || k->evdnctype == EV_COMPOSITE) && u->evdnctype == EV_SPLIT_READ) {
                    if (kstart == ustart && kstop == ustop) { is_found = 1; break; }
                } else if (((k->evdnctype == EV_SPLIT_READ || k->evdnctype == EV_COMPOSITE) && u->evdnctype == EV_PAIRED_READ) ||
                           (k->evdnctype == EV_PAIRED_READ && u->evdnctype == EV_SPLIT_READ) ||
                           (k->evdnctype == EV_PAIRED_READ && u->evdnctype == EV_PAIRED_READ)) {
                    const uint32_t sx = k->start > u->start ? k->start : u->start;
                    const uint32_t ex = k->bpstop < u->stop ? k->bpstop : u->stop;
                    uint32_t olap = 0;
                    if (ex >= sx) olap = ex - sx;
                    if ((olap * 100.00 / (k->bpstop - k->start)) > 50) { is_found = 1; break; }
                }
            }
        }
        if (ui == vars->n) {
            const variant_t* last = vars->v[vars->n - 1];
            if (last->start < kstart) break;
        }
        print_vcf_line(d, k);
        if (is_found) printf(";%s", g_sample_name);
        else if (k->evdnctype == EV_SPLIT_READ && is_indel_supported(d, k)) printf(";%s", g_sample_name);
        printf("\n");
    }
    kl->next = ki;
}

static known_list g_known;

static void flush_variants(driver* d, int32_t tid, int marker)
{
    variant_list vs = {0};
    process_evidence(d, tid, marker, &vs);
    if (g_vcfname == NULL) {
        merge_variants(&vs, d->sequences[tid], d->seqlen[tid], 1);
        print_variants(d, &vs);
    } else {
        /* annotate mode (src/indelminer.c:647-661, 824-855): SR and PE variants stay apart */
        merge_variants(&vs, d->sequences[tid], d->seqlen[tid], 0);
        print_knownvariants(d, &g_known, &vs);
    }
    fflush(stdout);
    free_used_evidence(d);
    for (int i = 0; i < vs.n; i++) variant_free(vs.v[i]);
    free(vs.v);
}
