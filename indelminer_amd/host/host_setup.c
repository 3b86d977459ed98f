/* host_setup.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: start-up: the GPU context thread, config file / insert-length estimation, the mean-coverage table, the VCF preamble, and
 * the record-at-a-time path (pass A / GPU / pass B per contig) that aborted runs are handed to. */

/* ------------------------------------------------------ GPU start-up ------- */

static void* gpu_open_thread(void* arg)
{
    driver* d = (driver*)arg;
    const char* dev_env = getenv("INDELMINER_DEVICE");
    d->gpu_rc = im_ctx_create(dev_env ? atoi(dev_env) : (g_mg_local >= 0 ? g_mg_local : 0), &d->gpu);
    if (d->gpu_rc != IM_OK) snprintf(d->gpu_err, sizeof d->gpu_err, "cannot open the GPU: %s", im_last_error(NULL));
    pthread_mutex_lock(&d->gpu_mu);
    d->ctx_rc = d->gpu_rc;
    d->ctx_ready = 1;                                   /* the walkers' buffers can be set up from here on */
    pthread_cond_broadcast(&d->gpu_cv);
    pthread_mutex_unlock(&d->gpu_mu);
    if (d->gpu_rc != IM_OK) return NULL;
    pthread_mutex_lock(&d->gpu_mu);
    while (!d->seq_ready) pthread_cond_wait(&d->gpu_cv, &d->gpu_mu);
    pthread_mutex_unlock(&d->gpu_mu);
    const char** seqs = xcalloc((size_t)d->hdr->n_targets, sizeof(char*));
    int64_t* lens = xcalloc((size_t)d->hdr->n_targets, sizeof(int64_t));
    for (int32_t i = 0; i < d->hdr->n_targets; i++) { seqs[i] = d->sequences[i] ? d->sequences[i] : ""; lens[i] = d->sequences[i] ? d->seqlen[i] : 0; }
    d->gpu_rc = im_set_reference(d->gpu, d->hdr->n_targets, seqs, lens);
    if (d->gpu_rc != IM_OK) snprintf(d->gpu_err, sizeof d->gpu_err, "im_set_reference: %s", im_last_error(d->gpu));
    free(seqs); free(lens);
    return NULL;
}

/* the output header (src/indelminer.c:745-754) */
static void print_output_header(void)
{
    if (strncmp(O.outputformat, "vcf", 3) == 0) print_vcf_preamble();
    if (g_vcfname != NULL)
        printf("##INFO=<ID=%s,Number=0,Type=Flag,Description=\"The variant is also present in this sample\">\n", g_sample_name);
    if (strncmp(O.outputformat, "vcf", 3) == 0) printf("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    fflush(OUT);
}
/* One pass without a config file: the reference prints its header after its estimation pass and the FASTA read -- a run it ends
 * inside that pass (an RG tag that is not a string) has printed nothing.  Here that pass IS the walk: the header waits until the
 * insert-length table has been made (run_pipeline). */
static int g_header_held;

/* the output header (src/indelminer.c:745-754); multi-GPU: rank 0 prints it, as the first part */
static void header_out(void)
{
    if (g_mg_rank > 0) return;
    if (g_mg_header_path[0] && !freopen(g_mg_header_path, "w", stdout)) fatalf("cannot write %s", g_mg_header_path);
    print_output_header();
    fflush(stdout);
    /* the header part is complete; whatever a library prints on stdout from here on (librccl's banner) is not VCF */
    if (g_mg_header_path[0] && !freopen("/dev/stderr", "w", stdout)) { }
}

/* every GPU call site passes through here first */
static void gpu_wait(driver* d)
{
    if (!d->gpu_pending) return;
    pthread_join(d->gpu_thread, NULL);
    d->gpu_pending = 0;
    if (d->gpu_rc != IM_OK) fatalf("%s", d->gpu_err);
    phase_time("GPU context + reference upload (helper thread, joined)");
    /* the output header (src/indelminer.c:745-754) goes out only once the GPU is known to be there:
     * nothing is printed by a run that cannot compute */
    if (g_header_held) return;
    header_out();
}

/* ------------------------------------------------------ config / estimates -- */

/* the insert-length table's entries in the order they were added: the device rebuilds the chains from it */
static const char** g_rg_name; static int32_t** g_rg_range; static int g_rg_n, g_rg_cap;
static void rg_order_push(const char* name, int32_t* range)
{
    if (g_rg_n == g_rg_cap) {
        g_rg_cap = g_rg_cap ? g_rg_cap * 2 : 16;
        g_rg_name = xrealloc(g_rg_name, sizeof(char*) * (size_t)g_rg_cap);
        g_rg_range = xrealloc(g_rg_range, sizeof(int32_t*) * (size_t)g_rg_cap);
    }
    g_rg_name[g_rg_n] = xstrdup(name); g_rg_range[g_rg_n] = range; g_rg_n++;
}

static uint32_t* g_meancov;         /* [n_targets] mean coverage: the RC lines of a config file, or observed (cov_means) */

static void read_configuration(const char* filename, qhash* insertlengths, const bam_header* hdr)
{
    /* src/shared.c:5-44.  An RC line's contig goes through must_find_hashtable_int on a 32-bin table of the BAM header's names
     * (src/indelminer.c:700-706) -- a name the header does not know ends the run there; the coverage itself is only ever
     * printed on stderr (src/indelminer.c:731). */
    qhash* id2chroms = qhash_new(5);
    for (int32_t i = 0; i < hdr->n_targets; i++) qhash_add(id2chroms, hdr->target_name[i], (int)strlen(hdr->target_name[i]), (void*)(intptr_t)(i + 1));
    if (!g_meancov) g_meancov = xcalloc((size_t)(hdr->n_targets > 0 ? hdr->n_targets : 1), sizeof(uint32_t));
    size_t cap = 2;
    char* line = xmalloc(cap);
    FILE* fp = fopen(filename, "r");
    if (!fp) fatalf("error in opening the file %s", filename);
    while (im_getline(&line, &cap, fp) != -1) {
        char name[128]; unsigned a, b;
        if (strncmp(line, "IL", 2) == 0) {
            if (sscanf(line, "IL %127s %u %u\n", name, &a, &b) != 3) fatalf("error in reading the insert length range: %s", line);
            int32_t* range = xmalloc(2 * sizeof(int32_t));
            range[0] = (int32_t)a; range[1] = (int32_t)b;
            qhash_add(insertlengths, name, (int)strlen(name), range);
            rg_order_push(name, range);
        } else if (strncmp(line, "RC", 2) == 0) {
            if (sscanf(line, "RC %127s %u\n", name, &a) != 2) fatalf("error in reading the mean coverage: %s", line);
            qbin* hit = qhash_lookup(id2chroms, name, (int)strlen(name));
            if (!hit) fatalf("did not find %s in the hash", name);
            g_meancov[(intptr_t)hit->val - 1] = a;
        } else fatalf("unknown tag in configuration: %s", line);
    }
    free(line);
    fclose(fp);
    qhash_free(id2chroms, NULL);
}

/* ---- observed coverage per contig (estimate_average_coverage, src/bamoperations.c:88-147) ---------------------------
 * The reference pileups every contig once more and prints floor(sum of the pileup's n / positions with n > 0) on stderr
 * (src/indelminer.c:728-733); nothing else reads the number.  The pileup's n at a position is the number of records -- not
 * unmapped, secondary, QC-fail or duplicate (BAM_DEF_MASK) -- whose reference span [pos, bam_calend) holds the position,
 * deletions and skips included: the sum is the sum of the spans, the covered positions are the union of the spans.  Both
 * come out of any walk of the records in file order: a running segment per walker, closed where the next record starts
 * behind its end; the few segments of all walkers are merged at the end.  (Not kept: the pileup buffer's cap of 8000
 * records starting at one position, bam_pileup.c.) */
typedef struct { int32_t tid, beg, end; } covseg;
typedef struct { int32_t nt; uint64_t* sum; covseg* seg; int64_t n, cap; int open; covseg cur; } covlist;
static uint32_t* g_meancov;         /* [n_targets]: from the RC lines of a config file, or observed */

static void cov_init(covlist* c, int32_t nt) { memset(c, 0, sizeof *c); c->nt = nt; c->sum = xcalloc((size_t)(nt > 0 ? nt : 1), sizeof(uint64_t)); }
static void cov_push(covlist* c, covseg sg)
{
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 64; c->seg = xrealloc(c->seg, sizeof(covseg) * (size_t)c->cap); }
    c->seg[c->n++] = sg;
}
static void cov_close(covlist* c) { if (c->open) { cov_push(c, c->cur); c->open = 0; } }
static void cov_free(covlist* c) { free(c->sum); free(c->seg); memset(c, 0, sizeof *c); }
static inline void cov_record(covlist* c, const bam_record* b)
{
    if (b->flag & (0x4 | 0x100 | 0x200 | 0x400)) return;
    if (b->tid < 0 || b->tid >= c->nt || b->pos < 0) return;
    const int32_t end = bam_record_end(b);
    if (end <= b->pos) return;                                  /* no reference base: the pileup drops it unseen */
    c->sum[b->tid] += (uint64_t)(end - b->pos);
    if (c->open && c->cur.tid == b->tid && b->pos >= c->cur.beg && b->pos <= c->cur.end) { if (end > c->cur.end) c->cur.end = end; return; }
    cov_close(c);
    c->cur.tid = b->tid; c->cur.beg = b->pos; c->cur.end = end; c->open = 1;
}
static int cmp_covseg(const void* x, const void* y)
{
    const covseg* a = x; const covseg* b = y;
    if (a->tid != b->tid) return a->tid < b->tid ? -1 : 1;
    if (a->beg != b->beg) return a->beg < b->beg ? -1 : 1;
    return 0;
}
/* sums[nt] and the segments of every walker -> g_meancov */
static void cov_means(int32_t nt, const uint64_t* sums, covseg* seg, int64_t n)
{
    if (!g_meancov) g_meancov = xcalloc((size_t)(nt > 0 ? nt : 1), sizeof(uint32_t));
    uint64_t* covered = xcalloc((size_t)(nt > 0 ? nt : 1), sizeof(uint64_t));
    qsort(seg, (size_t)n, sizeof(covseg), cmp_covseg);
    for (int64_t i = 0; i < n; ) {
        const int32_t t = seg[i].tid;
        int32_t beg = seg[i].beg, end = seg[i].end;
        for (i++; i < n && seg[i].tid == t && seg[i].beg <= end; i++) if (seg[i].end > end) end = seg[i].end;
        covered[t] += (uint64_t)(end - beg);
    }
    for (int32_t t = 0; t < nt; t++) if (covered[t]) g_meancov[t] = (uint32_t)floor((double)sums[t] * 1.0 / (double)covered[t]);
    free(covered);
}
static void cov_means_of_lists(int32_t nt, covlist* const* ls, int n_lists)
{
    uint64_t* sums = xcalloc((size_t)(nt > 0 ? nt : 1), sizeof(uint64_t));
    int64_t n = 0;
    for (int i = 0; i < n_lists; i++) { cov_close(ls[i]); n += ls[i]->n; }
    covseg* seg = xmalloc(sizeof(covseg) * (size_t)(n ? n : 1));
    n = 0;
    for (int i = 0; i < n_lists; i++) {
        for (int32_t t = 0; t < nt; t++) sums[t] += ls[i]->sum[t];
        if (ls[i]->n) memcpy(seg + n, ls[i]->seg, sizeof(covseg) * (size_t)ls[i]->n);
        n += ls[i]->n;
    }
    cov_means(nt, sums, seg, n);
    free(sums); free(seg);
}
static void cov_print_table(const bam_header* hdr)
{
    /* src/indelminer.c:728-733 */
    fprintf(stderr, "\nChromosomeID\tMean-coverage\n-------------\t-----------\n");
    for (int32_t i = 0; i < hdr->n_targets; i++) fprintf(stderr, "%d\t%u\n", i, g_meancov ? g_meancov[i] : 0u);
    fprintf(stderr, "-------------\t-----------\n\n");
}

static void estimate_insertlengths(driver* d, int chromid)
{
    /* src/bamoperations.c:15-86: min / max proper-pair isize per read group */
    bgzf_reader* r = bgzf_open(d->bam_name);
    if (!r) fatalf("error in opening the file %s", d->bam_name);
    bam_header* h = bam_header_load(r);
    bam_record b; memset(&b, 0, sizeof b);
    covlist cov;
    cov_init(&cov, h->n_targets);
    for (int32_t t = 0; t < h->n_targets; t++) {
        if (chromid != -1 && t != chromid) continue;
        bam_region_iter it;
        if (bam_region_begin(&it, r, d->idx, t, 0, h->target_len[t]) != 0) continue;
        while (bam_region_next(&it, &b) == 1) {
            cov_record(&cov, &b);
            if ((b.flag & 0x1) == 0 || (b.flag & 0x4) || (b.flag & 0x2) == 0) continue;
            if (b.flag & (0x100 | 0x200 | 0x400)) continue;
            if (b.isize < 0) continue;
            const uint8_t* rg = bam_aux_find(&b, "RG");
            const char* rgname = "generic";
            if (rg) { forceassert(rg[0] == 'Z'); rgname = bam_aux_str(rg); }
            const int32_t isize = b.isize;
            if (b.mpos - b.pos < 0) continue;
            if (isize < b.mpos - b.pos) continue;
            qbin* q = qhash_lookup(d->insertlengths, rgname, (int)strlen(rgname));
            if (!q) {
                int32_t* range = xmalloc(2 * sizeof(int32_t));
                range[0] = range[1] = isize;
                qhash_add(d->insertlengths, rgname, (int)strlen(rgname), range);
                rg_order_push(rgname, range);
            } else {
                int32_t* range = q->val;
                if (range[0] > isize) range[0] = isize;
                if (range[1] < isize) range[1] = isize;
            }
        }
    }
    free(b.data);
    { covlist* one = &cov; cov_means_of_lists(h->n_targets, &one, 1); cov_free(&cov); }
    bam_header_free(h);
    bgzf_close(r);
}

/* --------------------------------------------------------------- preamble -- */

static void print_vcf_preamble(void)
{
    /* src/shared.c:84-109, byte for byte */
    printf("##fileformat=VCFv4.1\n");
    printf("##%sVersion=%2.2f\n", "indelminer", INDELMINER_VERSION);
    printf("##INFO=<ID=INSERTION,Number=0,Type=Flag,Description=\"Indicates that the variant is an insertion.\">\n");
    printf("##INFO=<ID=DELETION,Number=0,Type=Flag,Description=\"Indicates that the variant is a deletion.\">\n");
    printf("##INFO=<ID=SPLIT_READ,Number=0,Type=Flag,Description=\"Indicates that at least one split read supports this variant.\">\n");
    printf("##INFO=<ID=PAIRED_READ,Number=0,Type=Flag,Description=\"Indicates that at least one PE read supports this variant.\">\n");
    printf("##INFO=<ID=COMPOSITE,Number=0,Type=Flag,Description=\"Indicates that at least one split read and at least one PE read supports this variant.\">\n");
    printf("##INFO=<ID=NS,Number=1,Type=Integer,Description=\"Number of reads supporting the variant\">\n");
    printf("##INFO=<ID=END,Number=1,Type=Integer,Description=\"end position of the variant described in this record\">\n");
    printf("##INFO=<ID=BP_END,Number=1,Type=Integer,Description=\"possible 3' end of the breakpoint described in this record\">\n");
    printf("##INFO=<ID=NFS,Number=1,Type=Integer,Description=\"Number of reads supporting the variant on the forward strand\">\n");
    printf("##INFO=<ID=NRS,Number=1,Type=Integer,Description=\"Number of reads supporting the variant on the forward strand\">\n");
    printf("##INFO=<ID=UTAILS,Number=1,Type=Integer,Description=\"The number of unique tail distances in supporting reads for this variant\">\n");
    printf("##INFO=<ID=MQ,Number=1,Type=Integer,Description=\"RMS mapping quality of the reads covering the breakpoints\">\n");
    printf("##INFO=<ID=MQ30,Number=1,Type=Integer,Description=\"Number of reads with mapping quality greater than or equal to 30, covering the breakpoints\">\n");
    printf("##INFO=<ID=DF,Number=1,Type=Integer,Description=\"Average number of other differences on reads supporting the reported variant\">\n");
    printf("##INFO=<ID=DP,Number=1,Type=Integer,Description=\"Average read depth across the breakpoints\">\n");
    printf("##INFO=<ID=BF,Number=2,Type=Integer,Description=\"Flanks from the split read or pair best sorrounding the variant\">\n");
}

/* ----------------------------------------------------------------- pass B -- */

/* the evidence one candidate read contributes: the realigned segments when the GPU found any
 * (they replace the CIGAR-derived ones, src/indelminer.c:494-502), else the CIGAR-derived */
static void resolve_candidate(driver* d, const item_t* it, const im_read_result* r, int32_t tid)
{
    const cand_batch* cb = &d->cb;
    const int c = it->cand;
    if (r->status == IM_ST_EVIDENCE && r->n_ev > 0) {
        seglist whole;
        whole.ref_start = r->ref_start; whole.n = r->n_ops;
        whole.ops = (uint32_t*)r->ops;
        const int64_t len = cb->base_off[c + 1] - cb->base_off[c];
        char* bases = xmalloc((size_t)len + 1);
        memcpy(bases, cb->bases + cb->base_off[c], (size_t)len); bases[len] = 0;
        whole.bases = bases;
        for (int k = 0; k < r->n_ev; k++) {
            const im_evidence* ge = &r->ev[k];
            evidence_t* e = xcalloc(1, sizeof *e);
            e->type = EV_SPLIT_READ; e->cls = ge->cls; e->strand = cb->strand[c]; e->qual = cb->qual[c];
            e->qname = xstrdup(cb->qname[c]);
            e->aln = seglist_copy(&whole);
            e->seg = ge->seg; e->b1 = ge->b1; e->b2 = ge->b2;
            e->lflank = ge->lflank; e->rflank = ge->rflank; e->nd_print = ge->nd_print; e->nd_filter = ge->nd_filter;
            pending_push(d, e);
        }
        free(bases);
        for (int k = 0; k < it->nbwa; k++) evidence_free(it->bwa[k]);
    } else {
        for (int k = 0; k < it->nbwa; k++) pending_push(d, it->bwa[k]);
    }
    (void)tid;
}

static void run_contig(driver* d, int32_t tid, int32_t beg, int32_t end, bgzf_reader* r)
{
    d->n_items = 0; d->n_flushes = 0;
    cb_reset(&d->cb);
    bam_region_iter it;
    bam_record b; memset(&b, 0, sizeof b);
    if (bam_region_begin(&it, r, d->idx, tid, beg, end) != 0) fatalf("cannot seek in %s", d->bam_name);
    d->n_seg = 0;
    const int whole = (beg <= 0 && end >= d->hdr->target_len[tid]);
    volatile int died = 0;              /* a record the reference dies on ended the pass: the flushes in front of it are still to print */
    t_is_main_thread_of_passA = 1;
    if (setjmp(g_passA_jmp)) died = 1;
    else g_passA_armed = 1;
    while (!died && bam_region_next(&it, &b) == 1) {
        if (whole && b.tid >= 0 && !(b.flag & (0x4 | 0x100 | 0x200 | 0x400))) {
            /* what samtools' pileup would count for DP= (bam_pileup.c:171-172,238-265) */
            const uint8_t* cig = BAMR_CIGAR(&b);
            int32_t x = b.pos;
            for (int kk = 0; kk < b.n_cigar; kk++) {
                const int op = CIG_OP(bamr_cigar_at(cig, kk)), len = CIG_LEN(bamr_cigar_at(cig, kk));
                if (op == OP_M || op == OP_EQ || op == OP_X) {
                    if (d->n_seg == d->cap_seg) {
                        d->cap_seg = d->cap_seg ? d->cap_seg * 2 : (1 << 16);
                        d->seg_start = xrealloc(d->seg_start, sizeof(int32_t) * (size_t)d->cap_seg);
                        d->seg_len = xrealloc(d->seg_len, sizeof(int32_t) * (size_t)d->cap_seg);
                    }
                    d->seg_start[d->n_seg] = x; d->seg_len[d->n_seg] = len; d->n_seg++;
                    x += len;
                } else if (op == OP_D || op == OP_N) x += len;
            }
        }
        dispatch_record(d, &b);
    }
    g_passA_armed = 0;
    free(b.data);
    phase_time("pass A (BAM decode + dispatch)");
    d->depth_tid = -1;
    if (whole) {
        gpu_wait(d);
        if (im_depth_build(d->gpu, d->seqlen[tid], (int32_t)d->n_seg, d->seg_start, d->seg_len) != IM_OK)
            fatalf("im_depth_build: %s", im_last_error(d->gpu));
        d->depth_tid = tid;
    }
    phase_time("depth array (device)");

    im_read_result* res = NULL;
    if (d->cb.n > 0) {
        im_params P = { O.klength, O.numgaps, O.maxdelsize, O.ethreshold };
        im_read_batch batch = { d->cb.n, d->cb.bases, d->cb.base_off, d->cb.tid, d->cb.anchor, d->cb.range_max };
        res = xmalloc(sizeof(im_read_result) * (size_t)d->cb.n);
        gpu_wait(d);
        const int rc = im_realign_batch(d->gpu, &P, &batch, res);
        if (rc != IM_OK) fatalf("im_realign_batch: %s", im_last_error(d->gpu));
    }
    phase_time("realign batch (device, incl. copies)");
    int f = 0;
    for (int64_t i = 0; i <= d->n_items; i++) {
        while (f < d->n_flushes && d->flushes[f].n_items == i) {
            flush_variants(d, d->flushes[f].tid, d->flushes[f].marker);
            f++;
        }
        if (i == d->n_items) break;
        const item_t* itm = &d->items[i];
        if (itm->kind == ITEM_CAND) { resolve_candidate(d, itm, &res[itm->cand], tid); free(itm->bwa); }
        else pending_push(d, itm->pe);
    }
    free(res);
    if (died) {
        /* what the reference had printed when it met the record is out; its message and status follow */
        out_flush_on_exit();
        fflush(stdout);
        if (g_passA_msg[0] == 1) fprintf(stderr, "%s\n", g_passA_msg + 1);     /* an assertion's own form */
        else fprintf(stderr, "indelminer: %s\n", g_passA_msg);
        exit(EXIT_FAILURE);
    }
    flush_variants(d, tid, INT_MAX);        /* end of contig (src/indelminer.c:806-823) */
    phase_time("pass B (cluster, merge, print)");
    if (g_vcfname != NULL) {
        /* what print_knownvariants left over (src/indelminer.c:839-847) */
        for (int ki = g_known.next; ki < g_known.n; ki++) {
            knownvariant_t* k = g_known.v[ki];
            print_vcf_line(d, k);
            if (k->evdnctype == EV_SPLIT_READ && is_indel_supported(d, k)) printf(";%s", g_sample_name);
            printf("\n");
        }
        g_known.next = g_known.n;
    }
}

/* multi-GPU state (the section further down): declared here because the replay writes one part per contig */
#define MG_MAX_RG    64
#define MG_RG_WORDS  18         /* name[48] + min + max + first_tid + first position + seen on a proper pair + first record */

typedef struct {
    int rank, world, local_rank;
    im_comm* comm;
    char dir[400];
    int32_t* owner;             /* [n_targets] the rank that walks the contig (mg_plan) */
    int64_t* piece_prefix;      /* [pieces] counted reads of the run in front of each piece */
    int32_t* claim_walker;      /* [claims] the rank that walks the claim (reads the file, runs the triage) */
    int32_t* piece_walker;      /* [pieces] the same per piece */
    int32_t* claim_owner;       /* [claims] the rank that stages and replays it: the owner of its contig */
    int      split;             /* some claim is walked by a rank that does not own it (pieces of a contig over several GPUs) */
    int*     floor;             /* [n_targets] smallest start of a stale pair-table entry of an earlier contig */
    int      out_fd;            /* rank 0: the real stdout */
    uint8_t* skip;              /* [n_targets] annotate mode: contigs without known variants are not walked at all */
    int      abort_tid;         /* -1, or the first contig of this rank that holds a record the reference dies on */
    int      cross;             /* the exchanged pair-table logs show entries of one contig meeting records of another */
    int      abort_piece;       /* the first piece (plan index) whose walk some rank did not survive; the number of pieces: none */
} mgpu;

static mgpu* g_mg = NULL;
static int g_mg_self_ship = 0;          /* tests: claims this rank owns AND walks go through the exchange too */
static driver* g_mg_driver = NULL;
static void mg_finish(mgpu* m, driver* d);
static int g_mg_cur_tid = -1;          /* the first contig of the claim the main thread is working on */

static void mg_path(const mgpu* m, char* out, size_t cap, const char* what, int idx) { snprintf(out, cap, "%s/%s.%d", m->dir, what, idx); }
