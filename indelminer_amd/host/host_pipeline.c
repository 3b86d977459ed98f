/* host_pipeline.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: the device pipeline: walkers' chunk rings and triage, groups of pieces, pair table and flush points of a group, the stage
 * (realign + flush list + group-by on the device, results back), the replay of a group. */

/* ======================================================== device pipeline == */
/*
 * Whole-contig runs.  The host's part shrinks to what north_star keeps on it -- BGZF inflate, walking
 * the record stream, the discordant-pair table, merge / filter / print -- and everything per read
 * happens on the device without coming back in between:
 *
 *   walk     records are inflated STRAIGHT INTO PINNED CHUNKS (bam_region_next_raw), each chunk goes
 *            to the GPU with one asynchronous copy and is triaged there (im_dev_triage: fetch_func's
 *            candidate rules, base decode + reverse complement, CIGAR-derived evidence, the DP=
 *            pileup segments); the walking thread itself only counts reads (READCHUNK flush points
 *            and their markers, src/indelminer.c:617-623) and serves the pair table (516-615).
 *            Candidates accumulate on the device over the chunks of a GROUP of contigs, so that one
 *            realign launch fills the chip.
 *   run      one im_dev_realign_keep over the group's candidates, one im_dev_flush_cut per flush
 *            point in file order (which evidence each flush consumes), one im_dev_cluster_groupby;
 *            back come the realign results, the consumed marks and the cluster records.
 *   replay   per flush: variants from the device's clusters + the host's paired-read components,
 *            merge_variants, print_variants -- the reference's own order of output.
 */

/* A walker's ring: up to PIPE_NCHUNK pinned chunks (and their device twins) of PIPE_CHUNK_BYTES.  Pinning memory is slow
 * (and serialised in the driver): an input of a few hundred megabytes gets two chunks of 16 MB per walker instead of four of
 * 32 MB, set once before the walkers start (walkpool_start). */
#define PIPE_NCHUNK        4
static uint32_t g_chunk_bytes = 32u << 20;
static int g_nchunk = PIPE_NCHUNK;
static int g_small_input;           /* walkpool_start: an input of less than 64 MB -- no streams of the pipelines' own */
#define PIPE_CHUNK_BYTES   g_chunk_bytes
#define PIPE_CHUNK_RECS    (g_chunk_bytes / 64u)

typedef struct {
    uint8_t*  h_raw; uint32_t* h_off; int32_t* h_cnt;        /* pinned */
    void     *d_raw, *d_off, *d_class, *d_scratch;
    size_t    scratch_bytes;
    int32_t   n; uint32_t bytes; int64_t rec_base, seq_bytes;
    im_event* done;
    int       busy;
} pchunk;

typedef struct { int64_t rec; int32_t pe; int marker; int32_t tid; } gflush;
typedef struct { char name[48]; int32_t min, max, first_tid; int64_t first_rec; } rgstat_t;    /* first_rec: position << 32 | record index in its piece */
static int g_onepass;               /* set before the walkers start */

/* A PIECE of a contig: the records that start in [beg, end).  Whole small contigs are pieces too (first and last at once).
 * Pieces are what the walkers claim: a contig of any size spreads over all of them. */
typedef struct { int32_t tid, beg, end; int first, last; int64_t weight; int overlap; int32_t index; } piece_t;   /* overlap: a -c region's first piece also takes the records that begin in front of it and reach into it (bam_fetch) */
static int g_region_tid = -1, g_region_beg = 0, g_region_end = 0;      /* -c: the one stretch this run works on */

typedef struct {
    int32_t tid; int64_t rec0, rec1; int32_t pe0, pe1; int fl0, fl1;
    int64_t cn0, cn1; int32_t lm0, lm1;     /* this piece's runs in the counted-read log and the live-minimum log */
    int left_min;                           /* smallest start among the pair-table entries still waiting at the piece's end */
    int64_t dn0, dn1, sn0, sn1;             /* this piece's runs in the group's name logs (pair-table records; entries left waiting) */
    int32_t beg, end; int first, last;      /* the piece */
    int lm_init;                            /* find_marker's value when the piece begins (entries earlier pieces left in the table) */
    int64_t n_counted;                      /* counted reads of the piece (src/indelminer.c:617) */
    int32_t fp0, fp1;                       /* this piece's run of the group's flush points */
    int32_t piece;                          /* index of the piece in the run's plan */
} gcontig;
typedef struct { int64_t rec; int32_t pos; } gfpoint;     /* a READCHUNK flush point: the record bound and the position of the counted read */

struct pgroup_s;
/* What a piece leaves for the next piece of its contig: evidence no flush has consumed yet.  Split-read evidence travels as the
 * candidate it came from (its pending slots; the origin group keeps the realigned record and the BAM record), paired-read evidence
 * as the object.  `when` = piece sequence number << 32 | record index in that piece: the order of arrival over the whole contig. */
typedef struct {
    int64_t when;
    struct pgroup_s* g; int32_t cand;       /* split-read: origin group and candidate index there; g == NULL: paired-read */
    evidence_t* pe;
    int32_t cls[IM_MAX_EV], b1[IM_MAX_EV], b2[IM_MAX_EV];   /* split-read: the pending slots, -1 = consumed or empty */
} carry_item;
typedef struct { carry_item* v; int32_t n, cap; } carry_list;

typedef struct pgroup_s {
    int64_t n_rec;
    gcontig* ctg; int n_ctg, cap_ctg, cur_ctg;      /* cur_ctg: the piece the pair table is serving (host_discordant) */
    gflush* fl; int n_fl, cap_fl;
    gfpoint* fp; int32_t n_fp, cap_fp;
    evidence_t** pe; int64_t* pe_rec; int32_t n_pe, cap_pe;
    /* The walk does not know the global read counter it starts from (several pieces are walked at once), so it cannot
     * place the READCHUNK flush points itself (src/indelminer.c:617-670).  It logs what placing them needs -- for every
     * counted read its record bound and position -- and group_resolve_flushes places them once the pieces before this one
     * have been counted; the pair table's smallest waiting start is logged whenever it moves (group_pair_table). */
    /* the log of counted reads (record number, position): two bytes per read -- the steps from the read in front of it -- under
     * checkpoints of absolute values every CN_STRIDE reads, at the start of a piece and wherever a step does not fit a byte
     * (a whole-genome run counts ~1e9 reads: whole values were 7 GB of the run's memory; cn_at() looks a read up) */
    uint8_t* cn_step; int64_t n_cn, cap_cn;
    struct cn_ck { int64_t k; int32_t rec, pos; }* cn_ck; int64_t n_cnck, cap_cnck;
    int32_t cn_last_rec, cn_last_pos;
    int32_t *lm_rec; int *lm_val; int32_t n_lm, cap_lm;
    /* The reference keeps ONE pair table for the run (readpairs is never reset): a first mate left waiting in one contig is found
     * by a record of the same name in a later contig.  Contigs are worked on independently here, each with a table of its own, so the
     * names are logged -- of every record that goes through the table (dn) and of the entries a contig leaves waiting (sn) -- and
     * the main thread, taking the groups in contig order, hands the run to the record-at-a-time path if they ever meet. */
    char *dn, *sn; int64_t dn_len, dn_cap, sn_len, sn_cap;
    /* per-read-group insert-size extrema as estimate_insertlengths takes them (one-pass mode: no config file, the table is made
     * by the walk), the records of not-proper pairs kept aside for the pair table (served on the main thread, piece after piece of
     * a contig: the walkers run ahead of one another), and the group's candidate arrays parked in a device allocation of their
     * own until the main thread stages them */
    rgstat_t rgs[MG_MAX_RG]; int n_rgs;
    covlist cov;                        /* one-pass mode: the group's share of the observed coverage */
    char crg[MG_MAX_RG][48]; int n_crg, crg_last;   /* the read groups met on COUNTED reads (the reference looks each one's up) */
    uint8_t* npp_raw; int64_t npp_len, npp_cap; int64_t* npp_off; int32_t* npp_rec; int32_t n_npp, cap_npp;
    void* sv[10]; int32_t sv_n; int64_t sv_bytes; int32_t* sv_range;
    /* candidates as the device found them: record index + a host copy of the raw record */
    int32_t n_cand, cap_cand; int32_t* cand_rec; int64_t* craw_off; uint8_t* craw; int64_t craw_len, craw_cap;
    /* The stage: in front of the group's own candidates sit the ones earlier pieces of the contig left pending (front[], in order
     * of arrival), in front of its paired-read entries the pending ones; n_virt = how many such items there are in all -- the
     * group's own records count on from there, so that record numbers order the whole stage by arrival. */
    carry_item* front; int32_t n_front; int32_t* front_virt;
    int32_t n_pe_front, n_virt; int phantom;
    int seq;                                /* position of the group in the run (the `when` of what it leaves pending) */
    int from_package;                       /* multi-GPU: walked by another rank (its flush points came with it) */
    /* what came back from the stage */
    im_read_result* res; int32_t* res_slot;     /* the realigned records that hold evidence, packed; per own candidate its place there or -1 */
    int32_t *s_cls, *s_b1, *s_b2, *cons_sr, *cons_pe;
    int32_t n_cl, n_nodes; int32_t *cl_key, *cl_first, *cl_count, *order, *cl_sorted;
    evidence_t** ev_cache;
    /* groups of one contig are freed together, when the last of them has been replayed (pending evidence points back at them) */
    struct pgroup_s* next_of_contig;
} pgroup;

typedef struct {
    driver* d;
    void* stream;
    pchunk ck[PIPE_NCHUNK];
    int cur, oldest, n_busy;
    /* device arrays of the group (growable) */
    int32_t cap_cand; int64_t cap_bases; int32_t cap_pe, cap_fl;
    void *bases, *boff, *len, *tid, *anchor, *range, *res, *cls, *b1, *b2, *consumed, *cand_rec, *counters, *cut;
    void *rstat, *rslot, *rcompact, *rcount; int32_t cap_rc;    /* im_dev_compact_results of the stage pipeline */
    void* h_stage; size_t h_stage_cap;                          /* pinned block the stage's results arrive in */
    void *order, *clkey, *clfirst, *clcount, *counts, *gscratch, *fdesc, *fgscratch; size_t gscratch_bytes, fgscratch_bytes;
    /* confirmed by harvested chunks / still in flight */
    int32_t conf_cand, conf_err; int64_t conf_bytes, fly_recs, fly_seq;
    im_triage_params tp;
    int ready, own_stream;
    uint8_t* h_ring;                    /* the chunks' pinned memory when it is one allocation */
} ppipe;

#define GPU(call) do { if ((call) != IM_OK) fatalf("%s: %s", #call, im_last_error(P->d->gpu)); } while (0)
#define GPU2(drv, call) do { if ((call) != IM_OK) fatalf("%s: %s", #call, im_last_error((drv)->gpu)); } while (0)
struct walkpool_s;
static struct walkpool_s* g_handoff_pool = NULL;       /* set while run_pipeline can hand a run the reference aborts to a child */
static int g_free_slabs = 1;                            /* parked groups' device slabs are freed once staged */
static void pipeline_handoff(void);

static void* pdev_alloc(ppipe* P, size_t bytes) { void* p = NULL; GPU(im_dev_alloc(P->d->gpu, bytes ? bytes : 256, &p)); return p; }

static void pipe_alloc_cands(ppipe* P, int32_t cap_cand, int64_t cap_bases, int32_t cap_pe)
{
    const size_t nsl = (size_t)cap_cand * IM_MAX_EV + (size_t)cap_pe;
    P->bases = pdev_alloc(P, (size_t)cap_bases);
    P->boff = pdev_alloc(P, 8 * (size_t)cap_cand); P->len = pdev_alloc(P, 4 * (size_t)cap_cand);
    P->tid = pdev_alloc(P, 4 * (size_t)cap_cand); P->anchor = pdev_alloc(P, 4 * (size_t)cap_cand);
    P->range = pdev_alloc(P, 4 * (size_t)cap_cand); P->cand_rec = pdev_alloc(P, 4 * (size_t)cap_cand);
    P->res = pdev_alloc(P, sizeof(im_read_result) * (size_t)cap_cand);
    P->cls = pdev_alloc(P, 4 * nsl); P->b1 = pdev_alloc(P, 4 * nsl); P->b2 = pdev_alloc(P, 4 * nsl); P->consumed = pdev_alloc(P, 4 * nsl);
    P->order = pdev_alloc(P, 4 * nsl); P->clkey = pdev_alloc(P, 16 * nsl); P->clfirst = pdev_alloc(P, 4 * nsl); P->clcount = pdev_alloc(P, 4 * nsl);
    P->gscratch_bytes = im_dev_groupby_scratch_bytes((int32_t)nsl);
    P->gscratch = pdev_alloc(P, P->gscratch_bytes);
    GPU(im_dev_groupby_scratch_init(P->d->gpu, (int32_t)nsl, P->gscratch, P->gscratch_bytes, P->stream));
    /* the chip-wide flush list + group-by (im_dev_flush_groupby): its table and the range-minimum tree over the flush list */
    P->fgscratch_bytes = im_dev_flushgroup_scratch_bytes(cap_cand * IM_MAX_EV, P->cap_fl);
    P->fgscratch = pdev_alloc(P, P->fgscratch_bytes);
    GPU(im_dev_flushgroup_scratch_init(P->d->gpu, cap_cand * IM_MAX_EV, P->cap_fl, P->fgscratch, P->fgscratch_bytes, P->stream));
    P->cap_cand = cap_cand; P->cap_bases = cap_bases; P->cap_pe = cap_pe;
}

static void pipe_free_cands(ppipe* P)
{
    void* all[] = { P->bases, P->boff, P->len, P->tid, P->anchor, P->range, P->cand_rec, P->res, P->cls, P->b1, P->b2, P->consumed,
                    P->order, P->clkey, P->clfirst, P->clcount, P->gscratch, P->fgscratch };
    for (size_t i = 0; i < sizeof all / sizeof all[0]; i++) if (all[i]) im_dev_free(P->d->gpu, all[i]);
}

/* once per run, when the reference is on the device and the insert lengths are known */
static void pipe_global_init(driver* d)
{
    g_lean_evidence = g_vcfname == NULL && strncmp(O.outputformat, "vcf", 3) == 0;     /* host_logic.c: what nothing reads is not built */
    /* the insert-length table in the order its entries were added, range[1] of each */
    int32_t* rmax = xmalloc(sizeof(int32_t) * (size_t)(g_rg_n ? g_rg_n : 1));
    for (int i = 0; i < g_rg_n; i++) rmax[i] = g_rg_range[i][1];
    if (im_set_insert_ranges(d->gpu, g_rg_n, g_rg_name, rmax) != IM_OK) fatalf("im_set_insert_ranges: %s", im_last_error(d->gpu));
    free(rmax);
    if (im_depth_enable(d->gpu) != IM_OK) fatalf("im_depth_enable: %s", im_last_error(d->gpu));
}

/* one walker's buffers (with_chunks: the pinned chunk ring a walk delivers records through; the main thread's stage pipeline
 * has none): needs the GPU context (d->gpu), nothing else of the driver yet */
static void pipe_init(ppipe* P, driver* d, int with_chunks)
{
    memset(P, 0, sizeof *P);
    P->d = d;
    const int64_t t_in = wall_ns();
    /* a stream per walker: its uploads and triage launches, and the device stage of its groups.  INDELMINER_STREAMS=shared
     * puts every walker on the context's stream instead (the GPU then sees the run exactly as with one walker: a
     * debugging aid -- it is how the group-by scratch bug of profiles/README.md r02 was told apart from a device race) */
    {
        const char* sm = getenv("INDELMINER_STREAMS");             /* shared | own; default: own, except on small inputs (walkpool_start) */
        P->own_stream = sm ? strcmp(sm, "shared") != 0 : !g_small_input;
    }
    if (P->own_stream) GPU(im_stream_create(d->gpu, &P->stream));
    else P->stream = im_ctx_stream(d->gpu);
    /* The ring's pinned memory is ONE allocation (per chunk the records, their offsets, 64 bytes of counts): sixteen walkers pinning
     * three blocks per chunk each queue up in the driver (INDELMINER_RING=split is that layout, kept for the A/B). */
    const size_t pin_off = (((size_t)PIPE_CHUNK_BYTES + 255) & ~(size_t)255) + 256, pin_cnt = pin_off + ((4 * ((size_t)PIPE_CHUNK_RECS + 1) + 255) & ~(size_t)255);
    const size_t pin_chunk = pin_cnt + 256;
    const int ring_one = !(getenv("INDELMINER_RING") && strcmp(getenv("INDELMINER_RING"), "split") == 0);
    if (with_chunks && ring_one) GPU(im_host_alloc(d->gpu, pin_chunk * (size_t)g_nchunk, (void**)&P->h_ring));
    for (int i = 0; i < g_nchunk && with_chunks; i++) {
        pchunk* c = &P->ck[i];
        if (ring_one) {
            c->h_raw = P->h_ring + pin_chunk * (size_t)i;
            c->h_off = (uint32_t*)(c->h_raw + pin_off);
            c->h_cnt = (int32_t*)(c->h_raw + pin_cnt);
        } else {
            GPU(im_host_alloc(d->gpu, PIPE_CHUNK_BYTES, (void**)&c->h_raw));
            GPU(im_host_alloc(d->gpu, 4 * ((size_t)PIPE_CHUNK_RECS + 1), (void**)&c->h_off));
            GPU(im_host_alloc(d->gpu, 64, (void**)&c->h_cnt));
        }
        c->d_raw = pdev_alloc(P, PIPE_CHUNK_BYTES + 64);
        c->d_off = pdev_alloc(P, 4 * ((size_t)PIPE_CHUNK_RECS + 1));
        c->d_class = pdev_alloc(P, PIPE_CHUNK_RECS);
        c->scratch_bytes = im_dev_triage_scratch_bytes((int32_t)PIPE_CHUNK_RECS);
        c->d_scratch = pdev_alloc(P, c->scratch_bytes);
        GPU(im_dev_triage_scratch_init(d->gpu, (int32_t)PIPE_CHUNK_RECS, c->d_scratch, c->scratch_bytes, P->stream));
        GPU(im_event_create(d->gpu, &c->done));
    }
    const int64_t t_ring = wall_ns();
    P->counters = pdev_alloc(P, 64);
    P->counts = pdev_alloc(P, 64);
    /* device allocations are not zeroed (a recycled block keeps what its previous owner wrote): the triage's running counts start from 0 */
    GPU(im_dev_memset(d->gpu, P->counters, 0, 64, P->stream));
    GPU(im_dev_memset(d->gpu, P->counts, 0, 64, P->stream));
    P->cap_fl = 4096;
    P->cut = pdev_alloc(P, 8 * (size_t)P->cap_fl);
    P->fdesc = pdev_alloc(P, sizeof(im_flush_desc) * (size_t)P->cap_fl);
    pipe_alloc_cands(P, 1 << 20, (int64_t)(1 << 20) * 160, 1 << 16);
    P->tp.qthreshold = O.qthreshold; P->tp.ethreshold_vcfcheck = O.ethreshold_vcfcheck; P->tp.maxpedelsize = O.maxpedelsize;   /* options are parsed before any thread starts */
    P->tp.want_depth = g_region_tid < 0;        /* -c: DP= comes from the file around each variant, like the reference's (region_depth) */
    P->tp.defer_ranges = g_onepass;
    P->ready = 1;
    if (g_timing) fprintf(stderr, "[timing]     a pipeline's buffers: %.2f ms (stream + %d ring chunks %.2f, candidate arrays %.2f)\n", (wall_ns() - t_in) / 1e6, with_chunks ? g_nchunk : 0, (t_ring - t_in) / 1e6, (wall_ns() - t_ring) / 1e6);
}

static void pipe_destroy(ppipe* P)
{
    if (!P->ready) return;
    for (int i = 0; i < PIPE_NCHUNK && P->ck[i].h_raw; i++) {
        pchunk* c = &P->ck[i];
        if (!P->h_ring) { im_host_free(P->d->gpu, c->h_raw); im_host_free(P->d->gpu, c->h_off); im_host_free(P->d->gpu, c->h_cnt); }
        im_dev_free(P->d->gpu, c->d_raw); im_dev_free(P->d->gpu, c->d_off); im_dev_free(P->d->gpu, c->d_class); im_dev_free(P->d->gpu, c->d_scratch);
        im_event_destroy(c->done);
    }
    pipe_free_cands(P);
    if (P->rstat) { im_dev_free(P->d->gpu, P->rstat); im_dev_free(P->d->gpu, P->rslot); im_dev_free(P->d->gpu, P->rcompact); }
    if (P->rcount) im_dev_free(P->d->gpu, P->rcount);
    if (P->h_stage) im_host_free(P->d->gpu, P->h_stage);
    if (P->h_ring) im_host_free(P->d->gpu, P->h_ring);
    im_dev_free(P->d->gpu, P->counters); im_dev_free(P->d->gpu, P->counts); im_dev_free(P->d->gpu, P->cut); im_dev_free(P->d->gpu, P->fdesc);
    if (P->own_stream) im_stream_destroy(P->d->gpu, P->stream);
    P->ready = 0;
}

static void group_free(pgroup* G)
{
    if (G->phantom && G->pe && G->n_pe_front > 0 && G->pe[G->n_pe_front - 1] && G->pe[G->n_pe_front - 1]->type == EV_PHANTOM) evidence_free(G->pe[G->n_pe_front - 1]);
    free(G->ctg); free(G->fl); free(G->fp); free(G->pe); free(G->pe_rec); free(G->cand_rec); free(G->craw_off); free(G->craw);
    free(G->cn_step); free(G->cn_ck); free(G->lm_rec); free(G->lm_val);
    free(G->npp_raw); free(G->npp_off); free(G->npp_rec); free(G->dn); free(G->sn);
    free(G->res); free(G->res_slot); free(G->s_cls); free(G->s_b1); free(G->s_b2); free(G->cons_sr); free(G->cons_pe); free(G->front); free(G->front_virt);
    free(G->cl_key); free(G->cl_first); free(G->cl_count); free(G->order); free(G->cl_sorted); free(G->ev_cache);
    free(G->cov.sum); free(G->cov.seg);
    memset(G, 0, sizeof *G);
}

/* the chunk's triage is complete: note what it found, copy its candidates' records to the host side store */
static int g_verify_triage;
static void pipe_harvest(ppipe* P, pgroup* G, pchunk* c)
{
    GPU(im_event_sync(c->done));
    const int32_t n_after = c->h_cnt[0], n_err = c->h_cnt[3];
    if (c->h_cnt[4] != 0) fatalf("internal: candidate buffers overflowed on the device");
    if (n_err > P->conf_err) {
        /* a record the reference exits on: replay it through the host's own fetch_func restatement for the
         * reference's message, or name the limit it ran into */
        uint8_t* cls = xmalloc((size_t)c->n);
        GPU(im_dev_download(P->d->gpu, cls, c->d_class, (size_t)c->n));
        for (int32_t i = 0; i < c->n; i++) {
            if (cls[i] < IM_REC_ERR_RG) continue;
            bam_record b;
            bam_record_view(c->h_raw + c->h_off[i], (int32_t)(c->h_off[i + 1] - c->h_off[i]), &b);
            /* a record the reference exits on, or one beyond a kernel limit (more than IM_MAX_EV indels of one CIGAR pass the
             * end-distance rule: check_variants has no such bound, src/indelminer.c:285-337): the record-at-a-time run, whose
             * CIGAR-derived evidence is made on the host, takes over when the main thread gets to this group */
            if (t_abort_jmp || g_main_in_walk) { free(cls); walker_bails_out(); }
            if (cls[i] == IM_REC_ERR_LIMIT)
                fatalf("read %s: more than %d indels in its CIGAR pass the end-distance rule, or the record is malformed (kernel limit IM_MAX_EV)", BAMR_QNAME(&b), IM_MAX_EV);
            dispatch_record(P->d, &b);
            fatalf("read %s: record rejected by the device triage (class %d)", BAMR_QNAME(&b), (int)cls[i]);
        }
        free(cls);
    }
    const int32_t fresh = n_after - P->conf_cand;
    if (fresh > 0) {
        if (n_after > G->cap_cand) {
            G->cap_cand = n_after * 2 + 1024;
            G->cand_rec = xrealloc(G->cand_rec, sizeof(int32_t) * (size_t)G->cap_cand);
            G->craw_off = xrealloc(G->craw_off, sizeof(int64_t) * ((size_t)G->cap_cand + 1));
        }
        GPU(im_dev_download(P->d->gpu, G->cand_rec + P->conf_cand, (char*)P->cand_rec + 4 * (size_t)P->conf_cand, 4 * (size_t)fresh));
        if (g_verify_triage) {
            /* INDELMINER_VERIFY_TRIAGE=1: the chunk's candidates as the device placed them against the records themselves */
            int64_t* boff = xmalloc(8 * (size_t)fresh); int32_t* len = xmalloc(4 * (size_t)fresh);
            int32_t* tid = xmalloc(4 * (size_t)fresh); int32_t* anc = xmalloc(4 * (size_t)fresh);
            GPU(im_dev_download(P->d->gpu, boff, (char*)P->boff + 8 * (size_t)P->conf_cand, 8 * (size_t)fresh));
            GPU(im_dev_download(P->d->gpu, len, (char*)P->len + 4 * (size_t)P->conf_cand, 4 * (size_t)fresh));
            GPU(im_dev_download(P->d->gpu, tid, (char*)P->tid + 4 * (size_t)P->conf_cand, 4 * (size_t)fresh));
            GPU(im_dev_download(P->d->gpu, anc, (char*)P->anchor + 4 * (size_t)P->conf_cand, 4 * (size_t)fresh));
            int64_t at = P->conf_bytes;
            for (int32_t j = 0; j < fresh; j++) {
                const int64_t li = (int64_t)G->cand_rec[P->conf_cand + j] - c->rec_base;
                if (li < 0 || li >= c->n || (j > 0 && G->cand_rec[P->conf_cand + j] <= G->cand_rec[P->conf_cand + j - 1]))
                    fatalf("verify: candidate %d of the chunk names record %d (chunk holds %ld..%ld)", j, G->cand_rec[P->conf_cand + j], (long)c->rec_base, (long)c->rec_base + c->n - 1);
                bam_record b;
                bam_record_view(c->h_raw + c->h_off[li], (int32_t)(c->h_off[li + 1] - c->h_off[li]), &b);
                if (boff[j] != at || len[j] != b.l_seq || tid[j] != b.mtid || anc[j] != b.mpos)
                    fatalf("verify: candidate %d (+%d) of the chunk, record %ld: device {off %ld len %d tid %d anchor %d}, record {off %ld len %d tid %d anchor %d}; "
                           "chunk of %d records, counters before %d / %ld, after %d / %d", j, P->conf_cand, (long)li, (long)boff[j], len[j], tid[j], anc[j],
                           (long)at, (int)b.l_seq, b.mtid, b.mpos, c->n, P->conf_cand, (long)P->conf_bytes, n_after, c->h_cnt[1]);
                at += ((int64_t)b.l_seq + 3) & ~(int64_t)3;
            }
            if (at != c->h_cnt[1]) fatalf("verify: the chunk's candidates end at byte %ld, the device says %d", (long)at, c->h_cnt[1]);
            free(boff); free(len); free(tid); free(anc);
        }
        for (int32_t j = P->conf_cand; j < n_after; j++) {
            const int64_t li = (int64_t)G->cand_rec[j] - c->rec_base;
            forceassert(li >= 0 && li < c->n);
            const uint32_t o = c->h_off[li], l = c->h_off[li + 1] - o;
            if (G->craw_len + l > G->craw_cap) { G->craw_cap = (G->craw_cap + l) * 2 + (1 << 20); G->craw = xrealloc(G->craw, (size_t)G->craw_cap); }
            memcpy(G->craw + G->craw_len, c->h_raw + o, l);
            G->craw_off[j] = G->craw_len;
            G->craw_len += l;
            G->craw_off[j + 1] = G->craw_len;
        }
    }
    P->conf_cand = n_after; P->conf_err = n_err; P->conf_bytes = c->h_cnt[1];
    G->n_cand = n_after;
    P->fly_recs -= c->n; P->fly_seq -= c->seq_bytes;
    c->busy = 0; c->n = 0; c->bytes = 0; c->seq_bytes = 0;
    P->n_busy--;
    P->oldest = (P->oldest + 1) % g_nchunk;
}

static void pipe_drain(ppipe* P, pgroup* G) { while (P->n_busy > 0) pipe_harvest(P, G, &P->ck[P->oldest]); }

/* every record in flight may turn out to be a candidate: make room before a chunk is sent */
static void pipe_ensure_capacity(ppipe* P, pgroup* G, int64_t add_recs, int64_t add_seq)
{
    int64_t need_c = (int64_t)P->conf_cand + P->fly_recs + add_recs;
    int64_t need_b = P->conf_bytes + P->fly_seq + add_seq + 64;
    if (need_c <= P->cap_cand && need_b <= P->cap_bases) return;
    pipe_drain(P, G);
    need_c = (int64_t)P->conf_cand + add_recs; need_b = P->conf_bytes + add_seq + 64;
    if (need_c <= P->cap_cand && need_b <= P->cap_bases) return;
    if (need_c > 0x1fffffff) fatalf("more than 2^29 candidate reads in one group of contigs");
    ppipe old = *P;
    int32_t nc = P->cap_cand; int64_t nb = P->cap_bases;
    while (nc < need_c) nc *= 2;
    while (nb < need_b) nb *= 2;
    pipe_alloc_cands(P, nc, nb, P->cap_pe);
    const size_t n = (size_t)P->conf_cand;
    im_ctx* g = P->d->gpu;
    GPU(im_dev_copy_async(g, P->bases, old.bases, (size_t)P->conf_bytes, P->stream));
    GPU(im_dev_copy_async(g, P->boff, old.boff, 8 * n, P->stream)); GPU(im_dev_copy_async(g, P->len, old.len, 4 * n, P->stream));
    GPU(im_dev_copy_async(g, P->tid, old.tid, 4 * n, P->stream)); GPU(im_dev_copy_async(g, P->anchor, old.anchor, 4 * n, P->stream));
    GPU(im_dev_copy_async(g, P->range, old.range, 4 * n, P->stream)); GPU(im_dev_copy_async(g, P->cand_rec, old.cand_rec, 4 * n, P->stream));
    GPU(im_dev_copy_async(g, P->cls, old.cls, 4 * n * IM_MAX_EV, P->stream)); GPU(im_dev_copy_async(g, P->b1, old.b1, 4 * n * IM_MAX_EV, P->stream));
    GPU(im_dev_copy_async(g, P->b2, old.b2, 4 * n * IM_MAX_EV, P->stream));
    GPU(im_stream_sync(g, P->stream));
    pipe_free_cands(&old);
}

static void pipe_submit(ppipe* P, pgroup* G)
{
    pchunk* c = &P->ck[P->cur];
    if (c->n == 0) return;
    c->h_off[c->n] = c->bytes;
    pipe_ensure_capacity(P, G, c->n, c->seq_bytes);
    im_ctx* g = P->d->gpu;
    GPU(im_dev_upload_async(g, c->d_raw, c->h_raw, c->bytes, P->stream));
    GPU(im_dev_upload_async(g, c->d_off, c->h_off, 4 * ((size_t)c->n + 1), P->stream));
    im_dev_records recs = { c->n, c->d_raw, c->d_off, (int32_t)c->rec_base };
    im_dev_cands out;
    memset(&out, 0, sizeof out);
    out.batch.bases = P->bases; out.batch.base_off = P->boff; out.batch.read_len = P->len; out.batch.tid = P->tid;
    out.batch.anchor = P->anchor; out.batch.range_max = P->range; out.batch.out = P->res;
    out.batch.ev_cls = P->cls; out.batch.ev_b1 = P->b1; out.batch.ev_b2 = P->b2;
    out.cand_rec = P->cand_rec; out.counters = P->counters; out.rec_class = c->d_class;
    out.cap_cand = P->cap_cand; out.cap_bases = P->cap_bases; out.consumed = NULL;     /* cleared once per group in pipe_run_group */
    GPU(im_dev_triage(g, &P->tp, &recs, &out, c->d_scratch, c->scratch_bytes, P->stream));
    GPU(im_dev_download_async(g, c->h_cnt, P->counters, 32, P->stream));
    GPU(im_event_record(c->done, P->stream));
    c->busy = 1; P->n_busy++;
    P->fly_recs += c->n; P->fly_seq += c->seq_bytes;
    P->cur = (P->cur + 1) % g_nchunk;
    if (P->ck[P->cur].busy) pipe_harvest(P, G, &P->ck[P->cur]);     /* the ring is full: its oldest chunk comes back first */
    P->ck[P->cur].rec_base = G->n_rec;
}

static void name_log(char** buf, int64_t* len, int64_t* cap, const char* name)
{
    const int64_t l = (int64_t)strlen(name) + 1;
    if (*len + l > *cap) { *cap = (*cap + l) * 2 + 4096; *buf = xrealloc(*buf, (size_t)*cap); }
    memcpy(*buf + *len, name, (size_t)l);
    *len += l;
}

/* the entries a contig leaves waiting in its pair table, by name */
static void group_log_waiting(const driver* d, pgroup* G, gcontig* cg)
{
    cg->sn0 = G->sn_len;
    for (int32_t i = 0; i < d->n_live; i++) name_log(&G->sn, &G->sn_len, &G->sn_cap, d->live[i]->qname);
    cg->sn1 = G->sn_len;
}

/* a record of a not-proper pair through the pair table (src/indelminer.c:516-615); rec = the group's record count with it */
static void host_discordant(driver* d, pgroup* G, const bam_record* b, int64_t rec)
{
    const int32_t* range = record_range(d, b);
    if (abs(b->isize) > range[1] && (uint32_t)abs(b->isize) < O.maxpedelsize && ((b->flag & 0x10) != 0) != ((b->flag & 0x20) != 0))
        name_log(&G->dn, &G->dn_len, &G->dn_cap, BAMR_QNAME(b));            /* it is entered in, or looked up in, the table */
    evidence_t* e = discordant_pair(d, b, range);
    if (e) {
        if (G->n_pe == G->cap_pe) {
            G->cap_pe = G->cap_pe ? G->cap_pe * 2 : 1024;
            G->pe = xrealloc(G->pe, sizeof(evidence_t*) * (size_t)G->cap_pe);
            G->pe_rec = xrealloc(G->pe_rec, sizeof(int64_t) * (size_t)G->cap_pe);
        }
        e->arrival = ((int64_t)G->n_virt + rec - 1) * 8 + 7;
        e->when = ((int64_t)G->seq << 32) | (rec - 1);
        G->pe[G->n_pe] = e; G->pe_rec[G->n_pe] = rec - 1; G->n_pe++;
    }
    if (d->live_changed) {
        /* find_marker (src/indelminer.c:211-233) is a function of the pair table alone: its value is logged where it moves */
        d->live_changed = 0;
        const int m = find_marker_live(d);
        if (G->n_lm == G->ctg[G->cur_ctg].lm0 || G->lm_val[G->n_lm - 1] != m) {
            if (G->n_lm == G->cap_lm) {
                G->cap_lm = G->cap_lm ? G->cap_lm * 2 : 4096;
                G->lm_rec = xrealloc(G->lm_rec, sizeof(int32_t) * (size_t)G->cap_lm);
                G->lm_val = xrealloc(G->lm_val, sizeof(int) * (size_t)G->cap_lm);
            }
            G->lm_rec[G->n_lm] = (int32_t)rec; G->lm_val[G->n_lm] = m; G->n_lm++;
        }
    }
}

/* estimate_insertlengths' share of a record (src/bamoperations.c:15-86): extrema of the insert size per read group, and where
 * the group was first seen (the table lists the groups in file order: its prefix-match look-up depends on that) */
static void host_rg_stat(pgroup* G, const bam_record* b, const uint8_t* rg, int64_t rec_in_contig)
{
    const int flag = b->flag;
    if (!((flag & 0x1) && !(flag & 0x4) && (flag & 0x2) && !(flag & (0x100 | 0x200 | 0x400)) && b->isize >= 0)) return;
    const char* rgname = "generic";
    if (rg) { forceassert(rg[0] == 'Z'); rgname = bam_aux_str(rg); }       /* asserted in front of the two mate-position tests (src/bamoperations.c:37-46) */
    if (!(b->mpos - b->pos >= 0 && b->isize >= b->mpos - b->pos)) return;
    int k = G->n_rgs - 1;                           /* the last group seen first: records of one library come in runs */
    while (k >= 0 && strcmp(G->rgs[k].name, rgname) != 0) k--;
    if (k < 0) {
        if (G->n_rgs == MG_MAX_RG || strlen(rgname) >= sizeof G->rgs[0].name)
            fatalf("at most %d read groups with names under %zu bytes are supported here", MG_MAX_RG, sizeof G->rgs[0].name);
        rgstat_t* n = &G->rgs[G->n_rgs++];
        snprintf(n->name, sizeof n->name, "%s", rgname);
        n->min = n->max = b->isize; n->first_tid = b->tid; n->first_rec = rec_in_contig;
    } else {
        if (G->rgs[k].min > b->isize) G->rgs[k].min = b->isize;
        if (G->rgs[k].max < b->isize) G->rgs[k].max = b->isize;
    }
}

/* the host's share of fetch_func for one record: count it, serve the pair table, log what the flush points need */
/* Reads beyond 255 bases take the realign kernels' later launches (im_expect_read_length, include/indelminer_amd.h): the context
 * hears of the longest read so far the moment a walker meets it, i.e. before the group that holds it is launched. */
static volatile int g_longest_read = 255;
static void note_long_read(driver* d, int l_seq)
{
    if (im_expect_read_length(d->gpu, l_seq) != IM_OK) fatalf("im_expect_read_length: %s", im_last_error(d->gpu));
    int cur = __atomic_load_n(&g_longest_read, __ATOMIC_RELAXED);
    while (l_seq > cur && !__atomic_compare_exchange_n(&g_longest_read, &cur, l_seq, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

#define CN_STRIDE 256
static void cn_log(pgroup* G, int32_t rec, int32_t pos)
{
    const int64_t k = G->n_cn;
    if (k == G->cap_cn) { G->cap_cn = G->cap_cn ? G->cap_cn * 2 : (1 << 20); G->cn_step = xrealloc(G->cn_step, 2 * (size_t)G->cap_cn); }
    const int64_t dr = (int64_t)rec - G->cn_last_rec, dp = (int64_t)pos - G->cn_last_pos;
    const int64_t since = G->n_cnck ? k - G->cn_ck[G->n_cnck - 1].k : CN_STRIDE;
    if (since >= CN_STRIDE || k == G->ctg[G->cur_ctg].cn0 || dr < 0 || dr > 255 || dp < 0 || dp > 255) {
        if (G->n_cnck == G->cap_cnck) { G->cap_cnck = G->cap_cnck ? G->cap_cnck * 2 : 4096; G->cn_ck = xrealloc(G->cn_ck, sizeof *G->cn_ck * (size_t)G->cap_cnck); }
        G->cn_ck[G->n_cnck].k = k; G->cn_ck[G->n_cnck].rec = rec; G->cn_ck[G->n_cnck].pos = pos; G->n_cnck++;
        G->cn_step[2 * k] = 0; G->cn_step[2 * k + 1] = 0;
    } else { G->cn_step[2 * k] = (uint8_t)dr; G->cn_step[2 * k + 1] = (uint8_t)dp; }
    G->cn_last_rec = rec; G->cn_last_pos = pos;
    G->n_cn = k + 1;
}
/* counted read k of the group (0 <= k < n_cn): its record number and position */
static void cn_at(const pgroup* G, int64_t k, int32_t* rec, int32_t* pos)
{
    int64_t lo = 0, hi = G->n_cnck - 1;                 /* the last checkpoint at or in front of k */
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (G->cn_ck[mid].k <= k) lo = mid; else hi = mid - 1; }
    int32_t r = G->cn_ck[lo].rec, p = G->cn_ck[lo].pos;
    for (int64_t j = G->cn_ck[lo].k + 1; j <= k; j++) { r += G->cn_step[2 * j]; p += G->cn_step[2 * j + 1]; }
    *rec = r; *pos = p;
}

/* fetch_func looks the read group of every read it counts up in the insert-length table and dies where one is missing
 * (src/indelminer.c:369-376).  With the table in hand (configuration file, several ranks) a walker does the same at the first read
 * of each group it meets -- its exit hands the run over (walker_bails_out); in one pass the names wait in the group for the table
 * (run_pipeline). */
static void note_counted_rg(driver* d, pgroup* G, const bam_record* b, const uint8_t* rg)
{
    const char* rgname = "generic";
    if (rg) {
        if (rg[0] != 'Z') fatalf("a read group tag of %s is not a string", BAMR_QNAME(b));     /* bam_aux2Z gives NULL, strlen(NULL) follows */
        rgname = bam_aux_str(rg);
    }
    if (G->n_crg && strcmp(G->crg[G->crg_last], rgname) == 0) return;
    int k = 0;
    while (k < G->n_crg && strcmp(G->crg[k], rgname) != 0) k++;
    if (k == G->n_crg) {
        if (!g_onepass && !qhash_lookup(d->insertlengths, rgname, (int)strlen(rgname))) fatalf("did not find %s in the hash", rgname);
        if (G->n_crg == MG_MAX_RG || strlen(rgname) >= sizeof G->crg[0])
            fatalf("at most %d read groups with names under %zu bytes are supported here", MG_MAX_RG, sizeof G->crg[0]);
        snprintf(G->crg[G->n_crg++], sizeof G->crg[0], "%s", rgname);
    }
    G->crg_last = k;
}

static void pipe_host_record(driver* d, pgroup* G, const bam_record* b)
{
    const int flag = b->flag;
    const uint8_t* rg = NULL;               /* the record's RG tag, looked for once */
    if (g_onepass) {
        rg = bam_aux_find(b, "RG");
        host_rg_stat(G, b, rg, ((int64_t)(b->pos < 0 ? 0 : b->pos) << 32) | (G->n_rec - 1 - G->ctg[G->cur_ctg].rec0));
        if (!G->cov.sum) cov_init(&G->cov, d->hdr->n_targets);
        cov_record(&G->cov, b);
    }
    if (flag & (0x100 | 0x200 | 0x400 | 0x800)) return;
    if ((flag & 0x1) == 0) return;
    if (b->l_seq > g_longest_read) note_long_read(d, b->l_seq);
    const int is_aligned = (flag & 0x4) == 0, is_mate_aligned = (flag & 0x8) == 0;
    if (is_aligned && is_mate_aligned && b->tid != b->mtid) return;
    note_counted_rg(d, G, b, g_onepass ? rg : bam_aux_find(b, "RG"));
    if (is_aligned && is_mate_aligned && (flag & 0x2) == 0) {
        /* the pair table (src/indelminer.c:516-615) carries entries from one piece of a contig into the next, and pieces are
         * walked at the same time: the record waits, with its place in the group, for the main thread (group_pair_table) */
        (void)d;
        const int64_t len = (int64_t)b->l_data + 32;
        if (G->npp_len + len > G->npp_cap) { G->npp_cap = (G->npp_cap + len) * 2 + (1 << 16); G->npp_raw = xrealloc(G->npp_raw, (size_t)G->npp_cap); }
        if (G->n_npp == G->cap_npp) {
            G->cap_npp = G->cap_npp ? G->cap_npp * 2 : 4096;
            G->npp_off = xrealloc(G->npp_off, sizeof(int64_t) * ((size_t)G->cap_npp + 1));
            G->npp_rec = xrealloc(G->npp_rec, sizeof(int32_t) * (size_t)G->cap_npp);
        }
        memcpy(G->npp_raw + G->npp_len, b->data - 32, (size_t)len);
        G->npp_off[G->n_npp] = G->npp_len; G->npp_rec[G->n_npp] = (int32_t)G->n_rec; G->n_npp++;
        G->npp_len += len;
        G->npp_off[G->n_npp] = G->npp_len;
    }
    /* a counted read (src/indelminer.c:617): every READCHUNK-th of the whole run is a flush point */
    cn_log(G, (int32_t)G->n_rec, b->pos);
}

static void group_push_flush(pgroup* G, int64_t rec, int32_t pe, int marker, int32_t tid)
{
    if (G->n_fl == G->cap_fl) { G->cap_fl = G->cap_fl ? G->cap_fl * 2 : 256; G->fl = xrealloc(G->fl, sizeof(gflush) * (size_t)G->cap_fl); }
    gflush* f = &G->fl[G->n_fl++];
    f->rec = rec; f->pe = pe; f->marker = marker; f->tid = tid;
}

/* Where the READCHUNK flushes of a walked group fall (src/indelminer.c:617): every READCHUNK-th counted read of the RUN, found from
 * the walk's log of counted reads once the run's read counter in front of the group is known -- *numread, advanced past the
 * group.  (A multi-GPU run knows the counter in front of every piece from the exchanged logs: there the rank that walked the
 * piece does this and only the points travel.) */
static void group_flush_points(pgroup* G, int64_t* numread)
{
    G->n_fp = 0;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        gcontig* cg = &G->ctg[ci];
        if (g_mg) *numread = g_mg->piece_prefix[cg->piece];
        cg->fp0 = G->n_fp;
        const int64_t ncount = cg->cn1 - cg->cn0;
        cg->n_counted = ncount;
        /* the k-th counted read of the piece (k from 0) is read number *numread + k + 1 of the run */
        for (int64_t k = (READCHUNK - 1 - (*numread % READCHUNK)) % READCHUNK; k < ncount; k += READCHUNK) {
            if (G->n_fp == G->cap_fp) { G->cap_fp = G->cap_fp ? G->cap_fp * 2 : 64; G->fp = xrealloc(G->fp, sizeof(gfpoint) * (size_t)G->cap_fp); }
            int32_t rec_k, pos_k;
            cn_at(G, cg->cn0 + k, &rec_k, &pos_k);
            G->fp[G->n_fp].rec = rec_k; G->fp[G->n_fp].pos = pos_k; G->n_fp++;
            timestamp("Read %ld reads", (long)(*numread + k + 1));
        }
        cg->fp1 = G->n_fp;
        *numread += ncount;
    }
}

/* The flushes themselves (src/indelminer.c:617-670, 806-823), piece by piece, from the flush points and the pair table's log:
 * *floor = the smallest start among the pair-table entries that CONTIGS before this one left waiting (the reference never
 * removes those, so find_marker keeps seeing them), advanced past the group.  In a multi-GPU run it comes per contig from the
 * exchanged logs. */
static void group_resolve_flushes(pgroup* G, int* floor)
{
    G->n_fl = 0;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        gcontig* cg = &G->ctg[ci];
        if (g_mg && cg->first) *floor = g_mg->floor[cg->tid];
        cg->fl0 = G->n_fl;
        int32_t lm = cg->lm0, pe = cg->pe0;
        int live_min = cg->lm_init;
        for (int32_t k = cg->fp0; k < cg->fp1; k++) {
            const int64_t rec = G->fp[k].rec;
            while (lm < cg->lm1 && G->lm_rec[lm] <= rec) live_min = G->lm_val[lm++];
            while (pe < cg->pe1 && G->pe_rec[pe] < rec) pe++;
            int marker = live_min;
            if (*floor < marker) marker = *floor;
            if (G->fp[k].pos < marker) marker = G->fp[k].pos;
            group_push_flush(G, rec, pe, marker, cg->tid);
        }
        /* end of contig (src/indelminer.c:806-823): everything still pending is consumed */
        if (cg->last) group_push_flush(G, cg->rec1, cg->pe1, INT_MAX, cg->tid);
        cg->fl1 = G->n_fl;
        if (cg->last && cg->left_min < *floor) *floor = cg->left_min;
    }
}

/* Main thread, groups in contig order: does a record of this group go through the pair table under the name of an entry an
 * EARLIER contig left waiting?  Then the reference's one table pairs them (or hands the old entry to the new pair's second mate:
 * its look-up takes the oldest entry of a name, src/hashtable.c:62-81) and the contigs are not independent.  Returns 1 if so. */
static qhash* g_run_waiting = NULL;
static int group_meets_earlier_contigs(const pgroup* G)
{
    if (!g_run_waiting) g_run_waiting = qhash_new(12);
    for (int ci = 0; ci < G->n_ctg; ci++) {
        const gcontig* cg = &G->ctg[ci];
        for (int64_t at = cg->dn0; at < cg->dn1; at += (int64_t)strlen(G->dn + at) + 1)
            if (qhash_lookup(g_run_waiting, G->dn + at, (int)strlen(G->dn + at) + 1)) return 1;
        for (int64_t at = cg->sn0; at < cg->sn1; at += (int64_t)strlen(G->sn + at) + 1)
            qhash_add(g_run_waiting, G->sn + at, (int)strlen(G->sn + at) + 1, NULL);
    }
    return 0;
}

static void pipe_walk_piece(ppipe* P, pgroup* G, const piece_t* pc, bgzf_reader* r)
{
    driver* d = P->d;
    const int32_t tid = pc->tid;
    if (G->n_ctg == G->cap_ctg) { G->cap_ctg = G->cap_ctg ? G->cap_ctg * 2 : 32; G->ctg = xrealloc(G->ctg, sizeof(gcontig) * (size_t)G->cap_ctg); }
    gcontig* cg = &G->ctg[G->n_ctg++];
    memset(cg, 0, sizeof *cg);
    cg->tid = tid; cg->rec0 = G->n_rec; cg->pe0 = G->n_pe; cg->fl0 = cg->fl1 = 0;
    cg->cn0 = G->n_cn; cg->lm0 = cg->lm1 = G->n_lm; cg->dn0 = cg->dn1 = G->dn_len; cg->sn0 = cg->sn1 = G->sn_len;
    cg->beg = pc->beg; cg->end = pc->end; cg->first = pc->first; cg->last = pc->last; cg->lm_init = INT_MAX; cg->left_min = INT_MAX;
    cg->piece = pc->index;
    G->cur_ctg = G->n_ctg - 1;
    bam_region_iter it;
    if ((pc->overlap ? bam_region_begin(&it, r, d->idx, tid, pc->beg, pc->end) : bam_piece_begin(&it, r, d->idx, tid, pc->beg, pc->end)) != 0) fatalf("cannot seek in %s", d->bam_name);
    /* the records travel without their base qualities (half their bytes; nothing on the path reads them): INDELMINER_KEEP_QUAL=1 keeps them */
    it.drop_qual = !getenv("INDELMINER_KEEP_QUAL");
    bam_record b; memset(&b, 0, sizeof b);
    for (;;) {
        pchunk* c = &P->ck[P->cur];
        if (c->n == 0) c->rec_base = G->n_rec;
        int32_t len = 0;
        const int rc = (c->n < (int32_t)PIPE_CHUNK_RECS)
            ? bam_region_next_raw(&it, c->h_raw + c->bytes, (int64_t)PIPE_CHUNK_BYTES - c->bytes, &len, &b) : -2;
        if (rc == -2) {
            if (c->n == 0) fatalf("a BAM record larger than %u bytes", PIPE_CHUNK_BYTES);
            DEV_TIMED(pipe_submit(P, G));
            continue;
        }
        if (rc < 0) fatalf("error while reading %s", d->bam_name);
        if (rc == 0) break;
        c->h_off[c->n++] = c->bytes;
        for (uint32_t z = (uint32_t)len; z & 3u; z++) c->h_raw[c->bytes + z] = 0;
        c->bytes += ((uint32_t)len + 3u) & ~3u;
        c->seq_bytes += ((int64_t)b.l_seq + 3) & ~(int64_t)3;
        G->n_rec++;
        if (G->n_rec >= 0x7fffffff) fatalf("more than 2^31 records in one group of pieces");
        pipe_host_record(d, G, &b);
    }
    cg = &G->ctg[G->n_ctg - 1];
    cg->rec1 = G->n_rec; cg->pe1 = G->n_pe; cg->cn1 = G->n_cn;
    /* the piece's last records go out now */
    DEV_TIMED(pipe_submit(P, G));
}

static int g_tie_for_sort;
static const int32_t* g_key_for_sort;
static int cmp_cluster_idx(const void* x, const void* y)
{
    const int32_t* a = g_key_for_sort + 4 * (size_t)*(const int32_t*)x;
    const int32_t* b = g_key_for_sort + 4 * (size_t)*(const int32_t*)y;
    if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;          /* flush */
    if (a[2] != b[2]) return a[2] < b[2] ? -1 : 1;          /* b1 */
    if (a[3] != b[3]) return a[3] < b[3] ? -1 : 1;          /* b2 */
    if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;          /* class */
    return 0;
}

/* THE STAGE of a walked group, on the main thread's pipeline S: in front the candidates earlier pieces of the contig left pending
 * (their slots as they were left), behind them the group's own candidates out of their parked arrays; realign of the own
 * candidates, the flush list and the group-by over all of them, results to the host.  Record numbers of the stage: the
 * front items 0 .. n_virt - 1 in order of arrival, the group's own records from n_virt on. */
static void stage_run_group(ppipe* P, pgroup* G)
{
    driver* d = P->d;
    im_ctx* g = d->gpu;
    const int32_t K = G->n_front, n_own = G->sv_n, nc = K + n_own;
    /* room: candidates, read bytes, paired-read entries, flushes */
    {
        int32_t need_c = nc > P->cap_cand ? nc : P->cap_cand; int64_t need_b = G->sv_bytes + 64 > P->cap_bases ? G->sv_bytes + 64 : P->cap_bases;
        int32_t need_pe = G->n_pe > P->cap_pe ? G->n_pe : P->cap_pe;
        int grow = 0;
        if (G->n_fl > P->cap_fl) {
            GPU(im_stream_sync(g, P->stream));
            im_dev_free(g, P->cut); im_dev_free(g, P->fdesc);
            while (P->cap_fl < G->n_fl) P->cap_fl *= 2;
            P->cut = pdev_alloc(P, 8 * (size_t)P->cap_fl); P->fdesc = pdev_alloc(P, sizeof(im_flush_desc) * (size_t)P->cap_fl);
            grow = 1;
        }
        if (need_c > P->cap_cand || need_b > P->cap_bases || need_pe > P->cap_pe || grow) {
            int32_t c2 = P->cap_cand, p2 = P->cap_pe; int64_t b2 = P->cap_bases;
            while (c2 < need_c) c2 *= 2;
            while (b2 < need_b) b2 *= 2;
            while (p2 < need_pe) p2 *= 2;
            GPU(im_stream_sync(g, P->stream));
            pipe_free_cands(P);
            pipe_alloc_cands(P, c2, b2, p2);
        }
    }
    const size_t pe_base = (size_t)P->cap_cand * IM_MAX_EV;
    const size_t nK = (size_t)K, nO = (size_t)n_own;
    /* the group's own arrays, behind the front */
    {
        const size_t bytes[10] = { (size_t)G->sv_bytes, 8 * nO, 4 * nO, 4 * nO, 4 * nO, 4 * nO, 4 * nO * IM_MAX_EV, 4 * nO * IM_MAX_EV, 4 * nO * IM_MAX_EV, 4 * nO };
        void* dst[10] = { P->bases, (char*)P->boff + 8 * nK, (char*)P->len + 4 * nK, (char*)P->tid + 4 * nK, (char*)P->anchor + 4 * nK, NULL,
                          (char*)P->cls + 16 * nK, (char*)P->b1 + 16 * nK, (char*)P->b2 + 16 * nK, (char*)P->range + 4 * nK };
        for (int k = 0; k < 10; k++) if (dst[k] && bytes[k] && G->sv[k]) GPU(im_dev_copy_async(g, dst[k], G->sv[k], bytes[k], P->stream));
        if (G->sv_range && n_own) GPU(im_dev_upload_async(g, (char*)P->range + 4 * nK, G->sv_range, 4 * nO, P->stream));    /* one-pass: known only now */
    }
    /* What the host adds -- record numbers (the front's, then the own ones counted on from n_virt), the front's slots, the
     * counters, the paired-read entries, the flush list -- is put together in the pinned block and goes up on the stage's stream
     * (the block is not touched again before the stage's wait at the end). */
    const size_t up_rec = 0, up_cls = up_rec + ((4 * ((size_t)nc + 1) + 255) & ~(size_t)255), up_b1 = up_cls + 16 * nK, up_b2 = up_b1 + 16 * nK,
                 up_cnt = (up_b2 + 16 * nK + 255) & ~(size_t)255, up_pe = up_cnt + 256, up_fd = (up_pe + 12 * (size_t)G->n_pe + 255) & ~(size_t)255,
                 up_end = up_fd + sizeof(im_flush_desc) * (size_t)(G->n_fl ? G->n_fl : 1);
    {
        if (!P->h_stage) { P->h_stage_cap = 1 << 20; GPU(im_host_alloc(g, P->h_stage_cap, &P->h_stage)); }
        if (up_end > P->h_stage_cap) {
            GPU(im_stream_sync(g, P->stream));
            im_host_free(g, P->h_stage);
            while (P->h_stage_cap < up_end) P->h_stage_cap *= 2;
            GPU(im_host_alloc(g, P->h_stage_cap, &P->h_stage));
        }
        char* hb = P->h_stage;
        int32_t* t = (int32_t*)(hb + up_rec);
        for (int32_t q = 0; q < K; q++) t[q] = G->front_virt[q];
        for (int32_t i = 0; i < n_own; i++) t[K + i] = G->cand_rec[i] + G->n_virt;
        if (nc) GPU(im_dev_upload_async(g, P->cand_rec, t, 4 * (size_t)nc, P->stream));
        if (K) {
            int32_t *c = (int32_t*)(hb + up_cls), *x1 = (int32_t*)(hb + up_b1), *x2 = (int32_t*)(hb + up_b2);
            for (int32_t q = 0; q < K; q++)
                for (int k = 0; k < IM_MAX_EV; k++) { c[q * IM_MAX_EV + k] = G->front[q].cls[k]; x1[q * IM_MAX_EV + k] = G->front[q].b1[k]; x2[q * IM_MAX_EV + k] = G->front[q].b2[k]; }
            GPU(im_dev_upload_async(g, P->cls, c, 16 * nK, P->stream)); GPU(im_dev_upload_async(g, P->b1, x1, 16 * nK, P->stream));
            GPU(im_dev_upload_async(g, P->b2, x2, 16 * nK, P->stream));
        }
        int32_t* cnt = (int32_t*)(hb + up_cnt);
        memset(cnt, 0, 64);
        cnt[0] = nc;
        GPU(im_dev_upload_async(g, P->counters, cnt, 64, P->stream));
        if (G->n_pe > 0) {
            /* paired-read entries (class 2) behind the split-read slots: pending ones of earlier pieces first; an entry without an
             * evidence object stands for the entries that wait for the contig's end (stage_leftovers) and carries their smallest key */
            int32_t* e = (int32_t*)(hb + up_pe);
            for (int32_t i = 0; i < G->n_pe; i++) { e[i] = 2; e[G->n_pe + i] = G->pe[i]->b1; e[2 * (size_t)G->n_pe + i] = G->pe[i]->b2; }
            GPU(im_dev_upload_async(g, (char*)P->cls + 4 * pe_base, e, 4 * (size_t)G->n_pe, P->stream));
            GPU(im_dev_upload_async(g, (char*)P->b1 + 4 * pe_base, e + G->n_pe, 4 * (size_t)G->n_pe, P->stream));
            GPU(im_dev_upload_async(g, (char*)P->b2 + 4 * pe_base, e + 2 * (size_t)G->n_pe, 4 * (size_t)G->n_pe, P->stream));
        }
    }
    /* The flush list of the group, in file order, and the split-read group-by.  Within a contig the markers never decrease
     * (find_marker is a minimum over pair-table entries that leave the table or enter it at the current position of a
     * coordinate-sorted walk), so which flush consumes an entry needs no history: three chip-wide launches do the whole
     * list and the group-by (im_dev_flush_groupby).  A BAM whose positions run backwards inside a contig can break that;
     * such a group takes the sequential forms: one workgroup walking the list, or one launch pair per flush when no
     * mid-contig flush consumes anything and the pending ranges grow long. */
    im_flush_desc* fd = (im_flush_desc*)((char*)P->h_stage + up_fd);
    memset(fd, 0, sizeof(im_flush_desc) * (size_t)(G->n_fl ? G->n_fl : 1));
    int64_t longest = 0;
    int monotone = 1;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        const gcontig* cg = &G->ctg[ci];
        for (int f = cg->fl0; f < cg->fl1; f++) {
            const gflush* fl = &G->fl[f];
            /* a piece that continues a contig is the only one of its group: its flushes see the front from record 0 on */
            fd[f].rec0 = G->n_virt ? 0 : (int32_t)cg->rec0; fd[f].rec1 = (int32_t)fl->rec + G->n_virt; fd[f].pe0 = cg->pe0; fd[f].pe1 = fl->pe;
            fd[f].marker = fl->marker; fd[f].id = f + 1; fd[f].last = cg->fl1 - 1;
            if (fl->rec - cg->rec0 > longest) longest = fl->rec - cg->rec0;
            if (f > cg->fl0 && (fl->marker < G->fl[f - 1].marker || fl->rec < G->fl[f - 1].rec || fl->pe < G->fl[f - 1].pe)) monotone = 0;
        }
    }
    const char* fm = getenv("INDELMINER_FLUSH_MODE");
    const int wide = fm ? strcmp(fm, "wide") == 0 : monotone;
    if (wide && !monotone) fatalf("INDELMINER_FLUSH_MODE=wide: the flush markers of a contig decrease (is %s coordinate-sorted?)", d->bam_name);
    const int per_flush = !wide && (fm && strcmp(fm, "seq") != 0 ? strcmp(fm, "per-flush") == 0 : (longest > 16 * (int64_t)READCHUNK && G->n_fl > 64));
    if (!wide) {
        GPU(im_dev_memset(g, P->consumed, 0, 4 * (pe_base + (size_t)G->n_pe), P->stream));
        if (per_flush) GPU(im_dev_memset(g, P->cut, 0xFF, 8 * (size_t)G->n_fl, P->stream));
    }
    im_params prm = { O.klength, O.numgaps, O.maxdelsize, O.ethreshold };
    if (n_own > 0) {
        im_dev_batch bt;
        memset(&bt, 0, sizeof bt);
        bt.n = n_own; bt.bases = P->bases; bt.base_off = (const int64_t*)P->boff + nK; bt.read_len = (const int32_t*)P->len + nK; bt.tid = (const int32_t*)P->tid + nK;
        bt.anchor = (const int32_t*)P->anchor + nK; bt.range_max = (const int32_t*)P->range + nK; bt.out = (im_read_result*)P->res + nK;
        bt.ev_cls = (int32_t*)P->cls + nK * IM_MAX_EV; bt.ev_b1 = (int32_t*)P->b1 + nK * IM_MAX_EV; bt.ev_b2 = (int32_t*)P->b2 + nK * IM_MAX_EV;
        GPU(im_dev_realign_keep(g, &prm, &bt, P->stream));
    }
    if (!per_flush && G->n_fl) GPU(im_dev_upload_async(g, P->fdesc, fd, sizeof(im_flush_desc) * (size_t)G->n_fl, P->stream));
    if (wide) {
        GPU(im_dev_flush_groupby(g, (const im_flush_desc*)P->fdesc, G->n_fl, P->cls, P->b1, P->b2, P->consumed, P->cand_rec, P->counters, nc,
                                 (int32_t)pe_base, G->n_pe, O.tie_desc, P->order, P->clkey, P->clfirst, P->clcount, P->counts,
                                 P->fgscratch, P->fgscratch_bytes, P->stream));
    } else {
        if (!per_flush) {
            GPU(im_dev_flush_cuts(g, (const im_flush_desc*)P->fdesc, G->n_fl, P->cls, P->b1, P->b2, P->consumed,
                                  P->cand_rec, P->counters, P->cap_cand, (int32_t)pe_base, G->n_pe, P->stream));
        } else {
            for (int f = 0; f < G->n_fl; f++)
                GPU(im_dev_flush_cut_rec(g, P->cls, P->b1, P->b2, P->consumed, fd[f].rec0, fd[f].rec1, P->cand_rec, P->counters, P->cap_cand,
                                         (int32_t)pe_base + fd[f].pe0, (int32_t)pe_base + fd[f].pe1, fd[f].marker, fd[f].id,
                                         (uint64_t*)P->cut + f, P->stream));
        }
        GPU(im_dev_cluster_groupby(g, nc * IM_MAX_EV, P->cls, P->b1, P->b2, P->consumed, O.tie_desc,
                                   P->order, P->clkey, P->clfirst, P->clcount, P->counts, P->gscratch, P->gscratch_bytes, P->stream));
    }
    if (n_own > 0) {
        /* only the realigned records that hold evidence travel whole (im_dev_compact_results) */
        if (n_own > P->cap_rc) {
            GPU(im_stream_sync(g, P->stream));
            if (P->rstat) { im_dev_free(g, P->rstat); im_dev_free(g, P->rslot); im_dev_free(g, P->rcompact); }
            if (!P->rcount) P->rcount = pdev_alloc(P, 256);
            P->cap_rc = P->cap_cand > n_own ? P->cap_cand : n_own;
            P->rstat = pdev_alloc(P, 4 * (size_t)P->cap_rc); P->rslot = pdev_alloc(P, 4 * (size_t)P->cap_rc);
            P->rcompact = pdev_alloc(P, sizeof(im_read_result) * (size_t)P->cap_rc);
        }
        GPU(im_dev_compact_results(g, (const im_read_result*)P->res + nK, n_own, NULL, P->rstat, P->rslot, P->rcompact, P->rcount, P->stream));
    }
    GPU(im_stream_sync(g, P->stream));
    phase_time("device: realign + flush cuts + group-by");

    const size_t nn = (size_t)(nc ? nc : 1);
    /* Results to the host: the counts first, then every array with ONE asynchronous copy into a pinned block and one wait -- a
     * plain hipMemcpy into freshly allocated pageable memory pins and un-pins the destination per call (a few hundred
     * microseconds each; a dozen arrays per group, hundreds of groups). */
    int32_t n_evd = 0;
    int32_t counts[2] = { 0, 0 };
    {
        if (!P->h_stage) { P->h_stage_cap = 1 << 20; GPU(im_host_alloc(g, P->h_stage_cap, &P->h_stage)); }
        int32_t* hc = P->h_stage;
        hc[0] = 0;
        if (n_own > 0) GPU(im_dev_download_async(g, hc, P->rcount, 4, P->stream));
        GPU(im_dev_download_async(g, hc + 2, P->counts, 8, P->stream));
        GPU(im_stream_sync(g, P->stream));
        n_evd = hc[0]; counts[0] = hc[2]; counts[1] = hc[3];
    }
    G->n_cl = counts[0]; G->n_nodes = counts[1];
    int32_t* rstat = xmalloc(4 * (size_t)(n_own ? n_own : 1));
    G->res = xrealloc(G->res, sizeof(im_read_result) * (size_t)(n_evd ? n_evd : 1));
    G->res_slot = xrealloc(G->res_slot, 4 * (size_t)(n_own ? n_own : 1));
    G->s_cls = xrealloc(G->s_cls, 4 * nn * IM_MAX_EV); G->s_b1 = xrealloc(G->s_b1, 4 * nn * IM_MAX_EV); G->s_b2 = xrealloc(G->s_b2, 4 * nn * IM_MAX_EV);
    G->cons_sr = xrealloc(G->cons_sr, 4 * nn * IM_MAX_EV);
    G->cons_pe = xrealloc(G->cons_pe, 4 * (size_t)(G->n_pe ? G->n_pe : 1));
    G->cl_key = xrealloc(G->cl_key, 16 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_first = xrealloc(G->cl_first, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_count = xrealloc(G->cl_count, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_sorted = xrealloc(G->cl_sorted, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->order = xrealloc(G->order, 4 * (size_t)(G->n_nodes ? G->n_nodes : 1));
    {
        const size_t nslb = 4 * (size_t)nc * IM_MAX_EV, ncl = (size_t)G->n_cl;
        struct { void* dst; const void* src; size_t bytes; } job[13] = {
            { rstat, P->rstat, n_own > 0 ? 4 * nO : 0 }, { G->res_slot, P->rslot, n_own > 0 ? 4 * nO : 0 },
            { G->res, P->rcompact, sizeof(im_read_result) * (size_t)n_evd },
            { G->s_cls, P->cls, nslb }, { G->s_b1, P->b1, nslb }, { G->s_b2, P->b2, nslb }, { G->cons_sr, P->consumed, nslb },
            { G->cons_pe, (char*)P->consumed + 4 * pe_base, 4 * (size_t)G->n_pe },
            { G->cl_key, P->clkey, 16 * ncl }, { G->cl_first, P->clfirst, 4 * ncl }, { G->cl_count, P->clcount, 4 * ncl },
            { G->order, P->order, ncl ? 4 * (size_t)G->n_nodes : 0 }, { NULL, NULL, 0 } };
        /* a large array goes straight to its destination (one pin + un-pin of the destination is cheaper than a second pass
         * over megabytes of freshly allocated memory); the many small ones share the pinned block */
        for (int k = 0; job[k].dst; k++)
            if (job[k].bytes > ((size_t)256 << 10)) { GPU(im_dev_download(g, job[k].dst, job[k].src, job[k].bytes)); job[k].bytes = 0; }
        size_t total = 0;
        for (int k = 0; job[k].dst; k++) total += (job[k].bytes + 255) & ~(size_t)255;
        if (total > P->h_stage_cap) {
            im_host_free(g, P->h_stage);
            while (P->h_stage_cap < total) P->h_stage_cap *= 2;
            GPU(im_host_alloc(g, P->h_stage_cap, &P->h_stage));
        }
        size_t at = 0;
        for (int k = 0; job[k].dst; k++) {
            if (job[k].bytes) GPU(im_dev_download_async(g, (char*)P->h_stage + at, job[k].src, job[k].bytes, P->stream));
            at += (job[k].bytes + 255) & ~(size_t)255;
        }
        GPU(im_stream_sync(g, P->stream));
        at = 0;
        for (int k = 0; job[k].dst; k++) {
            if (job[k].bytes) memcpy(job[k].dst, (char*)P->h_stage + at, job[k].bytes);
            at += (job[k].bytes + 255) & ~(size_t)255;
        }
    }
    /* the device groups; the host puts the few clusters in (flush, b1, b2, class) order */
    for (int32_t i = 0; i < G->n_cl; i++) G->cl_sorted[i] = i;
    g_key_for_sort = G->cl_key;
    qsort(G->cl_sorted, (size_t)G->n_cl, sizeof(int32_t), cmp_cluster_idx);
    G->ev_cache = xrealloc(G->ev_cache, sizeof(evidence_t*) * nn * IM_MAX_EV);
    memset(G->ev_cache, 0, sizeof(evidence_t*) * nn * IM_MAX_EV);
    for (int32_t i = 0; i < n_own; i++) {
        const int st = rstat[i];
        if (st >= 0) continue;
        if ((st == IM_ST_ABORT || st == IM_ST_OVERFLOW || st == IM_ST_UNSUPPORTED) && g_handoff_pool) pipeline_handoff();
        if (st == IM_ST_ABORT) fatalf("im_dev_realign: read %d: the reference would abort on this input", i);
        if (st == IM_ST_OVERFLOW) fatalf("im_dev_realign: read %d: segment list longer than IM_MAX_OPS", i);
        if (st == IM_ST_UNSUPPORTED) fatalf("im_dev_realign: read %d: left unrealigned (the context was not told of its length)", i);
    }
    free(rstat);
    if (G->sv[0]) { if (getenv("INDELMINER_TIDY_EXIT") || g_free_slabs) im_dev_free(g, G->sv[0]); G->sv[0] = NULL; }
    phase_time("results to the host");
}

/* the evidence objects of candidate `cand` of group S (its realigned record, its BAM record) into slot[0 .. IM_MAX_EV): the realigned
 * segments when the device found any (they replace the CIGAR-derived ones, src/indelminer.c:494-502), else the CIGAR-derived.
 * arrival0 = the candidate's place in the order of arrival of the stage that asks. */
static void candidate_evidence(driver* d, const pgroup* S, int32_t cand, int64_t arrival0, evidence_t** slot)
{
    bam_record b;
    bam_record_view(S->craw + S->craw_off[cand], (int32_t)(S->craw_off[cand + 1] - S->craw_off[cand]), &b);
    const int flag = b.flag;
    const int is_aligned = (flag & 0x4) == 0, is_rc = (flag & 0x10) != 0, is_mate_rc = (flag & 0x20) != 0;
    const char* qname = BAMR_QNAME(&b);
    const im_read_result* r = S->res_slot[cand] >= 0 ? &S->res[S->res_slot[cand]] : NULL;
    if (r) {                                        /* status == IM_ST_EVIDENCE, n_ev > 0 */
        /* the read as it was realigned (src/indelminer.c:388-455: the mate's strand decides) -- when something will read it */
        int want_bases = !g_lean_evidence;
        for (int k = 0; k < r->n_ev && k < IM_MAX_EV && !want_bases; k++) want_bases = CIG_OP(((const uint32_t*)r->ops)[r->ev[k].seg]) != OP_D;
        char* bases = want_bases ? decode_bases(&b) : NULL;
        char strand = is_rc ? '-' : '+';
        uint8_t qual;
        if (!is_aligned) {
            qual = (uint8_t)mate_mapq(&b, 1);
            if (!is_mate_rc) { if (bases) revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
        } else {
            qual = b.mapq;
            if (is_rc == is_mate_rc) { if (bases) revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
        }
        seglist whole;
        whole.ref_start = r->ref_start; whole.n = r->n_ops; whole.ops = (uint32_t*)r->ops; whole.bases = bases;
        for (int k = 0; k < r->n_ev && k < IM_MAX_EV; k++) {
            const im_evidence* ge = &r->ev[k];
            evidence_t* e = xcalloc(1, sizeof *e);
            e->type = EV_SPLIT_READ; e->cls = ge->cls; e->strand = strand; e->qual = qual;
            e->qname = g_lean_evidence ? NULL : xstrdup(qname);
            e->aln = seglist_copy_for_evidence(&whole, ge->seg);
            e->seg = ge->seg; e->b1 = ge->b1; e->b2 = ge->b2;
            e->lflank = ge->lflank; e->rflank = ge->rflank; e->nd_print = ge->nd_print; e->nd_filter = ge->nd_filter;
            e->arrival = arrival0 + k;
            slot[k] = e;
        }
        free(bases);
    } else if (is_aligned) {
        seglist rln = seglist_from_record(&b);
        evidence_t** bwa = xmalloc(sizeof(evidence_t*) * (size_t)(rln.n ? rln.n : 1));
        const int n = check_variants(&rln, is_rc ? '-' : '+', b.mapq, qname, d->sequences[b.tid], bwa);
        forceassert(n <= IM_MAX_EV);
        for (int k = 0; k < n; k++) { bwa[k]->arrival = arrival0 + k; slot[k] = bwa[k]; }
        free(bwa);
        seglist_free(&rln);
    }
}

/* stage candidate q of group G: one of the front (the candidate of an earlier piece; only its pending slots count) or an own one */
static void group_candidate_evidence(driver* d, pgroup* G, int32_t q)
{
    evidence_t** slot = &G->ev_cache[(size_t)q * IM_MAX_EV];
    if (q < G->n_front) {
        const carry_item* it = &G->front[q];
        candidate_evidence(d, it->g, it->cand, (int64_t)G->front_virt[q] * 8, slot);
        for (int k = 0; k < IM_MAX_EV; k++) if (slot[k] && it->cls[k] < 0) { evidence_free(slot[k]); slot[k] = NULL; }      /* consumed by an earlier piece's flush */
    } else {
        const int32_t cand = q - G->n_front;
        candidate_evidence(d, G, cand, ((int64_t)G->cand_rec[cand] + G->n_virt) * 8, slot);
    }
}

static evidence_t* group_sr_evidence(driver* d, pgroup* G, int32_t slot)
{
    if (!G->ev_cache[slot]) group_candidate_evidence(d, G, slot / IM_MAX_EV);
    if (G->ev_cache[slot] == NULL) {
        const int32_t q = slot / IM_MAX_EV;
        fatalf("internal: the device names evidence slot %d of stage candidate %d (%d in front, %d own, %ld records; device class %d) "
               "but the host finds no evidence there", slot % IM_MAX_EV, q, G->n_front, G->n_cand, (long)G->n_rec, G->s_cls[slot]);
    }
    return G->ev_cache[slot];
}

/* position of evidence in process_evidence's sorted list, as a comparison (src/evidence.c:50-58 + the stable
 * sort of a prepend list): (b1, b2), then newest first -- oldest first with tie_desc */
static int sorted_before(int32_t a1, int32_t a2, int64_t aarr, int32_t b1, int32_t b2, int64_t barr)
{
    if (a1 != b1) return a1 < b1;
    if (a2 != b2) return a2 < b2;
    return O.tie_desc ? aarr < barr : aarr > barr;
}

static int cmp_pe_sorted(const void* x, const void* y)
{
    const evidence_t* a = *(evidence_t* const*)x; const evidence_t* b = *(evidence_t* const*)y;
    if (a->b1 != b->b1) return a->b1 < b->b1 ? -1 : 1;
    if (a->b2 != b->b2) return a->b2 < b->b2 ? -1 : 1;
    if (a->arrival == b->arrival) return 0;
    if (g_tie_for_sort) return a->arrival < b->arrival ? -1 : 1;
    return a->arrival > b->arrival ? -1 : 1;
}

/* process_evidence (src/indelminer.c:117-209) for flush f of the group: the nodes are what the device
 * marked with this flush's id; split-read components are the device's clusters, paired-read components
 * are made here (src/graph.c:100-121) */
static void group_process_flush(driver* d, pgroup* G, const gcontig* cg, int f, int32_t* cl_cursor,
                                variant_list* out, evidence_t*** used_out, int64_t* n_used_out)
{
    variant_list vars = {0};
    const int id = f + 1;
    int64_t n_used = 0, cap_used = 64;
    evidence_t** used = xmalloc(sizeof(evidence_t*) * (size_t)cap_used);
#define USED_PUSH(e) do { if (n_used == cap_used) { cap_used *= 2; used = xrealloc(used, sizeof(evidence_t*) * (size_t)cap_used); } used[n_used++] = (e); } while (0)
    while (*cl_cursor < G->n_cl && G->cl_key[4 * (size_t)G->cl_sorted[*cl_cursor]] == id) {
        const int32_t c = G->cl_sorted[(*cl_cursor)++];
        const int32_t* key = G->cl_key + 4 * (size_t)c;
        const int32_t first = G->cl_first[c], cnt = G->cl_count[c];
        variant_t* v = xcalloc(1, sizeof *v);
        v->type = key[1]; v->evdnctype = EV_SPLIT_READ; v->tid = cg->tid;
        v->start = (uint32_t)key[2]; v->stop = (uint32_t)key[3]; v->support = (uint32_t)cnt;
        v->evidence = xmalloc(sizeof(evidence_t*) * (size_t)cnt);
        int64_t rep = -1;
        for (int32_t k = 0; k < cnt; k++) {
            evidence_t* e = group_sr_evidence(d, G, G->order[first + k]);
            G->ev_cache[G->order[first + k]] = NULL;        /* the flush owns it now (freed with the flush's evidence) */
            v->evidence[k] = e;
            USED_PUSH(e);
            /* the member with the largest sorted position: oldest arrival, newest with tie_desc */
            if (rep < 0 || (O.tie_desc ? e->arrival > rep : e->arrival < rep)) rep = e->arrival;
        }
        v->rep_b1 = key[2]; v->rep_b2 = key[3]; v->rep_arrival = rep;
        if (v->start <= v->stop) vl_push(&vars, v); else variant_free(v);
    }
    /* paired-read nodes of this flush, in sorted order */
    int npe = 0;
    for (int32_t i = cg->pe0; i < G->fl[f].pe; i++) if (G->cons_pe[i] == id && G->pe[i]->type != EV_PHANTOM) npe++;
    if (npe > 0) {
        evidence_t** pe = xmalloc(sizeof(evidence_t*) * (size_t)npe);
        int m = 0;
        for (int32_t i = cg->pe0; i < G->fl[f].pe; i++) if (G->cons_pe[i] == id && G->pe[i]->type != EV_PHANTOM) { pe[m++] = G->pe[i]; USED_PUSH(G->pe[i]); }
        g_tie_for_sort = O.tie_desc;
        qsort(pe, (size_t)npe, sizeof(evidence_t*), cmp_pe_sorted);
        int* parent = xmalloc(sizeof(int) * (size_t)npe);
        for (int i = 0; i < npe; i++) parent[i] = i;
        /* add_node compares every pair (src/graph.c:94-121); an edge needs d2 = (e1.b1 - start of e2's first read) + ... < e2.max,
         * and that first term alone is already >= e1.b1 - e2.b1: in (b1)-sorted order the partners of e1 lie within the largest
         * insert-length bound below it, so the sweep stops there -- same edges, same components, without the N^2 */
        int32_t widest = 0;
        for (int j = 0; j < npe; j++) if (pe[j]->max > widest) widest = pe[j]->max;
        for (int j = 0; j < npe; j++) {
            const evidence_t* e1 = pe[j];
            for (int i = j - 1; i >= 0; i--) {
                const evidence_t* e2 = pe[i];
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
                evidence_t* e = pe[t];
                v->evidence[v->support++] = e;
                if (left == -1 || e->b1 > left) left = e->b1;
                if (right == -1 || e->b2 < right) right = e->b2;
            }
            const evidence_t* e0 = v->evidence[0];
            v->type = e0->cls; v->evdnctype = e0->type; v->tid = cg->tid;
            v->start = (uint32_t)left; v->stop = (uint32_t)right;
            v->rep_b1 = pe[j]->b1; v->rep_b2 = pe[j]->b2; v->rep_arrival = pe[j]->arrival;
            if (v->start <= v->stop) vl_push(&vars, v); else variant_free(v);
        }
        free(parent); free(done); free(pe);
    }
    /* components are numbered from the largest sorted position down, the variant list is built by
     * prepending, and sort_variants is stable: equal (start,stop) come out in ascending order of the
     * component's largest sorted position */
    for (int i = 1; i < vars.n; i++) {
        variant_t* v = vars.v[i]; int j = i - 1;
        while (j >= 0 && sorted_before((int32_t)v->rep_b1, (int32_t)v->rep_b2, v->rep_arrival,
                                       (int32_t)vars.v[j]->rep_b1, (int32_t)vars.v[j]->rep_b2, vars.v[j]->rep_arrival)) { vars.v[j + 1] = vars.v[j]; j--; }
        vars.v[j + 1] = v;
    }
    sort_variants(&vars);
    *out = vars;
    *used_out = used; *n_used_out = n_used;
#undef USED_PUSH
}

static __thread void (*t_part_log)(void* arg, int32_t tid, int first, size_t off);
static __thread void* t_part_arg;
static void group_replay(driver* d, pgroup* G)
{
    int32_t cursor = 0;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        const gcontig* cg = &G->ctg[ci];
        const int32_t tid = cg->tid;
        d->depth_tid = g_region_tid < 0 ? tid : -1;     /* -c: the depth of a variant is taken from the file (it reaches outside the stretch) */
        if (g_mg && t_part_log) {                       /* a replay worker: the job's buffer, cut per contig (replay_thread) */
            fflush(t_out);
            t_part_log(t_part_arg, tid, cg->first, (size_t)ftell(t_out));
        } else if (g_mg) {                              /* one VCF part per contig, concatenated by rank 0 in contig order */
            char path[512];
            mg_path(g_mg, path, sizeof path, "part", tid);
            fflush(stdout);
            if (!freopen(path, cg->first ? "w" : "a", stdout)) fatalf("cannot write %s", path);
        }
        for (int f = cg->fl0; f < cg->fl1; f++) {
            variant_list vs = {0};
            evidence_t** used = NULL; int64_t n_used = 0;
            group_process_flush(d, G, cg, f, &cursor, &vs, &used, &n_used);
            if (g_vcfname == NULL) {
                merge_variants(&vs, d->sequences[tid], d->seqlen[tid], 1);
                print_variants(d, &vs);
            } else {
                merge_variants(&vs, d->sequences[tid], d->seqlen[tid], 0);
                print_knownvariants(d, &g_known, &vs);
            }
            fflush(OUT);
            for (int i = 0; i < vs.n; i++) variant_free(vs.v[i]);
            free(vs.v);
            for (int64_t i = 0; i < n_used; i++) evidence_free(used[i]);
            free(used);
        }
        if (g_vcfname != NULL && cg->last) {
            for (int ki = g_known.next; ki < g_known.n; ki++) {
                knownvariant_t* k = g_known.v[ki];
                print_vcf_line(d, k);
                if (k->evdnctype == EV_SPLIT_READ && is_indel_supported(d, k)) printf(";%s", g_sample_name);
                printf("\n");
            }
            g_known.next = g_known.n;
        }
    }
    /* evidence objects that were built with their candidate but belong to slots no flush of this group consumed (they are still
     * pending: a later piece builds them again from the candidate) */
    const size_t ns = (size_t)(G->n_front + G->sv_n) * IM_MAX_EV;
    for (size_t i = 0; i < ns; i++) if (G->ev_cache[i]) { evidence_free(G->ev_cache[i]); G->ev_cache[i] = NULL; }
}
