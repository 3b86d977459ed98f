/* host_multirank.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: one process per GPU: plan (contig owners, piece walkers), rendezvous, the pre-walk logs and their ONE all-gather, packages
 * of walked groups for the owner, hand-over of aborted runs, the parts joined by rank 0. */

/* ============================================================== multi-GPU == */
/*
 * One process per GPU (RANK / WORLD_SIZE / LOCAL_RANK in the environment, as torch.distributed.run sets them).  Contigs
 * are independent in the reference except for three things that are carried from one contig to the next, and those are
 * what the ranks exchange -- in ONE all-gather (RCCL over xGMI) of per-rank logs, before any rank starts its main pass:
 *   the global read counter that places the READCHUNK flushes (numread is never reset, src/indelminer.c:617,764)
 *       -> counted reads per contig, so that a rank starts contig c at the single run's count;
 *   the pair table (516-615), whose stale entries (first mates whose second mate never comes) lower every later marker
 *       (find_marker, 211-233)  -> EVERY record that may go through the pair table, as an event (position, |isize|, read
 *       group, first or second mate, name): each rank replays all contigs' events through the table with the final
 *       insert lengths -- the same adds, look-ups and removals as the walk, exactly, whatever the names and sizes are;
 *   the insert-length table when no config file is given (estimate_insertlengths, src/bamoperations.c:15-86)
 *       -> per read group min / max and where it was first seen, merged in file order.
 * Each rank gets them from ONE pre-walk over its own contigs (the estimation pass the reference runs anyway).  Contigs go
 * to ranks by size (longest first onto the least loaded rank, sizes = compressed bytes from the index).  Then every rank
 * runs the device pipeline over its contigs, writes one VCF part per contig and a flag file when it is done; rank 0
 * concatenates the parts in contig order behind the header: the single run's bytes.
 */

typedef struct { uint8_t* p; size_t n, cap; } mgbuf;
static void* mgbuf_take(mgbuf* b, size_t bytes)
{
    if (b->n + bytes > b->cap) { b->cap = (b->cap + bytes) * 2 + 4096; b->p = xrealloc(b->p, b->cap); }
    void* at = b->p + b->n;
    memset(at, 0, bytes);
    b->n += bytes;
    return at;
}

/* a run that hangs in a collective (a rank died, a stale rendezvous) ends here, not never */
static void mg_rank_failed(void);
static volatile double g_mg_deadline = 0;
static const char* volatile g_mg_waiting_for = "";
static void* mg_watchdog(void* arg)
{
    (void)arg;
    for (;;) {
        struct timespec ts = { 0, 200 * 1000 * 1000 };
        nanosleep(&ts, NULL);
        const double dl = g_mg_deadline;
        if (dl > 0 && now_ms() > dl) {
            fprintf(stderr, "indelminer: rank %d gave up waiting for the other ranks (%s)\n", g_mg_rank, g_mg_waiting_for);
            mg_rank_failed();                      /* the others stop with this rank's message instead of waiting out their own limit */
            _exit(3);
        }
    }
    return NULL;
}
static double mg_timeout_ms(void) { const char* e = getenv("INDELMINER_MG_TIMEOUT"); return (e ? atof(e) : 600.0) * 1e3; }
static void mg_arm(const char* what) { g_mg_waiting_for = what; g_mg_deadline = now_ms() + mg_timeout_ms(); }
static void mg_disarm(void) { g_mg_deadline = 0; }

/* contigs to ranks: longest first onto the least loaded rank (every rank computes the same plan from the same index) */
static void mg_plan(mgpu* m, const driver* d)
{
    const int32_t nt = d->hdr->n_targets;
    m->owner = xmalloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
    int64_t* w = xmalloc(sizeof(int64_t) * (size_t)(nt ? nt : 1));
    int32_t* by = xmalloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
    for (int32_t t = 0; t < nt; t++) { w[t] = (m->skip && m->skip[t]) ? 0 : bai_contig_bytes(d->idx, t); by[t] = t; }
    for (int32_t i = 1; i < nt; i++) {                  /* by weight, heaviest first; equal weights in contig order */
        const int32_t t = by[i]; int32_t j = i - 1;
        while (j >= 0 && w[by[j]] < w[t]) { by[j + 1] = by[j]; j--; }
        by[j + 1] = t;
    }
    int64_t* load = xcalloc((size_t)m->world, sizeof(int64_t));
    const char* pl = getenv("INDELMINER_MG_PLAN");
    for (int32_t i = 0; i < nt; i++) {
        const int32_t t = by[i];
        int best = 0;
        for (int r = 1; r < m->world; r++) if (load[r] < load[best]) best = r;
        if (pl && strcmp(pl, "modulo") == 0) best = t % m->world;
        m->owner[t] = best;
        load[best] += w[t] + 1;                          /* + 1: empty contigs spread out too */
    }
    free(w); free(by); free(load);
}

static void mg_write_flag(const mgpu* m, const char* text);
static void mg_rank_failed(void)
{
    static int once = 0;
    if (!g_mg || g_mg->dir[0] == 0 || __sync_lock_test_and_set(&once, 1)) return;
    mg_write_flag(g_mg, "-2\n");
}

static void mg_write_flag(const mgpu* m, const char* text)
{
    char path[512], tmp[520];
    mg_path(m, path, sizeof path, "done", m->rank);
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE* fp = fopen(tmp, "w");
    if (!fp) return;
    fputs(text, fp);
    fclose(fp);
    rename(tmp, path);
}

static struct { mgpu* m; driver* d; char path[512]; char token[96]; pthread_t th; int pending; } g_rdv;
static void* mg_comm_bringup(void* arg);

static void mg_rendezvous(mgpu* m, driver* d)
{
    /* The RCCL unique id travels through a file in a directory every rank can see (one node).  The directory is this run's
     * alone: named after the launcher's process (the ranks of one run share a parent) unless the caller names one, emptied
     * by rank 0 before the id is published, and an id file is believed only if it carries this run's token. */
    const char* dir = getenv("INDELMINER_RENDEZVOUS");
    if (dir) snprintf(m->dir, sizeof m->dir, "%s", dir);
    else snprintf(m->dir, sizeof m->dir, "/tmp/indelminer_mgpu_%s_%ld", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", (long)getppid());
    pthread_t wd;
    if (pthread_create(&wd, NULL, mg_watchdog, NULL) == 0) pthread_detach(wd);
    char path[512];
    snprintf(path, sizeof path, "%s/rccl_id", m->dir);
    /* what the ranks of ONE run share and no other run has: the launcher's run id, port and process (or what the caller says) */
    char token[96];
    memset(token, 0, sizeof token);
    if (getenv("INDELMINER_RUN_TOKEN")) snprintf(token, sizeof token, "%s", getenv("INDELMINER_RUN_TOKEN"));
    else snprintf(token, sizeof token, "%s:%s:%ld", getenv("TORCHELASTIC_RUN_ID") ? getenv("TORCHELASTIC_RUN_ID") : "", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", (long)getppid());
    if (m->rank == 0) {
        /* no fork() here: the GPU helper thread is inside the HIP runtime's start-up */
        if (mkdir(m->dir, 0700) != 0 && errno != EEXIST) fatalf("cannot create the rendezvous directory %s", m->dir);
        {
            /* a directory that was already there must be this user's own and nobody else's to write (a predictable name under
             * /tmp: somebody else may have made it, or a link of that name) -- the ranks trust what they find in it */
            struct stat sb;
            if (lstat(m->dir, &sb) != 0 || !S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & (S_IWGRP | S_IWOTH)))
                fatalf("the rendezvous directory %s is not a directory of this user's alone", m->dir);
        }
        unlink(path);
        {   /* whatever an earlier run left behind: parts, flags, logs */
            DIR* dp = opendir(m->dir);
            if (dp) {
                struct dirent* de;
                while ((de = readdir(dp)) != NULL) {
                    if (strncmp(de->d_name, "part.", 5) != 0 && strncmp(de->d_name, "done.", 5) != 0 && strncmp(de->d_name, "pkg.", 4) != 0 && strncmp(de->d_name, "rccl_id", 7) != 0) continue;
                    char victim[800];
                    snprintf(victim, sizeof victim, "%s/%s", m->dir, de->d_name);
                    unlink(victim);
                }
                closedir(dp);
            }
        }
        snprintf(g_mg_header_path, sizeof g_mg_header_path, "%s/part.header", m->dir);
    }
    /* The id's trip and the communicator bring-up (about two seconds inside librccl) run beside the pre-walk, which needs neither:
     * the first collective joins this thread (mg_allgather). */
    g_rdv.m = m; g_rdv.d = d;
    snprintf(g_rdv.path, sizeof g_rdv.path, "%s", path);
    memcpy(g_rdv.token, token, sizeof token);
    if (pthread_create(&g_rdv.th, NULL, mg_comm_bringup, NULL) != 0) fatalf("cannot start the communicator thread");
    g_rdv.pending = 1;
}

static void* mg_comm_bringup(void* arg)
{
    (void)arg;
    mgpu* m = g_rdv.m; driver* d = g_rdv.d;
    const char* path = g_rdv.path;
    char token[96], tmp[520];
    memcpy(token, g_rdv.token, sizeof token);
    uint8_t id[IM_COMM_ID_BYTES];
    /* the HIP runtime is up before librccl is asked for anything: the context exists (its reference may still be on its way) */
    pthread_mutex_lock(&d->gpu_mu);
    while (!d->ctx_ready) pthread_cond_wait(&d->gpu_cv, &d->gpu_mu);
    pthread_mutex_unlock(&d->gpu_mu);
    if (d->ctx_rc != IM_OK) return NULL;                /* the main thread reports it (gpu_wait) */
    if (m->rank == 0) {
        if (im_comm_unique_id(id) != IM_OK) fatalf("im_comm_unique_id: %s", im_comm_last_error());
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE* fp = fopen(tmp, "wb");
        if (!fp || fwrite(id, 1, sizeof id, fp) != sizeof id || fwrite(token, 1, sizeof token, fp) != sizeof token) fatalf("cannot write %s", tmp);
        fclose(fp);
        if (rename(tmp, path) != 0) fatalf("cannot publish %s", path);
    } else {
        const double t_end = now_ms() + 120e3;
        for (;;) {
            /* an id file that does not carry this run's token is somebody else's (an earlier run in a re-used directory) */
            {
                struct stat sb;
                if (lstat(m->dir, &sb) == 0 && (!S_ISDIR(sb.st_mode) || sb.st_uid != getuid() || (sb.st_mode & (S_IWGRP | S_IWOTH))))
                    fatalf("the rendezvous directory %s is not a directory of this user's alone", m->dir);
            }
            char seen[sizeof token];
            FILE* fp = fopen(path, "rb");
            if (fp) {
                const size_t got = fread(id, 1, sizeof id, fp), got2 = fread(seen, 1, sizeof seen, fp);
                fclose(fp);
                if (got == sizeof id && got2 == sizeof seen && memcmp(seen, token, sizeof token) == 0) break;
            }
            if (now_ms() > t_end) fatalf("rank %d: no RCCL id of this run at %s after 120 s", m->rank, path);
            struct timespec ts = { 0, 20 * 1000 * 1000 };
            nanosleep(&ts, NULL);
        }
    }
    mg_arm("communicator bring-up");
    if (im_comm_init(d->gpu, id, m->rank, m->world, &m->comm) != IM_OK) fatalf("im_comm_init: %s", im_comm_last_error());
    mg_disarm();
    if (g_timing) fprintf(stderr, "    [timing] rank %d: RCCL communicator of %d ranks is up\n", m->rank, m->world);
    return NULL;
}
static void mg_comm_join(void) { if (g_rdv.pending) { g_rdv.pending = 0; pthread_join(g_rdv.th, NULL); } }

/* every rank contributes `bytes` bytes (a multiple of 4); all[] receives world * bytes */
static void mg_allgather(mgpu* m, driver* d, const void* mine, void* all, size_t bytes)
{
    mg_comm_join();                                 /* the communicator: brought up beside the pre-walk */
    void *ds = NULL, *dr = NULL;
    if (im_dev_alloc(d->gpu, bytes, &ds) != IM_OK || im_dev_alloc(d->gpu, bytes * (size_t)m->world, &dr) != IM_OK) fatalf("im_dev_alloc: %s", im_last_error(d->gpu));
    if (im_dev_upload(d->gpu, ds, mine, bytes) != IM_OK) fatalf("im_dev_upload: %s", im_last_error(d->gpu));
    void* st = im_ctx_stream(d->gpu);
    mg_arm("the all-gather of the shard logs");
    if (im_comm_allgather(m->comm, ds, dr, bytes, st) != IM_OK) fatalf("im_comm_allgather: %s", im_comm_last_error());
    if (im_stream_sync(d->gpu, st) != IM_OK) fatalf("im_stream_sync: %s", im_last_error(d->gpu));
    mg_disarm();
    if (im_dev_download(d->gpu, all, dr, bytes * (size_t)m->world) != IM_OK) fatalf("im_dev_download: %s", im_last_error(d->gpu));
    im_dev_free(d->gpu, ds); im_dev_free(d->gpu, dr);
}

typedef struct { char name[48]; int32_t min, max, first_tid; int64_t first_rec; int32_t seen; } mg_rg;

static int mg_rg_index(mg_rg* rgs, int* pn, const char* rgname)
{
    int k = *pn - 1;                                    /* the last one first: records of a library come in runs */
    while (k >= 0 && strcmp(rgs[k].name, rgname) != 0) k--;
    if (k >= 0) return k;
    if (*pn == MG_MAX_RG || strlen(rgname) >= sizeof rgs[0].name) fatalf("at most %d read groups with names under %zu bytes are supported here", MG_MAX_RG, sizeof rgs[0].name);
    k = (*pn)++;
    memset(&rgs[k], 0, sizeof rgs[k]);
    snprintf(rgs[k].name, sizeof rgs[k].name, "%s", rgname);
    return k;
}

/* One contig of the pre-walk: insert-length statistics per read group (estimate_insertlengths, src/bamoperations.c:15-86),
 * counted reads, and the log of the records that may go through the pair table.  out = the rank's exchange buffer (NULL:
 * statistics only).  Thread-safe: everything it touches is the caller's.
 * The contig's block: { tid, counted (2 words), events, bytes of events }, then per event { pos, |isize|, record index,
 * first-mate flag | read group << 8 | name length << 16 } and the name with its NUL, padded to a word.
 * cov (NULL: not wanted): the piece's share of the observed coverage (estimate_average_coverage). */
static void prewalk_piece(const driver* d, bgzf_reader* r, int32_t t, int32_t beg, int32_t end, int estimate, mg_rg* rgs, int* pn_rg, mgbuf* out, covlist* cov)
{
    bam_region_iter it;
    size_t head_at = 0;
    if (out) { head_at = out->n; int32_t* hd = mgbuf_take(out, 20); hd[0] = t; }
    if (bam_piece_begin(&it, r, d->idx, t, beg, end) != 0) return;
    bam_record b; memset(&b, 0, sizeof b);
    int64_t counted = 0;
    int32_t rec = 0, n_ev = 0;
    const size_t ev_at = out ? out->n : 0;
    /* the walkers' record loop: records taken from the inflated block in place, base qualities left behind (nothing here reads them) */
    it.drop_qual = 1;
    const int64_t cap = 1 << 20;
    uint8_t* one = xmalloc((size_t)cap + 64);
    for (;;) {
        int32_t len = 0;
        const int rc = bam_region_next_raw(&it, one, cap, &len, &b);
        if (rc == -2) fatalf("a BAM record larger than %ld bytes", (long)cap);
        if (rc < 0) fatalf("error while reading %s", d->bam_name);
        if (rc == 0) break;
        const int flag = b.flag;
        const int32_t this_rec = rec++;
        if (cov) cov_record(cov, &b);
        if (estimate && (flag & 0x1) && !(flag & 0x4) && (flag & 0x2) && !(flag & (0x100 | 0x200 | 0x400)) && b.isize >= 0) {
            /* the tag's type is asserted in front of the two mate-position tests (src/bamoperations.c:37-46) */
            const uint8_t* rg = bam_aux_find(&b, "RG");
            const char* rgname = "generic";
            if (rg) { forceassert(rg[0] == 'Z'); rgname = bam_aux_str(rg); }
            if (b.mpos - b.pos >= 0 && b.isize >= b.mpos - b.pos) {
                mg_rg* g = &rgs[mg_rg_index(rgs, pn_rg, rgname)];
                if (!g->seen) { g->seen = 1; g->min = g->max = b.isize; g->first_tid = t; g->first_rec = ((int64_t)(b.pos < 0 ? 0 : b.pos) << 32) | (uint32_t)this_rec; }
                else { if (g->min > b.isize) g->min = b.isize; if (g->max < b.isize) g->max = b.isize; }
            }
        }
        if (!out) continue;
        if (flag & (0x100 | 0x200 | 0x400 | 0x800)) continue;
        if (!(flag & 0x1)) continue;
        const int aligned = !(flag & 0x4), mate_aligned = !(flag & 0x8);
        if (aligned && mate_aligned && b.tid != b.mtid) continue;
        counted++;
        /* what src/indelminer.c:516-522 asks of a record apart from |isize| > range[1], which waits for the final table */
        if (aligned && mate_aligned && !(flag & 0x2) && ((flag & 0x10) != 0) != ((flag & 0x20) != 0) &&
            (uint32_t)abs(b.isize) < O.maxpedelsize) {
            const uint8_t* rg = bam_aux_find(&b, "RG");
            const int gi = mg_rg_index(rgs, pn_rg, rg ? bam_aux_str(rg) : "generic");
            const size_t nl = (size_t)b.l_qname;
            int32_t* ev = mgbuf_take(out, 16 + ((nl + 3) & ~(size_t)3));
            ev[0] = b.pos; ev[1] = abs(b.isize); ev[2] = this_rec;
            ev[3] = (b.pos < b.mpos ? 1 : 0) | (gi << 8) | ((int32_t)nl << 16);
            memcpy(ev + 4, BAMR_QNAME(&b), nl);
            n_ev++;
        }
    }
    free(one);
    if (!out) return;
    int32_t* hd = (int32_t*)(out->p + head_at);
    hd[1] = (int32_t)(counted & 0xffffffff); hd[2] = (int32_t)(counted >> 32); hd[3] = n_ev; hd[4] = (int32_t)(out->n - ev_at);
}

#define MG_MAGIC 0x4d473033
#define MG_HEAD_WORDS 8         /* magic, read groups, contigs, bytes used (2 words), 3 spare */

/* The pre-walk over the pieces this rank walks (mg_plan_walks), spread over threads: the rank's log, ready for the exchange --
 * header, read groups, then one block per piece in file order. */
typedef struct { const driver* d; const piece_t* pieces; const int32_t* mine; int n_mine, t0, step, estimate; mg_rg rgs[MG_MAX_RG]; int n_rg; mgbuf* out; covlist cov; } prewalk_job;
static void* prewalk_thread(void* arg)
{
    prewalk_job* j = arg;
    bgzf_reader* r = bgzf_open(j->d->bam_name);
    if (!r) fatalf("error in opening the file %s", j->d->bam_name);
    bgzf_set_workers(r, 0);
    bam_header* h = bam_header_load(r);
    for (int k = j->t0; k < j->n_mine; k += j->step) {
        const piece_t* pc = &j->pieces[j->mine[k]];
        const size_t at = j->out[k].n;
        prewalk_piece(j->d, r, pc->tid, pc->beg, pc->end, j->estimate, j->rgs, &j->n_rg, &j->out[k], j->estimate ? &j->cov : NULL);
        ((int32_t*)(j->out[k].p + at))[0] = pc->index;
        /* the read-group indices of the events are this thread's: the main thread maps them onto the rank's list (mg_prewalk) */
    }
    bam_header_free(h);
    bgzf_close(r);
    return NULL;
}

static void mg_prewalk(mgpu* m, driver* d, int estimate, mgbuf* out, const piece_t* pieces, int n_pieces, const int32_t* piece_walker)
{
    int32_t* mine = xmalloc(sizeof(int32_t) * (size_t)(n_pieces ? n_pieces : 1));
    int n_mine = 0;
    for (int i = 0; i < n_pieces; i++) if (piece_walker[i] == m->rank) mine[n_mine++] = i;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    {
        FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r");
        long quota = 0, period = 0;
        if (fp) { if (fscanf(fp, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < ncpu) ncpu = quota / period; fclose(fp); }
    }
    int nt = getenv("INDELMINER_WALKERS") ? atoi(getenv("INDELMINER_WALKERS")) : (int)(ncpu > 16 ? 16 : ncpu);
    if (nt > n_mine) nt = n_mine;
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    mgbuf* pieces_out = xcalloc((size_t)(n_mine ? n_mine : 1), sizeof(mgbuf));
    prewalk_job* jobs = xcalloc((size_t)nt, sizeof(prewalk_job));
    pthread_t* th = xmalloc(sizeof(pthread_t) * (size_t)nt);
    for (int i = 0; i < nt; i++) {
        jobs[i].d = d; jobs[i].pieces = pieces; jobs[i].mine = mine; jobs[i].n_mine = n_mine; jobs[i].t0 = i; jobs[i].step = nt;
        jobs[i].estimate = estimate; jobs[i].out = pieces_out;
        cov_init(&jobs[i].cov, d->hdr->n_targets);
        if (pthread_create(&th[i], NULL, prewalk_thread, &jobs[i]) != 0) fatalf("cannot start a pre-walk thread");
    }
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    /* one read-group list for the rank: each thread's list onto it (names; extrema and first sightings merged) */
    mg_rg* rgs = xcalloc(MG_MAX_RG, sizeof(mg_rg));
    int n_rg = 0;
    int (*remap)[MG_MAX_RG] = xcalloc((size_t)nt, sizeof *remap);
    for (int i = 0; i < nt; i++)
        for (int k = 0; k < jobs[i].n_rg; k++) {
            const mg_rg* g = &jobs[i].rgs[k];
            const int at = mg_rg_index(rgs, &n_rg, g->name);
            remap[i][k] = at;
            if (!g->seen) continue;
            mg_rg* t = &rgs[at];
            if (!t->seen) { const int32_t keep = 1; *t = *g; t->seen = keep; }
            else {
                if (g->min < t->min) t->min = g->min;
                if (g->max > t->max) t->max = g->max;
                if (g->first_tid < t->first_tid || (g->first_tid == t->first_tid && g->first_rec < t->first_rec)) { t->first_tid = g->first_tid; t->first_rec = g->first_rec; }
            }
        }
    mgbuf_take(out, 4 * (MG_HEAD_WORDS + (size_t)MG_MAX_RG * MG_RG_WORDS));
    for (int k = 0; k < n_mine; k++) {
        /* the piece's block, its events' read groups in the rank's numbering */
        int32_t* hd = (int32_t*)pieces_out[k].p;
        int32_t* ev = hd + 5;
        const int who = k % nt;
        for (int32_t e = 0; e < hd[3]; e++) {
            const int gi = (ev[3] >> 8) & 0xff, nl = (ev[3] >> 16) & 0xffff;
            ev[3] = (ev[3] & ~0xff00) | (remap[who][gi] << 8);
            ev += 4 + (nl + 3) / 4;
        }
        memcpy(mgbuf_take(out, pieces_out[k].n), pieces_out[k].p, pieces_out[k].n);
        free(pieces_out[k].p);
    }
    {
        /* the rank's share of the observed coverage (no config file): { segments, per contig the span sum (2 words) }, the segments */
        const int32_t ntg = d->hdr->n_targets;
        int64_t nseg = 0;
        for (int i = 0; i < nt; i++) { cov_close(&jobs[i].cov); nseg += jobs[i].cov.n; }
        int32_t* cw = mgbuf_take(out, 4 * (1 + 2 * (size_t)ntg + 3 * (size_t)nseg));
        cw[0] = (int32_t)nseg;
        for (int32_t t = 0; t < ntg; t++) {
            uint64_t sm = 0;
            for (int i = 0; i < nt; i++) sm += jobs[i].cov.sum[t];
            cw[1 + 2 * t] = (int32_t)(uint32_t)sm; cw[2 + 2 * t] = (int32_t)(uint32_t)(sm >> 32);
        }
        int32_t* sg = cw + 1 + 2 * (size_t)ntg;
        for (int i = 0; i < nt; i++) {
            for (int64_t k = 0; k < jobs[i].cov.n; k++) { *sg++ = jobs[i].cov.seg[k].tid; *sg++ = jobs[i].cov.seg[k].beg; *sg++ = jobs[i].cov.seg[k].end; }
            cov_free(&jobs[i].cov);
        }
    }
    int32_t* w = (int32_t*)out->p;
    w[0] = MG_MAGIC; w[1] = n_rg; w[2] = n_mine; w[3] = (int32_t)(out->n & 0xffffffff); w[4] = (int32_t)((uint64_t)out->n >> 32);
    for (int k = 0; k < n_rg; k++) {
        int32_t* g = w + MG_HEAD_WORDS + (size_t)k * MG_RG_WORDS;
        memcpy(g, rgs[k].name, 48);
        g[12] = rgs[k].min; g[13] = rgs[k].max; g[14] = rgs[k].first_tid; g[15] = (int32_t)(rgs[k].first_rec >> 32); g[16] = rgs[k].seen; g[17] = (int32_t)(uint32_t)rgs[k].first_rec;
    }
    free(rgs); free(remap); free(pieces_out); free(jobs); free(th); free(mine);
}

static int cmp_mg_rg(const void* x, const void* y)
{
    const mg_rg* a = x; const mg_rg* b = y;
    if (a->first_tid != b->first_tid) return a->first_tid < b->first_tid ? -1 : 1;
    if (a->first_rec != b->first_rec) return a->first_rec < b->first_rec ? -1 : 1;
    return 0;
}

/* read groups met in several places -> one list in the order ONE sequential pass would have met them */
static int merge_rgs(mg_rg* all, int n_all, mg_rg* out)
{
    int n = 0;
    for (int i = 0; i < n_all; i++) {
        int j = 0;
        while (j < n && strcmp(out[j].name, all[i].name) != 0) j++;
        if (j == n) out[n++] = all[i];
        else {
            if (all[i].min < out[j].min) out[j].min = all[i].min;
            if (all[i].max > out[j].max) out[j].max = all[i].max;
            if (all[i].first_tid < out[j].first_tid || (all[i].first_tid == out[j].first_tid && all[i].first_rec < out[j].first_rec)) { out[j].first_tid = all[i].first_tid; out[j].first_rec = all[i].first_rec; }
        }
    }
    qsort(out, (size_t)n, sizeof(mg_rg), cmp_mg_rg);
    return n;
}

/* The merged list (exact names, first-met order) into the insert-length table the way ONE sequential pass builds it
 * (src/bamoperations.c:48-57): a name is looked up before it is added, and the table's look-up takes an OLDER entry of the same
 * bin whose name merely starts with it (src/hashtable.c:62-81) -- "lib1" met after "lib10" never gets an entry, its sizes widen
 * lib10's range.  Which entry a name goes to is settled when it is first met (entries are never removed, the oldest match wins),
 * so replaying the names in first-met order gives the sequential table exactly.  Returns the entry's range. */
static int32_t* rg_table_enter(driver* d, const mg_rg* g)
{
    qbin* hit = qhash_lookup(d->insertlengths, g->name, (int)strlen(g->name));
    if (hit) {
        int32_t* range = hit->val;
        if (g->min < range[0]) range[0] = g->min;
        if (g->max > range[1]) range[1] = g->max;
        return range;
    }
    int32_t* range = xmalloc(2 * sizeof(int32_t));
    range[0] = g->min; range[1] = g->max;
    qhash_add(d->insertlengths, g->name, (int)strlen(g->name), range);
    rg_order_push(g->name, range);
    return range;
}

/* estimate_insertlengths (src/bamoperations.c:15-86) with the contigs spread over threads: the pass is pure decode + a
 * min / max per read group, so contigs are independent and the per-thread lists merge exactly (rg_table_enter) */
typedef struct { const driver* d; const piece_t* pieces; int n_pieces, t0, step; mg_rg rgs[MG_MAX_RG]; int n_rg; covlist cov; } est_job;
static void* est_thread(void* arg)
{
    est_job* j = arg;
    bgzf_reader* r = bgzf_open(j->d->bam_name);
    if (!r) fatalf("error in opening the file %s", j->d->bam_name);
    bgzf_set_workers(r, 0);
    bam_header* h = bam_header_load(r);
    for (int i = j->t0; i < j->n_pieces; i += j->step) prewalk_piece(j->d, r, j->pieces[i].tid, j->pieces[i].beg, j->pieces[i].end, 1, j->rgs, &j->n_rg, NULL, &j->cov);
    bam_header_free(h);
    bgzf_close(r);
    return NULL;
}
/* pieces / n_pieces: how the file is cut for the walkers (walkpool_start); the pre-pass takes the same pieces, one thread per core */
static void estimate_insertlengths_threads(driver* d, const piece_t* pieces, int n_pieces)
{
    const char* e = getenv("INDELMINER_WALKERS");
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    {
        FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r");
        long quota = 0, period = 0;
        if (fp) { if (fscanf(fp, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < ncpu) ncpu = quota / period; fclose(fp); }
    }
    int nt = e ? atoi(e) : (int)(ncpu > 16 ? 16 : ncpu);
    if (nt > n_pieces) nt = n_pieces;
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    est_job* jobs = xcalloc((size_t)nt, sizeof(est_job));
    pthread_t* th = xmalloc(sizeof(pthread_t) * (size_t)nt);
    for (int i = 0; i < nt; i++) {
        jobs[i].d = d; jobs[i].pieces = pieces; jobs[i].n_pieces = n_pieces; jobs[i].t0 = i; jobs[i].step = nt;
        cov_init(&jobs[i].cov, d->hdr->n_targets);
        if (pthread_create(&th[i], NULL, est_thread, &jobs[i]) != 0) fatalf("cannot start an estimation thread");
    }
    mg_rg* all = xcalloc((size_t)nt * MG_MAX_RG, sizeof(mg_rg));
    int n_all = 0;
    for (int i = 0; i < nt; i++) { pthread_join(th[i], NULL); for (int k = 0; k < jobs[i].n_rg; k++) if (jobs[i].rgs[k].seen) all[n_all++] = jobs[i].rgs[k]; }
    mg_rg* merged = xcalloc((size_t)(n_all ? n_all : 1), sizeof(mg_rg));
    const int n = merge_rgs(all, n_all, merged);
    for (int j = 0; j < n; j++) rg_table_enter(d, &merged[j]);
    {
        covlist** ls = xmalloc(sizeof(covlist*) * (size_t)nt);
        for (int i = 0; i < nt; i++) ls[i] = &jobs[i].cov;
        cov_means_of_lists(d->hdr->n_targets, ls, nt);
        for (int i = 0; i < nt; i++) cov_free(&jobs[i].cov);
        free(ls);
    }
    free(all); free(merged); free(jobs); free(th);
}

/* the pair table of the replay: name -> the waiting first mate's start and contig; the entries that wait, for the floor */
typedef struct { int32_t start, tid, slot; } mg_wait;

static void mg_exchange_logs(mgpu* m, driver* d, int estimate, mgbuf mine, const piece_t* pieces, int n_pieces, const int32_t* piece_walker);
/* annotate mode (contigs walked one by one, by the main thread): the logs come from a pass of their own over this rank's pieces */
static void mg_exchange(mgpu* m, driver* d, int estimate, const piece_t* pieces, int n_pieces, const int32_t* piece_walker)
{
    mgbuf mine = { NULL, 0, 0 };
    mg_prewalk(m, d, estimate, &mine, pieces, n_pieces, piece_walker);
    phase_time("pre-walk of this rank's contigs (count, pair-table events, insert lengths)");
    mg_exchange_logs(m, d, estimate, mine, pieces, n_pieces, piece_walker);
}

/* exchange + merge: the insert-length table (when estimated), the counter prefix and the marker floor of every contig, and the first
 * piece some rank's walk did not survive (m->abort_piece) */
static void mg_exchange_logs(mgpu* m, driver* d, int estimate, mgbuf mine, const piece_t* pieces, int n_pieces, const int32_t* piece_walker)
{
    const int32_t nt = d->hdr->n_targets;
    /* ONE all-gather of fixed-size buffers.  Every rank derives the same size from the same file: pair-table events are a few
     * per thousand records, so a 64th of the file holds them many times over; a rank whose log does not fit says so in its
     * header and the exchange is repeated once with the size that does (all ranks see all headers: all agree). */
    size_t cap;
    {
        struct stat sb;
        const int64_t fsize = stat(d->bam_name, &sb) == 0 ? (int64_t)sb.st_size : 0;
        const char* e = getenv("INDELMINER_MG_LOG_BYTES");
        int64_t c = e ? atoll(e) : fsize / 64;
        const int64_t least = 4 * (MG_HEAD_WORDS + (int64_t)MG_MAX_RG * MG_RG_WORDS) + 20 * ((int64_t)n_pieces + 1) + 4 * (1 + 2 * (int64_t)nt);
        if (c < least) c = least;
        if (!e && c < (4 << 20)) c = 4 << 20;
        cap = ((size_t)c + 255) & ~(size_t)255;
    }
    uint8_t* all = NULL;
    for (int round = 0; round < 2; round++) {
        uint8_t* send = xcalloc(cap, 1);
        memcpy(send, mine.p, mine.n < cap ? mine.n : 4 * (size_t)MG_HEAD_WORDS);     /* too long: the header alone, it says how long */
        all = xmalloc(cap * (size_t)m->world);
        mg_allgather(m, d, send, all, cap);
        free(send);
        size_t need = 0;
        for (int rk = 0; rk < m->world; rk++) {
            const int32_t* a = (const int32_t*)(all + (size_t)rk * cap);
            forceassert(a[0] == MG_MAGIC);
            const size_t used = (size_t)(uint32_t)a[3] | ((size_t)(uint32_t)a[4] << 32);
            if (used > need) need = used;
        }
        if (need <= cap) break;
        if (round == 1) fatalf("internal: the shard logs did not fit the second exchange either");
        free(all); all = NULL;
        cap = (need + 255) & ~(size_t)255;
    }
    free(mine.p);
    if (estimate) {
        mg_rg* got = xcalloc((size_t)MG_MAX_RG * (size_t)m->world, sizeof(mg_rg));
        int n_got = 0;
        for (int rk = 0; rk < m->world; rk++) {
            const int32_t* a = (const int32_t*)(all + (size_t)rk * cap);
            for (int k = 0; k < a[1]; k++) {
                const int32_t* w = a + MG_HEAD_WORDS + (size_t)k * MG_RG_WORDS;
                if (!w[16]) continue;                   /* met on pair-table records only: not part of the estimate */
                mg_rg* g = &got[n_got++];
                memcpy(g->name, w, 48); g->name[47] = 0;
                g->min = w[12]; g->max = w[13]; g->first_tid = w[14]; g->first_rec = ((int64_t)w[15] << 32) | (uint32_t)w[17]; g->seen = 1;
            }
        }
        mg_rg* rgs = xcalloc((size_t)(n_got ? n_got : 1), sizeof(mg_rg));
        const int n = merge_rgs(got, n_got, rgs);          /* the order in which one process would have met them */
        free(got);
        for (int j = 0; j < n; j++) rg_table_enter(d, &rgs[j]);
        free(rgs);
    }
    /* where each piece's block lies, and range[1] of every (rank, read group) through the table's own look-up */
    const int32_t** block = xcalloc((size_t)n_pieces + 1, sizeof(int32_t*));
    const int32_t** cov_at = xcalloc((size_t)m->world, sizeof(int32_t*));
    int32_t* rmax = xmalloc(sizeof(int32_t) * (size_t)m->world * MG_MAX_RG);
    for (int rk = 0; rk < m->world; rk++) {
        const int32_t* a = (const int32_t*)(all + (size_t)rk * cap);
        for (int k = 0; k < MG_MAX_RG; k++) rmax[rk * MG_MAX_RG + k] = -1;
        for (int k = 0; k < a[1]; k++) {
            char name[48];
            memcpy(name, a + MG_HEAD_WORDS + (size_t)k * MG_RG_WORDS, 48); name[47] = 0;
            qbin* hit = qhash_lookup(d->insertlengths, name, (int)strlen(name));
            if (hit) rmax[rk * MG_MAX_RG + k] = ((int32_t*)hit->val)[1];    /* no entry: the walk stops at that record (must_find_hashtable) */
        }
        const int32_t* at = a + MG_HEAD_WORDS + (size_t)MG_MAX_RG * MG_RG_WORDS;
        for (int c = 0; c < a[2]; c++) {
            forceassert(at[0] >= 0 && at[0] < n_pieces && piece_walker[at[0]] == rk && block[at[0]] == NULL);
            block[at[0]] = at;
            at += 5 + at[4] / 4;
        }
        cov_at[rk] = at;
    }
    if (estimate) {
        /* every rank's span sums and covered segments -> the coverage table, the same on all ranks (rank 0 prints it) */
        uint64_t* sums = xcalloc((size_t)nt + 1, sizeof(uint64_t));
        int64_t nseg = 0;
        for (int rk = 0; rk < m->world; rk++) nseg += cov_at[rk][0];
        covseg* seg = xmalloc(sizeof(covseg) * (size_t)(nseg ? nseg : 1));
        nseg = 0;
        for (int rk = 0; rk < m->world; rk++) {
            const int32_t* cw = cov_at[rk];
            for (int32_t t = 0; t < nt; t++) sums[t] += (uint64_t)(uint32_t)cw[1 + 2 * t] | ((uint64_t)(uint32_t)cw[2 + 2 * t] << 32);
            const int32_t* sg = cw + 1 + 2 * (size_t)nt;
            for (int32_t k = 0; k < cw[0]; k++, sg += 3) { seg[nseg].tid = sg[0]; seg[nseg].beg = sg[1]; seg[nseg].end = sg[2]; nseg++; }
        }
        cov_means(nt, sums, seg, nseg);
        free(sums); free(seg);
    }
    free(cov_at);
    /* the replay: every piece's events through ONE pair table, in file order, as the single run serves it */
    m->piece_prefix = xcalloc((size_t)n_pieces + 1, sizeof(int64_t));
    m->floor = xmalloc(sizeof(int) * ((size_t)nt + 1));
    for (int32_t t = 0; t <= nt; t++) m->floor[t] = INT_MAX;
    qhash* table = qhash_new(16);
    mg_wait** live = NULL; int32_t n_live = 0, cap_live = 0;
    int64_t run = 0;
    m->abort_piece = n_pieces;
    for (int pi = 0; pi < n_pieces; pi++) {
        const int32_t t = pieces[pi].tid;
        m->piece_prefix[pi] = run;
        if (block[pi] && block[pi][1] == -1 && block[pi][2] == -1) { m->abort_piece = pi; break; }      /* that piece's walk met a record the reference dies on: nothing behind it is printed by this run */
        if (pieces[pi].first) {
            int fl = INT_MAX;
            for (int32_t i = 0; i < n_live; i++) if (live[i]->start < fl) fl = live[i]->start;
            m->floor[t] = fl;
        }
        const int32_t* hd = block[pi];
        forceassert(hd != NULL);
        run += (int64_t)(uint32_t)hd[1] | ((int64_t)hd[2] << 32);
        const int32_t* ev = hd + 5;
        for (int32_t k = 0; k < hd[3]; k++) {
            const int32_t pos = ev[0], aisize = ev[1], word = ev[3];
            const int first = word & 1, gi = (word >> 8) & 0xff, nl = (word >> 16) & 0xffff;
            const char* name = (const char*)(ev + 4);
            ev += 4 + (nl + 3) / 4;
            const int32_t r1 = rmax[piece_walker[pi] * MG_MAX_RG + gi];
            if (r1 < 0 || aisize <= r1) continue;                                   /* src/indelminer.c:519 */
            qbin* hb = qhash_lookup(table, name, nl);
            if (hb && ((mg_wait*)hb->val)->tid != t) m->cross = 1;                  /* an entry of an earlier contig under this name */
            if (first) {
                mg_wait* w = xmalloc(sizeof *w);
                w->start = pos; w->tid = t; w->slot = n_live;
                qhash_add(table, name, nl, w);
                if (n_live == cap_live) { cap_live = cap_live ? cap_live * 2 : 256; live = xrealloc(live, sizeof(mg_wait*) * (size_t)cap_live); }
                live[n_live++] = w;
            } else {
                /* not in the table: the mate is fetched from the file, entered and removed at once (537-575, 610-612); in it: removed */
                mg_wait* w = qhash_remove(table, name, nl);
                if (w) { live[w->slot] = live[--n_live]; live[w->slot]->slot = w->slot; free(w); }
            }
        }
    }
    qhash_free(table, free);
    free(live); free(block); free(rmax); free(all);
}

static void mg_restore_stdout(mgpu* m) { fflush(stdout); if (m->out_fd >= 0) dup2(m->out_fd, STDOUT_FILENO); }

static void mg_discard_dir(mgpu* m)
{
    DIR* dp = opendir(m->dir);
    if (!dp) return;
    struct dirent* de;
    while ((de = readdir(dp)) != NULL) {
        if (strncmp(de->d_name, "part.", 5) != 0 && strncmp(de->d_name, "done.", 5) != 0 && strncmp(de->d_name, "pkg.", 4) != 0 && strncmp(de->d_name, "rccl_id", 7) != 0) continue;
        char victim[800];
        snprintf(victim, sizeof victim, "%s/%s", m->dir, de->d_name);
        unlink(victim);
    }
    closedir(dp);
    rmdir(m->dir);
}

/* rank 0, at the very end: the parts in contig order behind the header that is already on the real stdout */
static void mg_finish(mgpu* m, driver* d)
{
    fflush(stdout);
    if (!freopen("/dev/null", "w", stdout)) { }        /* the last part is closed */
    {
        char text[64];
        snprintf(text, sizeof text, "%d\n", m->abort_tid);
        mg_write_flag(m, text);                         /* every part of this rank is complete (up to the contig it names) */
    }
    if (m->rank == 0) {
        /* the other ranks' flags: no collective at the end, a rank that is done is done */
        int first_abort = m->abort_tid >= 0 ? m->abort_tid : INT_MAX;
        mg_arm("the other ranks' output");
        for (int rk = 1; rk < m->world; rk++) {
            char path[512];
            snprintf(path, sizeof path, "%s/done.%d", m->dir, rk);
            for (;;) {
                FILE* fp = fopen(path, "r");
                int v = 0, got = 0;
                if (fp) { got = fscanf(fp, "%d", &v) == 1; fclose(fp); }
                if (got) {
                    if (v == -2) { fprintf(stderr, "indelminer: rank %d failed\n", rk); _exit(EXIT_FAILURE); }
                    if (v >= 0 && v < first_abort) first_abort = v;
                    break;
                }
                struct timespec ts = { 0, 5 * 1000 * 1000 };
                nanosleep(&ts, NULL);
            }
        }
        mg_disarm();
        char path[512], buf[1 << 16];
        int det_blocks = 0;
        for (int32_t t = -1; t < d->hdr->n_targets && t < first_abort; t++) {
            if (t < 0) snprintf(path, sizeof path, "%s", g_mg_header_path); else mg_path(m, path, sizeof path, "part", t);
            FILE* fp = fopen(path, "rb");
            if (!fp) continue;                          /* a contig nobody printed for */
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, fp)) > 0) {
                size_t off = 0;
                while (off < got) {
                    /* -o detailed: a 0x01 byte stands where a block's number goes (print_det_output) */
                    const char* mark = memchr(buf + off, 1, got - off);
                    const size_t upto = mark ? (size_t)(mark - buf) : got;
                    while (off < upto) { const ssize_t w = write(m->out_fd, buf + off, upto - off); if (w <= 0) fatalf("write to stdout failed"); off += (size_t)w; g_out_bytes += w; }
                    if (mark) {
                        char num[16];
                        const int nl = snprintf(num, sizeof num, "%d", ++det_blocks);
                        if (write(m->out_fd, num, (size_t)nl) != nl) fatalf("write to stdout failed");
                        g_out_bytes += nl;
                        off++;
                    }
                }
            }
            fclose(fp);
        }
        for (int32_t t = -1; t < d->hdr->n_targets; t++) {
            if (t < 0) snprintf(path, sizeof path, "%s", g_mg_header_path); else mg_path(m, path, sizeof path, "part", t);
            unlink(path);
        }
        for (int rk = 0; rk < m->world; rk++) { snprintf(path, sizeof path, "%s/done.%d", m->dir, rk); unlink(path); }
        snprintf(path, sizeof path, "%s/rccl_id", m->dir);
        unlink(path);
        rmdir(m->dir);
        if (first_abort != INT_MAX) {
            /* a record the reference dies on, in contig first_abort: what is in front of that contig is out; the
             * record-at-a-time child prints the rest and dies as the reference does (handoff_to_host_child) */
            mg_restore_stdout(m);
            handoff_to_host_child();
        }
    }
    im_comm_destroy(m->comm);
}

/* PIECES of contigs are walked at once (inflate, count, triage launches: one thread's worth of host work per walker), each
 * walker with its own BAM reader, pinned chunk ring, device arrays and stream.  A claim is a run of consecutive pieces that
 * goes into one group: a piece of a large contig on its own, or several whole small contigs.  A walked group's candidate
 * arrays are parked in a device allocation of their own and the walker goes on to its next claim.  The main thread takes the
 * walked groups in file order: it serves the pair table (entries carry over from one piece of a contig to the next), places
 * the flush points (group_resolve_flushes: the read counter carries over too), runs the stage (stage_run_group: in front of
 * the group's own candidates the evidence earlier pieces left pending) and hands the group to a replay worker once the
 * contig's depth array is complete.  Order of output is the order of the file. */
struct walkpool_s;
typedef struct claim_s claim_t;
typedef struct {
    struct walkpool_s* pool;
    driver wd;                          /* private: read-group cache */
    ppipe P;
    bgzf_reader* r; bam_header* hdr;
    claim_t* cur_claim; jmp_buf abort_jmp;  /* the claim being walked; where a walk that met a record the reference dies on ends up */
    pthread_t th;
} walker_t;

struct claim_s { int first, count; pgroup* G; int walked, aborted; };

/* a group whose stage is done, on its way through a replay worker: what it prints waits in buf until every group
 * before it has been printed */
/* a replay job's output; in a multi-GPU run it is cut into the stretches of its contigs (each goes to that contig's part file) */
typedef struct { int32_t tid; int first; size_t off; } rpart_t;
typedef struct { pgroup* G; char* buf; size_t len; int done; int last_of_contig; rpart_t* part; int n_part, cap_part; } rjob_t;
typedef struct { struct walkpool_s* pool; driver rd; pthread_t th; } replayer_t;

typedef struct walkpool_s {
    driver* d;
    piece_t* pieces; int n_pieces;      /* this process's share of the file, in file order */
    claim_t* claims; int n_claims, next_claim;
    int staged;                         /* claims the main thread is through with: walkers stay a bounded number of claims ahead */
    walker_t* w; int nw;
    int serial, go;
    int inflate_workers;                /* per reader; -1: as INDELMINER_THREADS says */
    rjob_t* jobs; int n_jobs, next_job, jobs_closed;    /* replay queue, in file order */
    int printed;                                        /* jobs whose output has been written */
    pthread_mutex_t mu; pthread_cond_t cv;
} walkpool_t;

/* A walked group's candidate arrays leave the walker's pipeline for an allocation of their own (the walker goes on to its next
 * claim) and come into the main thread's pipeline when the group's turn comes (stage_run_group). */
static void group_park_device(ppipe* P, pgroup* G)
{
    im_ctx* g = P->d->gpu;
    const size_t n = (size_t)G->n_cand, ns = n * IM_MAX_EV;
    const size_t bytes[10] = { (size_t)P->conf_bytes, 8 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * ns, 4 * ns, 4 * ns, 4 * n };
    void* src[10] = { P->bases, P->boff, P->len, P->tid, P->anchor, P->cand_rec, P->cls, P->b1, P->b2, P->range };
    G->sv_n = (int32_t)n; G->sv_bytes = P->conf_bytes;
    size_t total = 0;
    for (int k = 0; k < 10; k++) total += (bytes[k] + 255) & ~(size_t)255;
    char* slab = pdev_alloc(P, total);              /* one allocation per group: device allocation calls are not cheap */
    for (int k = 0; k < 10; k++) {
        G->sv[k] = slab;
        if (bytes[k]) GPU(im_dev_copy_async(g, G->sv[k], src[k], bytes[k], P->stream));
        slab += (bytes[k] + 255) & ~(size_t)255;
    }
    GPU(im_dev_memset(g, P->counters, 0, 64, P->stream));
    GPU(im_stream_sync(g, P->stream));
    P->conf_cand = 0; P->conf_err = 0; P->conf_bytes = 0;
}

/* ---- multi-GPU: a group walked by one rank, staged and replayed by another ---- */
/* What the owner of the contig needs of a walked group: the pieces' bounds and counted reads, the flush points (placed by the
 * walking rank, which knows the read counter in front of its pieces from the exchange), the kept records of not-proper pairs,
 * the candidates' record numbers and BAM records -- the group's HOST part, one block of bytes -- and the parked device arrays, one
 * device allocation.  Both travel device to device in ONE RCCL send / receive group when every rank has walked what it walks
 * (mg_ship_groups); `aborted` = the walk met a record the reference dies on. */
#define PKG_MAGIC 0x504b4734
typedef struct { int32_t magic, aborted, n_ctg, n_fp, n_npp, n_cand, sv_n, longest_read, has_range, pad; int64_t n_rec, npp_len, craw_len, sv_bytes; } pkg_head;
typedef struct { char* blob; size_t blob_len; void* slab; size_t slab_bytes; int aborted, walked; } mg_parcel;     /* a walked group on its way */
static mg_parcel* g_parcel = NULL;            /* [claims] filled by this rank's walkers for the claims other ranks own */

static void pkg_put(FILE* fp, const void* p, size_t bytes) { if (bytes && fwrite(p, 1, bytes, fp) != bytes) fatalf("cannot serialise a walked group"); }
static void pkg_get(FILE* fp, void* p, size_t bytes) { if (bytes && fread(p, 1, bytes, fp) != bytes) fatalf("a walked group arrived cut short"); }

static size_t group_slab_bytes(int32_t sv_n, int64_t sv_bytes)
{
    const size_t n = (size_t)sv_n, ns = n * IM_MAX_EV;
    const size_t bytes[10] = { (size_t)sv_bytes, 8 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * ns, 4 * ns, 4 * ns, 4 * n };
    size_t total = 0;
    for (int k = 0; k < 10; k++) total += (bytes[k] + 255) & ~(size_t)255;
    return total;
}

/* the walker's side: the group's host part as one block (the group itself is freed by the caller), its device slab stays parked */
static void package_pack(int ci, pgroup* G, int aborted)
{
    mg_parcel* pc = &g_parcel[ci];
    FILE* fp = open_memstream(&pc->blob, &pc->blob_len);
    if (!fp) fatalf("cannot serialise a walked group");
    pkg_head h;
    memset(&h, 0, sizeof h);
    h.magic = PKG_MAGIC; h.aborted = aborted;
    if (!aborted) {
        h.n_ctg = G->n_ctg; h.n_fp = G->n_fp; h.n_npp = G->n_npp; h.n_cand = G->n_cand; h.sv_n = G->sv_n; h.longest_read = g_longest_read; h.has_range = G->sv_range != NULL;
        h.n_rec = G->n_rec; h.npp_len = G->npp_len; h.craw_len = G->craw_len; h.sv_bytes = G->sv_bytes;
    }
    pkg_put(fp, &h, sizeof h);
    if (!aborted) {
        pkg_put(fp, G->ctg, sizeof(gcontig) * (size_t)G->n_ctg);
        pkg_put(fp, G->fp, sizeof(gfpoint) * (size_t)G->n_fp);
        pkg_put(fp, G->npp_off, sizeof(int64_t) * ((size_t)G->n_npp + (G->n_npp ? 1 : 0)));
        pkg_put(fp, G->npp_rec, sizeof(int32_t) * (size_t)G->n_npp);
        pkg_put(fp, G->npp_raw, (size_t)G->npp_len);
        pkg_put(fp, G->cand_rec, sizeof(int32_t) * (size_t)G->n_cand);
        pkg_put(fp, G->craw_off, sizeof(int64_t) * ((size_t)G->n_cand + (G->n_cand ? 1 : 0)));
        pkg_put(fp, G->craw, (size_t)G->craw_len);
        if (G->sv_range) pkg_put(fp, G->sv_range, sizeof(int32_t) * (size_t)G->n_cand);
        pc->slab = G->sv[0]; pc->slab_bytes = group_slab_bytes(G->sv_n, G->sv_bytes);
    }
    if (fclose(fp) != 0) fatalf("cannot serialise a walked group");
    pc->aborted = aborted; pc->walked = 1;
}

/* the owner's side: the group rebuilt from its host part; slab = its device arrays as they arrived.  NULL = the walk was aborted */
static pgroup* package_unpack(driver* d, const char* blob, size_t len, void* slab)
{
    FILE* fp = fmemopen((void*)blob, len, "rb");
    if (!fp) fatalf("cannot read a walked group");
    pkg_head h;
    pkg_get(fp, &h, sizeof h);
    if (h.magic != PKG_MAGIC) fatalf("what arrived is not a walked group of this run");
    if (h.aborted) { fclose(fp); return NULL; }
    if (h.longest_read > g_longest_read) note_long_read(d, h.longest_read);      /* this rank's realign launches must know of reads beyond 255 bases too */
    pgroup* G = xcalloc(1, sizeof(pgroup));
    G->from_package = 1;
    G->n_ctg = G->cap_ctg = h.n_ctg; G->n_fp = G->cap_fp = h.n_fp; G->n_npp = G->cap_npp = h.n_npp; G->n_cand = G->cap_cand = h.n_cand; G->sv_n = h.sv_n;
    G->n_rec = h.n_rec; G->npp_len = G->npp_cap = h.npp_len; G->craw_len = G->craw_cap = h.craw_len; G->sv_bytes = h.sv_bytes;
    G->ctg = xmalloc(sizeof(gcontig) * (size_t)(h.n_ctg ? h.n_ctg : 1)); pkg_get(fp, G->ctg, sizeof(gcontig) * (size_t)h.n_ctg);
    G->fp = xmalloc(sizeof(gfpoint) * (size_t)(h.n_fp ? h.n_fp : 1)); pkg_get(fp, G->fp, sizeof(gfpoint) * (size_t)h.n_fp);
    G->npp_off = xmalloc(sizeof(int64_t) * ((size_t)h.n_npp + 1)); pkg_get(fp, G->npp_off, sizeof(int64_t) * ((size_t)h.n_npp + (h.n_npp ? 1 : 0)));
    G->npp_rec = xmalloc(sizeof(int32_t) * (size_t)(h.n_npp ? h.n_npp : 1)); pkg_get(fp, G->npp_rec, sizeof(int32_t) * (size_t)h.n_npp);
    G->npp_raw = xmalloc((size_t)h.npp_len + 1); pkg_get(fp, G->npp_raw, (size_t)h.npp_len);
    G->cand_rec = xmalloc(sizeof(int32_t) * (size_t)(h.n_cand ? h.n_cand : 1)); pkg_get(fp, G->cand_rec, sizeof(int32_t) * (size_t)h.n_cand);
    G->craw_off = xmalloc(sizeof(int64_t) * ((size_t)h.n_cand + 1)); pkg_get(fp, G->craw_off, sizeof(int64_t) * ((size_t)h.n_cand + (h.n_cand ? 1 : 0)));
    G->craw = xmalloc((size_t)h.craw_len + 1); pkg_get(fp, G->craw, (size_t)h.craw_len);
    if (h.has_range) { G->sv_range = xmalloc(sizeof(int32_t) * (size_t)(h.n_cand ? h.n_cand : 1)); pkg_get(fp, G->sv_range, sizeof(int32_t) * (size_t)h.n_cand); }
    fclose(fp);
    const size_t n = (size_t)G->sv_n, ns = n * IM_MAX_EV;
    const size_t bytes[10] = { (size_t)G->sv_bytes, 8 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * ns, 4 * ns, 4 * ns, 4 * n };
    char* at = slab;
    for (int k = 0; k < 10; k++) { G->sv[k] = at; at += (bytes[k] + 255) & ~(size_t)255; }
    return G;
}

/* Every rank has walked what it walks.  ONE all-gather tells all ranks what every claim's walk left (sizes, aborted or not), then
 * the groups walked for other ranks' contigs go to their owners in ONE group of RCCL sends / receives, device memory to device
 * memory -- claim by claim in file order on every rank, a group's host part (uploaded into a device block for the trip) in front
 * of its device arrays.  out_G[ci] = the arrived group for the claims this rank owns and another walked (NULL: aborted).
 * Returns the first claim whose walk was aborted anywhere (n_claims: none): every rank knows it, before any later collective. */
static int mg_ship_groups(mgpu* m, driver* d, int n_claims, const uint8_t* walk_aborted, pgroup** out_G)
{
    typedef struct { int64_t blob_len, slab_bytes, flags; } row;      /* flags: 1 walked by the sender, 2 aborted */
    const size_t nrow = (size_t)(n_claims ? n_claims : 1), tab = sizeof(row) * nrow;
    row* mine = xcalloc(1, tab);
    row* all = xmalloc(tab * (size_t)m->world);
    for (int ci = 0; ci < n_claims; ci++) {
        if (m->claim_walker[ci] != m->rank) continue;
        mine[ci].flags = 1 | (walk_aborted[ci] ? 2 : 0);
        if (!g_parcel[ci].walked) continue;                             /* stays here: this rank owns it */
        mine[ci].blob_len = (int64_t)g_parcel[ci].blob_len; mine[ci].slab_bytes = (int64_t)g_parcel[ci].slab_bytes;
        if (g_parcel[ci].aborted) mine[ci].flags |= 2;
    }
    mg_allgather(m, d, mine, all, tab);
    int first_abort = n_claims;
    int32_t n_op = 0;
    const size_t cap_op = 4 * ((size_t)n_claims + 1);
    int32_t* dir = xmalloc(sizeof(int32_t) * cap_op); int32_t* peer = xmalloc(sizeof(int32_t) * cap_op);
    void** dev = xmalloc(sizeof(void*) * cap_op); size_t* bytes = xmalloc(sizeof(size_t) * cap_op);
    void** out_blob = xcalloc((size_t)n_claims + 1, sizeof(void*));    /* device blocks the host parts travel in: outgoing, */
    void** in_blob = xcalloc((size_t)n_claims + 1, sizeof(void*));     /* incoming, */
    void** in_slab = xcalloc((size_t)n_claims + 1, sizeof(void*));     /* and the device arrays as they arrive */
    for (int ci = 0; ci < n_claims; ci++) {
        const int w = m->claim_walker[ci], o = m->claim_owner[ci];
        const row* r = &all[(size_t)w * nrow + (size_t)ci];
        forceassert(r->flags & 1);
        if ((r->flags & 2) && ci < first_abort) first_abort = ci;
        if (r->blob_len == 0) continue;                                  /* stays where it was walked */
        const size_t bl = ((size_t)r->blob_len + 255) & ~(size_t)255;
        if (w == m->rank) {
            if (im_dev_alloc(d->gpu, bl, &out_blob[ci]) != IM_OK) fatalf("im_dev_alloc: %s", im_last_error(d->gpu));
            if (im_dev_upload(d->gpu, out_blob[ci], g_parcel[ci].blob, g_parcel[ci].blob_len) != IM_OK) fatalf("im_dev_upload: %s", im_last_error(d->gpu));
            dir[n_op] = 0; peer[n_op] = o; dev[n_op] = out_blob[ci]; bytes[n_op] = bl; n_op++;
            if (r->slab_bytes) { dir[n_op] = 0; peer[n_op] = o; dev[n_op] = g_parcel[ci].slab; bytes[n_op] = (size_t)r->slab_bytes; n_op++; }
        }
        if (o == m->rank) {
            if (im_dev_alloc(d->gpu, bl, &in_blob[ci]) != IM_OK) fatalf("im_dev_alloc: %s", im_last_error(d->gpu));
            dir[n_op] = 1; peer[n_op] = w; dev[n_op] = in_blob[ci]; bytes[n_op] = bl; n_op++;
            if (r->slab_bytes) {
                if (im_dev_alloc(d->gpu, (size_t)r->slab_bytes, &in_slab[ci]) != IM_OK) fatalf("im_dev_alloc: %s", im_last_error(d->gpu));
                dir[n_op] = 1; peer[n_op] = w; dev[n_op] = in_slab[ci]; bytes[n_op] = (size_t)r->slab_bytes; n_op++;
            }
        }
    }
    void* st = im_ctx_stream(d->gpu);
    mg_arm("the walked groups of other ranks");
    if (im_comm_exchange(m->comm, n_op, dir, peer, dev, bytes, st) != IM_OK) fatalf("im_comm_exchange: %s", im_comm_last_error());
    if (im_stream_sync(d->gpu, st) != IM_OK) fatalf("im_stream_sync: %s", im_last_error(d->gpu));
    mg_disarm();
    for (int ci = 0; ci < n_claims; ci++) {
        const int w = m->claim_walker[ci], o = m->claim_owner[ci];
        const row* r = &all[(size_t)w * nrow + (size_t)ci];
        if (r->blob_len == 0) continue;
        if (o == m->rank) {
            char* host = xmalloc((size_t)r->blob_len + 8);
            if (im_dev_download(d->gpu, host, in_blob[ci], (size_t)r->blob_len) != IM_OK) fatalf("im_dev_download: %s", im_last_error(d->gpu));
            out_G[ci] = package_unpack(d, host, (size_t)r->blob_len, in_slab[ci]);
            free(host);
            im_dev_free(d->gpu, in_blob[ci]);
        }
        if (w == m->rank) {
            im_dev_free(d->gpu, out_blob[ci]);
            if (g_parcel[ci].slab) im_dev_free(d->gpu, g_parcel[ci].slab);
            free(g_parcel[ci].blob);
            memset(&g_parcel[ci], 0, sizeof g_parcel[ci]);
        }
    }
    free(mine); free(all); free(dir); free(peer); free(dev); free(bytes); free(out_blob); free(in_blob); free(in_slab);
    return first_abort;
}

/* ONE-PASS mode, once the insert lengths are known: every candidate's range[1] from its own record */
static int32_t* group_ranges(driver* d, pgroup* G)
{
    int32_t* range = xmalloc(sizeof(int32_t) * (size_t)(G->n_cand ? G->n_cand : 1));
    for (int32_t j = 0; j < G->n_cand; j++) {
        bam_record b;
        bam_record_view(G->craw + G->craw_off[j], (int32_t)(G->craw_off[j + 1] - G->craw_off[j]), &b);
        range[j] = record_range(d, &b)[1];
    }
    return range;
}

/* ONE walk per rank: the log the ranks exchange (mg_exchange_logs) is made of what the walk itself left in the groups -- the counted
 * reads of every piece, the records of not-proper pairs (kept whole for the pair table: the events are read off them), and, without
 * a configuration file, the insert-size extrema per read group and the coverage spans the walkers collected on the way.  Same
 * layout as the pre-walk's log (mg_prewalk); a claim whose walk met a record the reference dies on leaves blocks that say so. */
static void mg_log_from_groups(mgpu* m, driver* d, walkpool_t* o, int estimate, mgbuf* out)
{
    mg_rg* rgs = xcalloc(MG_MAX_RG, sizeof(mg_rg));
    int n_rg = 0, n_mine = 0;
    covlist cov;
    cov_init(&cov, d->hdr->n_targets);
    mgbuf_take(out, 4 * (MG_HEAD_WORDS + (size_t)MG_MAX_RG * MG_RG_WORDS));
    for (int ci = 0; ci < o->n_claims; ci++) {
        if (m->claim_walker[ci] != m->rank) continue;
        const claim_t* c = &o->claims[ci];
        pgroup* G = c->G;
        if (!G) {
            /* not survived (or never reached behind such a claim): its pieces say so */
            for (int k = 0; k < c->count; k++) {
                int32_t* hd = mgbuf_take(out, 20);
                hd[0] = o->pieces[c->first + k].index; hd[1] = -1; hd[2] = -1;
                n_mine++;
            }
            continue;
        }
        if (estimate) {
            for (int k = 0; k < G->n_rgs; k++) {
                mg_rg* t = &rgs[mg_rg_index(rgs, &n_rg, G->rgs[k].name)];
                const rgstat_t* g = &G->rgs[k];
                if (!t->seen) { t->seen = 1; t->min = g->min; t->max = g->max; t->first_tid = g->first_tid; t->first_rec = g->first_rec; }
                else {
                    if (g->min < t->min) t->min = g->min;
                    if (g->max > t->max) t->max = g->max;
                    if (g->first_tid < t->first_tid || (g->first_tid == t->first_tid && g->first_rec < t->first_rec)) { t->first_tid = g->first_tid; t->first_rec = g->first_rec; }
                }
            }
            if (G->cov.sum) {
                cov_close(&G->cov);
                for (int32_t t = 0; t < cov.nt; t++) cov.sum[t] += G->cov.sum[t];
                for (int64_t k = 0; k < G->cov.n; k++) cov_push(&cov, G->cov.seg[k]);
            }
        }
        int32_t k = 0;
        for (int pi = 0; pi < G->n_ctg; pi++) {
            const gcontig* cg = &G->ctg[pi];
            const size_t head_at = out->n;
            { int32_t* hd = mgbuf_take(out, 20); hd[0] = cg->piece; }
            const size_t ev_at = out->n;
            int32_t n_ev = 0;
            for (; k < G->n_npp && G->npp_rec[k] <= cg->rec1; k++) {
                bam_record b;
                bam_record_view(G->npp_raw + G->npp_off[k], (int32_t)(G->npp_off[k + 1] - G->npp_off[k]), &b);
                /* what src/indelminer.c:516-522 asks of a record apart from |isize| > range[1], which waits for the final table */
                if (!(((b.flag & 0x10) != 0) != ((b.flag & 0x20) != 0) && (uint32_t)abs(b.isize) < O.maxpedelsize)) continue;
                const uint8_t* rg = bam_aux_find(&b, "RG");
                const int gi = mg_rg_index(rgs, &n_rg, rg ? bam_aux_str(rg) : "generic");
                const size_t nl = (size_t)b.l_qname;
                int32_t* ev = mgbuf_take(out, 16 + ((nl + 3) & ~(size_t)3));
                ev[0] = b.pos; ev[1] = abs(b.isize); ev[2] = (int32_t)(G->npp_rec[k] - cg->rec0);
                ev[3] = (b.pos < b.mpos ? 1 : 0) | (gi << 8) | ((int32_t)nl << 16);
                memcpy(ev + 4, BAMR_QNAME(&b), nl);
                n_ev++;
            }
            int32_t* hd = (int32_t*)(out->p + head_at);
            const int64_t counted = cg->cn1 - cg->cn0;
            hd[1] = (int32_t)(counted & 0xffffffff); hd[2] = (int32_t)(counted >> 32); hd[3] = n_ev; hd[4] = (int32_t)(out->n - ev_at);
            n_mine++;
        }
    }
    {
        const int32_t ntg = d->hdr->n_targets;
        int32_t* cw = mgbuf_take(out, 4 * (1 + 2 * (size_t)ntg + 3 * (size_t)cov.n));
        cw[0] = (int32_t)cov.n;
        for (int32_t t = 0; t < ntg; t++) { cw[1 + 2 * t] = (int32_t)(uint32_t)cov.sum[t]; cw[2 + 2 * t] = (int32_t)(uint32_t)(cov.sum[t] >> 32); }
        int32_t* sg = cw + 1 + 2 * (size_t)ntg;
        for (int64_t k = 0; k < cov.n; k++) { *sg++ = cov.seg[k].tid; *sg++ = cov.seg[k].beg; *sg++ = cov.seg[k].end; }
        cov_free(&cov);
    }
    int32_t* w = (int32_t*)out->p;
    w[0] = MG_MAGIC; w[1] = n_rg; w[2] = n_mine; w[3] = (int32_t)(out->n & 0xffffffff); w[4] = (int32_t)((uint64_t)out->n >> 32);
    for (int k = 0; k < n_rg; k++) {
        int32_t* g = w + MG_HEAD_WORDS + (size_t)k * MG_RG_WORDS;
        memcpy(g, rgs[k].name, 48);
        g[12] = rgs[k].min; g[13] = rgs[k].max; g[14] = rgs[k].first_tid; g[15] = (int32_t)(rgs[k].first_rec >> 32); g[16] = rgs[k].seen; g[17] = (int32_t)(uint32_t)rgs[k].first_rec;
    }
    free(rgs);
}

/* what every rank does with the exchanged logs before it stages anything */
static void mg_after_logs(mgpu* m, driver* d)
{
    if (m->cross) {
        /* A first mate left waiting in one contig meets a record of the same name in a later one: the reference's one
         * pair table pairs them across contigs (readpairs is never reset, src/indelminer.c), so the contigs of this
         * input are not independent.  Every rank sees that in the exchanged logs; the run goes to ONE process that
         * serves one table record by record, the other ranks have nothing to add. */
        if (m->rank != 0) { im_comm_destroy(m->comm); fflush(stderr); _exit(EXIT_SUCCESS); }
        fprintf(stderr, "indelminer: read names are shared between contigs (pairs across contigs in the one pair table): one process takes the run\n");
        mg_discard_dir(m);
        mg_restore_stdout(m);
        handoff_to_host_child();
    }
    if (O.configfile == NULL && m->rank == 0) {
        fprintf(stderr, "\nRead-group\tMin-value\tMax-value (estimated over all ranks' contigs)\n");
        for (int j = 0; j < g_rg_n; j++) fprintf(stderr, "%s\t%d\t%d\n", g_rg_name[j], g_rg_range[j][0], g_rg_range[j][1]);
        cov_print_table(d->hdr);
    }
}

/* the driver's pair table, emptied (a contig begins: what earlier contigs left waiting reaches it as the marker floor, not as entries) */
static void pair_table_clear(driver* d)
{
    while (d->n_live > 0) {
        evidence_t* e = d->live[d->n_live - 1];
        live_del(d, e);
        qhash_remove(d->readpairs, e->qname, (int)strlen(e->qname) + 1);
        evidence_free(e);
    }
    d->live_changed = 0;
}

/* The kept records of not-proper pairs through the pair table (src/indelminer.c:516-615), piece by piece, on the main thread:
 * the table's entries carry over from one piece of a contig to the next.  Completed pairs join the group's paired-read entries
 * (behind the pending ones of earlier pieces, which stage_take_front put there), the table's smallest waiting start is logged
 * where it moves (find_marker, 211-233). */
static void group_pair_table(driver* d, pgroup* G)
{
    int32_t k = 0;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        gcontig* cg = &G->ctg[ci];
        G->cur_ctg = ci;
        if (cg->first) pair_table_clear(d);
        d->live_changed = 0;
        cg->lm_init = find_marker_live(d);
        cg->pe0 = ci == 0 ? 0 : G->n_pe; cg->lm0 = G->n_lm; cg->dn0 = G->dn_len;
        for (; k < G->n_npp && G->npp_rec[k] <= cg->rec1; k++) {
            bam_record b;
            bam_record_view(G->npp_raw + G->npp_off[k], (int32_t)(G->npp_off[k + 1] - G->npp_off[k]), &b);
            host_discordant(d, G, &b, G->npp_rec[k]);
        }
        cg->pe1 = G->n_pe; cg->lm1 = G->n_lm; cg->dn1 = G->dn_len;
        cg->left_min = find_marker_live(d);
        cg->sn0 = cg->sn1 = G->sn_len;
        if (cg->last) group_log_waiting(d, G, cg);
    }
}
