/* host_walk.c -- part of the indelminer host driver (one translation unit: imhost.c includes the parts in order, so that the
 * reference-shaped helpers can stay static).  Here: evidence that crosses piece boundaries, the walker pool (pieces, claims), run_pipeline (the main thread's loop over the
 * groups in file order, replay workers). */

/* ---- evidence that crosses piece boundaries ---- */

typedef struct {
    carry_list live;            /* pending: takes part in the next piece's flushes */
    carry_list frozen;          /* pending with b2 >= the contig's marker floor: no flush before the contig's last can consume it
                                 * (every marker is <= the floor, so it is a cutting candidate of every flush that sees it); it
                                 * waits for the last piece, and a single entry carries the smallest (b1,b2) among them so far */
    uint64_t frozen_min;
    int tid;
} carry_t;

static void carry_push(carry_list* l, const carry_item* it)
{
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 256; l->v = xrealloc(l->v, sizeof(carry_item) * (size_t)l->cap); }
    l->v[l->n++] = *it;
}

static evidence_t* phantom_entry(uint64_t key)
{
    evidence_t* e = xcalloc(1, sizeof *e);
    e->type = EV_PHANTOM; e->cls = CLS_DELETION;
    e->b1 = (int32_t)(key >> 32); e->b2 = (int32_t)(uint32_t)key;
    e->live_slot = -1;
    return e;
}

/* Before a group's pair table and stage: what the earlier pieces of its contig left pending goes in front -- split-read candidates
 * into front[], paired-read entries to the head of pe[] -- numbered 0 .. n_virt - 1 in order of arrival.  The last piece of a
 * contig takes the frozen entries too; any other piece takes one entry that stands for them. */
static void stage_take_front(pgroup* G, carry_t* C)
{
    const gcontig* cg = &G->ctg[0];
    G->n_front = 0; G->n_pe_front = 0; G->n_virt = 0; G->phantom = 0;
    if (cg->first) { C->live.n = 0; C->frozen.n = 0; C->frozen_min = ~0ull; C->tid = cg->tid; return; }
    forceassert(G->n_ctg == 1 && C->tid == cg->tid && G->n_pe == 0);
    const int take_frozen = cg->last;
    const int32_t n_all = C->live.n + (take_frozen ? C->frozen.n : 0);
    G->front = xrealloc(G->front, sizeof(carry_item) * (size_t)(n_all ? n_all : 1));
    G->front_virt = xrealloc(G->front_virt, sizeof(int32_t) * (size_t)(n_all ? n_all : 1));
    if (n_all + 1 > G->cap_pe) {
        G->cap_pe = n_all + 1024;
        G->pe = xrealloc(G->pe, sizeof(evidence_t*) * (size_t)G->cap_pe);
        G->pe_rec = xrealloc(G->pe_rec, sizeof(int64_t) * (size_t)G->cap_pe);
    }
    /* both lists are in order of arrival: merge */
    int32_t a = 0, b = 0, v = 0;
    const int32_t nb = take_frozen ? C->frozen.n : 0;
    while (a < C->live.n || b < nb) {
        const carry_item* it = (b >= nb || (a < C->live.n && C->live.v[a].when <= C->frozen.v[b].when)) ? &C->live.v[a++] : &C->frozen.v[b++];
        if (it->g) { G->front[G->n_front] = *it; G->front_virt[G->n_front] = v; G->n_front++; }
        else {
            it->pe->arrival = (int64_t)v * 8 + 7;
            G->pe[G->n_pe] = it->pe; G->pe_rec[G->n_pe] = -1; G->n_pe++;
        }
        v++;
    }
    if (!take_frozen && C->frozen_min != ~0ull) {
        G->pe[G->n_pe] = phantom_entry(C->frozen_min); G->pe_rec[G->n_pe] = -1; G->n_pe++;
        G->phantom = 1;
    }
    G->n_pe_front = G->n_pe;
    G->n_virt = v;
    C->live.n = 0;
    if (take_frozen) { C->frozen.n = 0; C->frozen_min = ~0ull; }
}

/* After a group's stage: what no flush of it has consumed.  Nothing is left behind the last piece of a contig (its last flush takes
 * everything).  floor = the marker floor of the contig (group_resolve_flushes): an entry with b2 >= floor is frozen. */
static void stage_leftovers(pgroup* G, carry_t* C, int floor)
{
    const gcontig* cg = &G->ctg[G->n_ctg - 1];
    if (cg->last) return;
    forceassert(G->n_ctg == 1);
    const int32_t nc = G->n_front + G->sv_n;
    const int32_t frozen0 = C->frozen.n;
    for (int32_t q = 0; q < nc; q++) {
        carry_item live, froz;
        int nl = 0, nf = 0;
        for (int k = 0; k < IM_MAX_EV; k++) {
            const size_t sl = (size_t)q * IM_MAX_EV + (size_t)k;
            live.cls[k] = froz.cls[k] = -1; live.b1[k] = froz.b1[k] = 0; live.b2[k] = froz.b2[k] = 0;
            if (G->s_cls[sl] < 0 || G->cons_sr[sl] != 0) continue;
            carry_item* to = G->s_b2[sl] >= floor ? &froz : &live;
            to->cls[k] = G->s_cls[sl]; to->b1[k] = G->s_b1[sl]; to->b2[k] = G->s_b2[sl];
            if (to == &froz) {
                nf++;
                const uint64_t key = ((uint64_t)(uint32_t)G->s_b1[sl] << 32) | (uint32_t)G->s_b2[sl];
                if (key < C->frozen_min) C->frozen_min = key;
            } else nl++;
        }
        if (!nl && !nf) continue;
        carry_item base;
        if (q < G->n_front) base = G->front[q];
        else { base.g = G; base.cand = q - G->n_front; base.pe = NULL; base.when = ((int64_t)G->seq << 32) | (uint32_t)G->cand_rec[q - G->n_front]; }
        if (nl) { live.when = base.when; live.g = base.g; live.cand = base.cand; live.pe = NULL; carry_push(&C->live, &live); }
        if (nf) { froz.when = base.when; froz.g = base.g; froz.cand = base.cand; froz.pe = NULL; carry_push(&C->frozen, &froz); }
    }
    /* the same for the paired-read entries; then each kind's pending items, both in order of arrival, merged into the lists */
    carry_list pl = { NULL, 0, 0 }, pf = { NULL, 0, 0 };
    for (int32_t i = 0; i < G->n_pe; i++) {
        evidence_t* e = G->pe[i];
        if (e->type == EV_PHANTOM || G->cons_pe[i] != 0) continue;
        carry_item it;
        memset(&it, 0, sizeof it);
        it.g = NULL; it.pe = e; it.when = e->when;
        for (int k = 0; k < IM_MAX_EV; k++) it.cls[k] = -1;
        if (e->b2 >= floor) {
            const uint64_t key = ((uint64_t)(uint32_t)e->b1 << 32) | (uint32_t)e->b2;
            if (key < C->frozen_min) C->frozen_min = key;
            carry_push(&pf, &it);
        } else carry_push(&pl, &it);
    }
    /* C->live / C->frozen hold this stage's split-read leftovers from index n0 on (stage_take_front emptied live; frozen keeps
     * what earlier pieces froze, all of which arrived before anything of this piece: front items are never frozen-kind) */
    for (int pass = 0; pass < 2; pass++) {
        carry_list* l = pass ? &C->frozen : &C->live;
        const carry_list* pe = pass ? &pf : &pl;
        const int32_t n0 = pass ? frozen0 : 0;
        if (pe->n == 0) continue;
        const int32_t nsr = l->n - n0;
        carry_item* m = xmalloc(sizeof(carry_item) * (size_t)(nsr + pe->n));
        int32_t a = 0, b = 0, w = 0;
        while (a < nsr || b < pe->n) m[w++] = (b >= pe->n || (a < nsr && l->v[n0 + a].when <= pe->v[b].when)) ? l->v[n0 + a++] : pe->v[b++];
        l->n = n0;
        for (int32_t i = 0; i < w; i++) carry_push(l, &m[i]);
        free(m);
    }
    free(pl.v); free(pf.v);
}

static void walker_adopt_driver(walker_t* W, driver* d)
{
    W->wd = *d;                                 /* shared, read-only from here on: header, index, reference, insert lengths, GPU */
    W->wd.readpairs = qhash_new(4);             /* the pair table is the main thread's (group_pair_table) */
    W->wd.live = NULL; W->wd.n_live = W->wd.cap_live = 0; W->wd.live_changed = 0;
    W->wd.rg_last_val = NULL; W->wd.rg_last_name[0] = 0;
    W->wd.gpu_pending = 0;
}

static void walker_setup(walker_t* W, driver* d)
{
    W->wd.gpu = d->gpu;
    pipe_init(&W->P, &W->wd, 1);
    W->r = bgzf_open(d->bam_name);
    if (!W->r) fatalf("error in opening the file %s", d->bam_name);
    if (W->pool->inflate_workers >= 0) bgzf_set_workers(W->r, W->pool->inflate_workers);
    W->hdr = bam_header_load(W->r);
    if (!W->hdr) fatalf("%s is not a BAM file", d->bam_name);
}

/* one claim: its pieces through the walker's pipeline into a new group, the group's device arrays parked */
static pgroup* walk_claim(walker_t* W, walkpool_t* o, const claim_t* c)
{
    pgroup* G = xcalloc(1, sizeof(pgroup));
    for (int k = 0; k < c->count; k++) pipe_walk_piece(&W->P, G, &o->pieces[c->first + k], W->r);
    DEV_TIMED(pipe_submit(&W->P, G));
    DEV_TIMED(pipe_drain(&W->P, G));
    DEV_TIMED(group_park_device(&W->P, G));
    return G;
}

static void* walker_thread(void* arg)
{
    walker_t* W = arg;
    walkpool_t* o = W->pool;
    driver* d = o->d;
    /* buffers as soon as the GPU context exists -- beside the insert-length pass and the FASTA read of the main thread */
    pthread_mutex_lock(&d->gpu_mu);
    while (!d->ctx_ready) pthread_cond_wait(&d->gpu_cv, &d->gpu_mu);
    pthread_mutex_unlock(&d->gpu_mu);
    if (d->ctx_rc != IM_OK) return NULL;            /* the main thread reports it (gpu_wait) */
    walker_setup(W, d);
    pthread_mutex_lock(&o->mu);
    while (!o->go) pthread_cond_wait(&o->cv, &o->mu);
    pthread_mutex_unlock(&o->mu);
    walker_adopt_driver(W, d);
    const int64_t w_start = wall_ns(), c_start = thread_cpu_ns();
    for (;;) {
        pthread_mutex_lock(&o->mu);
        /* walked groups wait for the main thread with their logs and parked arrays: stay a bounded number of claims ahead of it */
        if (g_timing && (!g_onepass || g_spec_active) && !g_mg && o->next_claim < o->n_claims && o->next_claim >= o->staged + 2 * o->nw + 4) {
            const int64_t t0w = wall_ns();
            while (o->next_claim < o->n_claims && o->next_claim >= o->staged + 2 * o->nw + 4) pthread_cond_wait(&o->cv, &o->mu);
            __atomic_fetch_add(&g_wall_walk_throttled_ns, wall_ns() - t0w, __ATOMIC_RELAXED);
        }
        while ((!g_onepass || g_spec_active) && !g_mg && o->next_claim < o->n_claims && o->next_claim >= o->staged + 2 * o->nw + 4) pthread_cond_wait(&o->cv, &o->mu);
        while (g_mg && o->next_claim < o->n_claims && g_mg->claim_walker[o->next_claim] != g_mg->rank) o->next_claim++;      /* another rank walks it */
        const int ci = o->next_claim < o->n_claims ? o->next_claim++ : -1;
        pthread_mutex_unlock(&o->mu);
        if (ci < 0) {
            __atomic_fetch_add(&g_cpu_walk_ns, thread_cpu_ns() - c_start, __ATOMIC_RELAXED); __atomic_fetch_add(&g_cpu_walk_dev_ns, t_cpu_dev_ns, __ATOMIC_RELAXED);
            __atomic_fetch_add(&g_wall_walk_ns, wall_ns() - w_start, __ATOMIC_RELAXED); __atomic_fetch_add(&g_wall_walk_dev_ns, t_wall_dev_ns, __ATOMIC_RELAXED);
            break;
        }
        claim_t* c = &o->claims[ci];
        volatile int ship = g_mg && (g_mg->claim_owner[ci] != g_mg->rank || g_mg_self_ship);       /* read behind a setjmp */
        if (g_handoff_pool) {
            /* a record the reference dies on ends this walker: the claim is published as it is, marked */
            W->cur_claim = c;
            if (setjmp(W->abort_jmp)) {
                pthread_mutex_lock(&o->mu);
                W->cur_claim->G = NULL; W->cur_claim->aborted = 1; W->cur_claim->walked = 1;
                pthread_cond_broadcast(&o->cv);
                pthread_mutex_unlock(&o->mu);
                return NULL;
            }
            t_abort_jmp = &W->abort_jmp;
        }
        pgroup* G = walk_claim(W, o, c);
        t_abort_jmp = NULL;
        if (g_onepass && g_spec_active && !G->sv_range) G->sv_range = group_ranges(&W->wd, G);    /* the provisional table is there: this candidate's range[1] now */
        if (g_mg && o->serial) { int64_t nr = 0; group_flush_points(G, &nr); G->from_package = 1; }      /* annotate mode: the counter in front of every piece is known (mg_exchange) */
        (void)ship;
        pthread_mutex_lock(&o->mu);
        c->G = G; c->walked = 1;
        pthread_cond_broadcast(&o->cv);
        pthread_mutex_unlock(&o->mu);
    }
    return NULL;
}

typedef struct { struct walkpool_s* o; int first, step; driver rd; pthread_t th; } apply_job;
static void* apply_thread(void* arg)
{
    apply_job* j = arg;
    for (int ci = j->first; ci < j->o->n_claims; ci += j->step) {
        pgroup* G = j->o->claims[ci].G;
        if (!G) continue;                   /* several ranks: another rank's walk, or a walk that was not survived */
        G->sv_range = group_ranges(&j->rd, G);
    }
    return NULL;
}

/* groups of a contig are freed together: pending evidence points back at the groups it came from */
static void groups_free_chain(pgroup* G)
{
    while (G) { pgroup* n = G->next_of_contig; group_free(G); free(G); G = n; }
}

static void job_part_log(void* arg, int32_t tid, int first, size_t off)
{
    rjob_t* J = arg;
    if (J->n_part == J->cap_part) { J->cap_part = J->cap_part ? J->cap_part * 2 : 4; J->part = xrealloc(J->part, sizeof(rpart_t) * (size_t)J->cap_part); }
    J->part[J->n_part].tid = tid; J->part[J->n_part].first = first; J->part[J->n_part].off = off; J->n_part++;
}
/* a finished job's output, in file order: to the output stream, or (multi-GPU) stretch by stretch to its contigs' part files */
static void job_emit(rjob_t* P)
{
    if (!g_mg) { if (P->len && fwrite(P->buf, 1, P->len, OUT) != P->len) fatalf("write to stdout failed"); return; }
    for (int i = 0; i < P->n_part; i++) {
        const size_t a = P->part[i].off, z = i + 1 < P->n_part ? P->part[i + 1].off : P->len;
        char path[512];
        mg_path(g_mg, path, sizeof path, "part", P->part[i].tid);
        FILE* fp = fopen(path, P->part[i].first ? "w" : "a");
        if (!fp || (z > a && fwrite(P->buf + a, 1, z - a, fp) != z - a) || fclose(fp) != 0) fatalf("cannot write %s", path);
    }
    free(P->part); P->part = NULL; P->n_part = P->cap_part = 0;
}

static void* replay_thread(void* arg)
{
    replayer_t* R = arg;
    walkpool_t* o = R->pool;
    for (;;) {
        pthread_mutex_lock(&o->mu);
        while (o->next_job >= o->n_jobs && !o->jobs_closed) pthread_cond_wait(&o->cv, &o->mu);
        const int j = o->next_job < o->n_jobs ? o->next_job++ : -1;
        pthread_mutex_unlock(&o->mu);
        if (j < 0) { __atomic_fetch_add(&g_cpu_replay_ns, thread_cpu_ns(), __ATOMIC_RELAXED); break; }
        rjob_t* J = &o->jobs[j];
        {   /* test hook: every other replay takes this much longer, so that replays finish out of order on any machine */
            const char* dl = getenv("INDELMINER_DEBUG_REPLAY_DELAY_MS");
            if (dl && (j & 1) == 0) { struct timespec ts = { atoi(dl) / 1000, (long)(atoi(dl) % 1000) * 1000000L }; nanosleep(&ts, NULL); }
        }
        t_out = open_memstream(&J->buf, &J->len);
        if (!t_out) fatalf("cannot buffer the output of a group");
        t_part_log = g_mg ? job_part_log : NULL; t_part_arg = J;
        group_replay(&R->rd, J->G);
        t_part_log = NULL;
        fclose(t_out);
        t_out = NULL;
        pthread_mutex_lock(&o->mu);
        J->done = 1;
        pthread_cond_broadcast(&o->cv);
        pthread_mutex_unlock(&o->mu);
    }
    return NULL;
}

/* Called as soon as the BAM header and index are known: cuts this process's share of the file into pieces, plans the claims and
 * starts the walkers, which set their buffers up in the background and then wait for run_pipeline's go. */
static walkpool_t* walkpool_start(driver* d)
{
    walkpool_t* o = xcalloc(1, sizeof *o);
    o->d = d;
    pthread_mutex_init(&o->mu, NULL); pthread_cond_init(&o->cv, NULL);
    const int32_t nt = d->hdr->n_targets;
    /* annotate mode shares the known-variant list with the replay and skips contigs without variants: one walker, one whole
     * contig per claim, walked by the main thread only after the previous one has been replayed */
    o->serial = g_vcfname != NULL;
    const char* e = getenv("INDELMINER_WALKERS");
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    {   /* the cores this process may use, not the machine's (a container's CPU quota) */
        FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r");
        long quota = 0, period = 0;
        if (fp) { if (fscanf(fp, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < ncpu) ncpu = quota / period; fclose(fp); }
        if (ncpu < 1) ncpu = 1;
    }
    /* one walker per core when there are pieces enough to go round (each then inflates its own blocks: no hand-over between
     * threads); with few pieces, few walkers and the other cores as inflate workers of their readers (set below) */
    int nw = e ? atoi(e) : (int)(ncpu > 16 ? 16 : ncpu);
    if (o->serial || nw < 1) nw = 1;
    if (nw > 32) nw = 32;
    int64_t total_bytes = 0, total_len = 0;
    for (int32_t i = 0; i < nt; i++) {
        if (g_mg && g_mg->skip && g_mg->skip[i]) continue;
        if (g_region_tid >= 0 && i != g_region_tid) continue;
        total_bytes += bai_contig_bytes(d->idx, i); total_len += d->hdr->target_len[i];
    }
    if (g_mg) { total_bytes /= g_mg->world; total_len /= g_mg->world; }      /* a rank's share: the plan below is the whole run's, the same on every rank */
    /* A walker costs its set-up (a pinned ring, device arrays, a stream: tens of milliseconds, serialised in the driver) before it
     * delivers anything: one walker per ~48 MB of BAM, and a small ring, for inputs that a handful of walkers finish in a few
     * hundred milliseconds anyway (measured on the 430 MB BAM of BASELINE configs[2]: 4 / 8 / 16 walkers 1.43 / 1.28 / 1.68 s) */
    if (!e && !o->serial) {
        const int by_size = (int)(total_bytes / (48 << 20));
        if (by_size < nw) nw = by_size < 2 ? 2 : by_size;
    }
    if (total_bytes < ((int64_t)2 << 30)) { g_chunk_bytes = 16u << 20; g_nchunk = 2; }
    /* A run of a few tenths of a second is mostly set-up: a HIP stream costs ~16 ms to create and pinning ~0.4 ms per MB, one after
     * the other in the driver (INDELMINER_TIMING: "a pipeline's buffers").  Below 64 MB of BAM the walkers and the stage take the
     * context's stream (nothing is there to overlap with) and ring chunks of 8 MB. */
    if (total_bytes < ((int64_t)64 << 20) && !g_mg) { g_chunk_bytes = 8u << 20; g_small_input = 1; }
    if (getenv("INDELMINER_CHUNK_MB") && atoi(getenv("INDELMINER_CHUNK_MB")) >= 1 && atoi(getenv("INDELMINER_CHUNK_MB")) <= 256) g_chunk_bytes = (uint32_t)atoi(getenv("INDELMINER_CHUNK_MB")) << 20;
    if (getenv("INDELMINER_CHUNKS") && atoi(getenv("INDELMINER_CHUNKS")) >= 2 && atoi(getenv("INDELMINER_CHUNKS")) <= PIPE_NCHUNK) g_nchunk = atoi(getenv("INDELMINER_CHUNKS"));
    /* pieces: a contig is cut where its compressed bytes cross multiples of the piece size -- about 1/(8 walkers) of the file, at
     * least 8 MB of it (a stage and a replay have fixed costs per group), so that large contigs spread over all walkers */
    int64_t piece_bytes = total_bytes / (8 * (int64_t)nw);
    if (piece_bytes > (64 << 20)) piece_bytes = 64 << 20;       /* a piece is what the walkers' ends can differ by: 64 MB is a third of a second */
    if (piece_bytes < (8 << 20)) piece_bytes = 8 << 20;
    if (getenv("INDELMINER_PIECE_BYTES")) piece_bytes = atoll(getenv("INDELMINER_PIECE_BYTES"));
    if (o->serial) piece_bytes = 0;
    int cap = 0;
    for (int32_t i = 0; i < nt; i++) {
        if (g_mg && g_mg->skip && g_mg->skip[i]) continue;          /* annotate mode: no known variant on it, nobody walks it */
        if (g_region_tid >= 0 && i != g_region_tid) continue;
        int32_t cuts[4096];
        int nc = piece_bytes > 0 ? bai_split_points(d->idx, i, d->hdr->target_len[i], piece_bytes, cuts, 4096) : 0;
        int32_t lo = 0, hi = d->hdr->target_len[i];
        if (g_region_tid >= 0) {
            /* -c: the stretch [beg, end) only; cuts outside it go */
            lo = g_region_beg < 0 ? 0 : g_region_beg; hi = g_region_end;
            int m = 0;
            for (int k = 0; k < nc; k++) if (cuts[k] > lo && cuts[k] < hi) cuts[m++] = cuts[k];
            nc = m;
        }
        if (o->n_pieces + nc + 1 > cap) { cap = (cap + nc + 1) * 2; o->pieces = xrealloc(o->pieces, sizeof(piece_t) * (size_t)cap); }
        const int64_t w = bai_contig_bytes(d->idx, i);
        for (int k = 0; k <= nc; k++) {
            piece_t* pc = &o->pieces[o->n_pieces++];
            pc->tid = i; pc->beg = k ? cuts[k - 1] : lo; pc->end = k < nc ? cuts[k] : hi;
            pc->first = k == 0; pc->last = k == nc; pc->weight = w / (nc + 1);
            pc->overlap = g_region_tid >= 0 && k == 0;
            pc->index = o->n_pieces - 1;
        }
    }
    /* claims: a piece of a cut contig on its own; whole small contigs together up to about a piece's worth (the stage and the
     * replay have fixed costs per group) */
    int64_t claim_len = total_len / (4 * (int64_t)nw);
    if (claim_len < 2000000) claim_len = 2000000;
    if (getenv("INDELMINER_CLAIM_BASES")) claim_len = atoll(getenv("INDELMINER_CLAIM_BASES"));
    o->claims = xcalloc((size_t)(o->n_pieces ? o->n_pieces : 1), sizeof(claim_t));
    for (int k = 0; k < o->n_pieces;) {
        claim_t* c = &o->claims[o->n_claims++];
        c->first = k;
        const piece_t* p0 = &o->pieces[k];
        int64_t len = 0;
        if (!(p0->first && p0->last)) { k++; c->count = 1; continue; }
        do { len += d->hdr->target_len[o->pieces[k].tid]; k++; }
        while (!o->serial && k < o->n_pieces && o->pieces[k].first && o->pieces[k].last && len + d->hdr->target_len[o->pieces[k].tid] <= claim_len &&
               (!g_mg || g_mg->owner[o->pieces[k].tid] == g_mg->owner[p0->tid]));
        c->count = k - c->first;
    }
    if (g_mg) {
        /* Who walks what.  A contig's stage and replay are its owner's (mg_plan: contigs to ranks by size); its pieces are WALKED --
         * read, inflated, triaged -- by whichever rank has done the least so far, so that one large contig, or fewer contigs than
         * GPUs, still keeps every rank's cores and GPU busy.  The walked group then travels to the owner (package_write). */
        mgpu* m = g_mg;
        m->claim_owner = xmalloc(sizeof(int32_t) * (size_t)(o->n_claims ? o->n_claims : 1));
        m->claim_walker = xmalloc(sizeof(int32_t) * (size_t)(o->n_claims ? o->n_claims : 1));
        m->piece_walker = xmalloc(sizeof(int32_t) * (size_t)(o->n_pieces ? o->n_pieces : 1));
        int64_t* load = xcalloc((size_t)m->world, sizeof(int64_t));
        const char* how = getenv("INDELMINER_MG_WALK");
        for (int ci = 0; ci < o->n_claims; ci++) {
            const claim_t* c = &o->claims[ci];
            int64_t w = 1;
            for (int k = 0; k < c->count; k++) w += o->pieces[c->first + k].weight;
            const int own = m->owner[o->pieces[c->first].tid];
            int best = own;
            for (int r = 0; r < m->world; r++) if (load[r] + w / 8 < load[best]) best = r;      /* the owner unless somebody is clearly idler */
            if (o->serial || (how && strcmp(how, "owner") == 0)) best = own;
            m->claim_owner[ci] = own; m->claim_walker[ci] = best;
            load[best] += w;
            for (int k = 0; k < c->count; k++) m->piece_walker[c->first + k] = best;
            if (best != own) m->split = 1;
        }
        /* INDELMINER_MG_FORCE_SPLIT=1 (tests): treat the run as one whose contigs were walked by several ranks -- the replays wait
         * for the sum of the depth arrays (im_depth_allreduce) -- also with one rank, where the sum changes nothing */
        if (getenv("INDELMINER_MG_FORCE_SPLIT")) m->split = 1;
        /* INDELMINER_MG_SELF_SHIP=1 (tests, one rank): every claim takes the road of a claim walked for another rank -- parcelled,
         * through the send / receive group (to this rank itself), unpacked */
        g_mg_self_ship = getenv("INDELMINER_MG_SELF_SHIP") != NULL && !o->serial;
        if (g_mg_self_ship) m->split = 1;
        g_parcel = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof(mg_parcel));
        free(load);
    }
    if (nw > o->n_claims) nw = o->n_claims ? o->n_claims : 1;
    o->nw = nw;
    o->inflate_workers = getenv("INDELMINER_THREADS") ? -1 : (o->n_claims >= 2 * nw && nw >= ncpu - 1 ? 0 : (int)((ncpu - nw + nw - 1) / nw));
    if (o->inflate_workers > 4) o->inflate_workers = 4;
    if (!getenv("INDELMINER_THREADS") && o->inflate_workers < 1 && nw < ncpu - 1) o->inflate_workers = 1;
    o->w = xcalloc((size_t)nw, sizeof(walker_t));
    for (int i = 0; i < nw; i++) o->w[i].pool = o;
    if (!o->serial)
        for (int i = 0; i < nw; i++)
            if (pthread_create(&o->w[i].th, NULL, walker_thread, &o->w[i]) != 0) fatalf("cannot start a walking thread");
    return o;
}

/* What the one-pass walk has learnt so far: per read group the insert-size extrema with the first sighting, the spans for the
 * coverage table (copied out of the groups, which go when their contig has been replayed), and what it has printed so far. */
typedef struct { mg_rg* rg; int n_rg, cap_rg; covlist cov; mg_rg* prov; int n_prov; char* out; size_t out_len, out_cap; int unknown_counted_rg; } spec_t;
static void spec_collect(spec_t* S, driver* d, pgroup* G)
{
    if (!G) return;
    if (S->n_rg + G->n_rgs > S->cap_rg) { S->cap_rg = (S->cap_rg + G->n_rgs) * 2 + 16; S->rg = xrealloc(S->rg, sizeof(mg_rg) * (size_t)S->cap_rg); }
    for (int k = 0; k < G->n_rgs; k++) {
        mg_rg* m = &S->rg[S->n_rg++];
        memset(m, 0, sizeof *m);
        snprintf(m->name, sizeof m->name, "%s", G->rgs[k].name);
        m->min = G->rgs[k].min; m->max = G->rgs[k].max; m->first_tid = G->rgs[k].first_tid; m->first_rec = G->rgs[k].first_rec; m->seen = 1;
    }
    if (!S->cov.sum) cov_init(&S->cov, d->hdr->n_targets);
    if (G->cov.sum) {
        cov_close(&G->cov);
        for (int32_t t = 0; t < S->cov.nt; t++) S->cov.sum[t] += G->cov.sum[t];
        for (int64_t k = 0; k < G->cov.n; k++) cov_push(&S->cov, G->cov.seg[k]);
    }
}
/* the collected extrema, read groups in the order one sequential pass meets them, into the insert-length table (table = 1) or
 * only into a list (the check at the end); returns the list's length, *out = the list */
static int spec_merge(const spec_t* S, mg_rg** out)
{
    mg_rg* all = xmalloc(sizeof(mg_rg) * (size_t)(S->n_rg ? S->n_rg : 1));
    memcpy(all, S->rg, sizeof(mg_rg) * (size_t)S->n_rg);
    mg_rg* merged = xcalloc((size_t)(S->n_rg ? S->n_rg : 1), sizeof(mg_rg));
    const int n = merge_rgs(all, S->n_rg, merged);
    free(all);
    *out = merged;
    return n;
}
static void onepass_print_tables(driver* d, spec_t* S)
{
    fprintf(stderr, "\nRead-group\tMin-value\tMax-value (estimated during the walk)\n");
    for (int j = 0; j < g_rg_n; j++) fprintf(stderr, "%s\t%d\t%d\n", g_rg_name[j], g_rg_range[j][0], g_rg_range[j][1]);
    covlist* one = &S->cov;
    if (!S->cov.sum) cov_init(&S->cov, d->hdr->n_targets);
    cov_means_of_lists(d->hdr->n_targets, &one, 1);
    cov_print_table(d->hdr);
}
static int onepass_table(driver* d, spec_t* S, int print)
{
    mg_rg* merged;
    const int n = spec_merge(S, &merged);
    for (int j = 0; j < n; j++) rg_table_enter(d, &merged[j]);
    S->prov = merged;
    if (print) onepass_print_tables(d, S);
    return n;
}
/* The whole file's list against the provisional one: the same groups in the same order with the same range[1]?  range[1] is what
 * the path reads (src/indelminer.c:520, 583-584, src/alignment.c:775-780); range[0] is only ever printed (and asserted to be
 * <= range[1]), so a smaller minimum found later just goes into the table before it is printed. */
static int spec_holds(driver* d, const spec_t* S)
{
    mg_rg* fin;
    const int n = spec_merge(S, &fin);
    int ok = n == S->n_prov;
    for (int j = 0; ok && j < n; j++) ok = strcmp(fin[j].name, S->prov[j].name) == 0 && fin[j].max == S->prov[j].max;
    if (ok) for (int j = 0; j < n; j++) rg_table_enter(d, &fin[j]);       /* the minima of the whole file */
    free(fin);
    return ok;
}
static void spec_keep_output(spec_t* S, const char* buf, size_t len)
{
    if (S->out_len + len > S->out_cap) { S->out_cap = (S->out_cap + len) * 2 + (1 << 20); S->out = xrealloc(S->out, S->out_cap); }
    memcpy(S->out + S->out_len, buf, len);
    S->out_len += len;
}

/* fetch_func looks the read group of every counted read up (src/indelminer.c:369-376, must_find_hashtable): a group that occurs on
 * counted reads but on no proper pair is not in the estimated table, and the reference dies at that read -- header and earlier
 * flushes out.  The record-at-a-time run reproduces that. */
static void onepass_counted_groups_known(driver* d, walkpool_t* o, int n_claims)
{
    for (int ci = 0; ci < n_claims; ci++) {
        const pgroup* G = o->claims[ci].G;
        for (int k = 0; G && k < G->n_crg; k++)
            if (!qhash_lookup(d->insertlengths, G->crg[k], (int)strlen(G->crg[k]))) {
                if (g_handoff_pool) pipeline_handoff();
                fatalf("did not find %s in the hash", G->crg[k]);
            }
    }
}

static jmp_buf g_mg_split_abort;
static int g_mg_split_armed = 0;

static void run_pipeline(driver* d, walkpool_t* o)
{
    d->pipe_mode = 1;
    g_verify_triage = getenv("INDELMINER_VERIFY_TRIAGE") != NULL;
    if (!g_mg && !getenv("INDELMINER_NO_HANDOFF")) {
        /* from here on this thread prints through the counting stream (see handoff_to_host_child) */
        t_out = out_cookie_open();
        if (t_out) g_handoff_pool = o;
    } else if (g_mg && !getenv("INDELMINER_NO_HANDOFF")) {
        /* multi-GPU: the parts in front of the contig are complete, rank 0 prints them and starts the child (mg_finish) */
        g_handoff_pool = o;
        g_mg_driver = d;
    }
    g_header_held = g_onepass;      /* one pass: the header follows the insert-length table (gpu_wait) */
    gpu_wait(d);                    /* the reference is on the device */
    pipe_global_init(d);
    if (o->serial) { walker_setup(&o->w[0], d); walker_adopt_driver(&o->w[0], d); }
    pthread_mutex_lock(&o->mu); o->go = 1; pthread_cond_broadcast(&o->cv); pthread_mutex_unlock(&o->mu);
    /* the main thread's own pipeline: the stage of every group */
    driver sd = *d;
    ppipe S;
    pipe_init(&S, &sd, 0);
    /* ONE pass over the BAM (no config file): the walk runs without insert lengths (which records are candidates does not depend on
     * them; the triage leaves range_max open), collecting the extrema per read group as estimate_insertlengths would
     * (src/bamoperations.c:15-86).  The table made from the FIRST claims' extrema -- proper pairs are flagged against the aligner's
     * own insert-size bounds, so the extrema of a library show within its first few hundred thousand pairs -- serves as the table
     * while the rest of the file is still being walked: groups are staged and replayed behind the walk as in a run with a config
     * file, their output kept back.  When every piece is in, the table of the whole file is made; if it is the provisional one,
     * the output goes out; if not (or if anything went wrong on the way), only the header is out and the program takes the run
     * again with the pre-pass (spec_fallback).  Small inputs, -o detailed: the table is made when the walk is over. */
    const char* re = getenv("INDELMINER_REPLAYERS");
    int nrep = re ? atoi(re) : 8;       /* idle while there is nothing to replay; at the end of the walk the cores are theirs (three left the last contigs a backlog of 1.3 s at WGS scale) */
    if (o->serial || strcmp(O.outputformat, "vcf") != 0 || nrep < 2) nrep = 0;
    if (nrep > 8) nrep = 8;
    spec_t spec;
    memset(&spec, 0, sizeof spec);
    int speculate = 0, spec_first = 0;
    int32_t mg_abort_tid = INT_MAX;         /* several ranks: the first contig some rank's WALK met a record the reference dies on -- all ranks know */
    if (g_mg && !o->serial) {
        /* ONE walk per rank.  Every rank walks the pieces the plan gives it -- without insert lengths when there is no config file
         * (which records are candidates does not depend on them), collecting what the estimate needs as it goes -- and the ranks
         * then exchange what their walks logged, in ONE all-gather: from it every rank makes the same insert-length table, read-counter
         * prefixes and marker floors a single process would have (mg_exchange_logs).  The BAM is inflated once. */
        for (int i = 0; i < o->nw; i++) pthread_join(o->w[i].th, NULL);
        phase_time("the walk of this rank's pieces (inflate + count; triage on the device)");
        const int estimate = O.configfile == NULL;
        mgbuf mine = { NULL, 0, 0 };
        mg_log_from_groups(g_mg, d, o, estimate, &mine);
        mg_exchange_logs(g_mg, d, estimate, mine, o->pieces, o->n_pieces, g_mg->piece_walker);
        phase_time("the ranks' walk logs exchanged (one all-gather) and merged");
        if (estimate && g_mg->abort_piece < o->n_pieces && !getenv("INDELMINER_NO_HANDOFF")) {
            /* a walk was not survived and the insert lengths were to come from the walks: the table would miss that piece's pairs.
             * Nothing is out yet, not even the header -- one record-at-a-time process takes the whole run, as in a single-process
             * one-pass run (the reference would have finished its estimation pass and died in its walk) */
            if (g_mg->rank != 0) { im_comm_destroy(g_mg->comm); fflush(stderr); _exit(EXIT_SUCCESS); }
            mg_discard_dir(g_mg);
            mg_restore_stdout(g_mg);
            handoff_to_host_child();
        }
        mg_after_logs(g_mg, d);
        if (g_mg->abort_piece < o->n_pieces) mg_abort_tid = o->pieces[g_mg->abort_piece].tid;
        if (g_header_held) { g_header_held = 0; header_out(); }
        if (estimate) {
            /* fetch_func looks the read group of every counted read up: a group the table does not know ends the reference at that
             * read -- the claim counts as one whose walk was not survived (the record-at-a-time child finds the read) */
            for (int ci = 0; ci < o->n_claims; ci++) {
                pgroup* G = o->claims[ci].G;
                for (int k = 0; G && k < G->n_crg; k++)
                    if (!qhash_lookup(d->insertlengths, G->crg[k], (int)strlen(G->crg[k]))) { o->claims[ci].aborted = 1; o->claims[ci].G = NULL; break; }
            }
            pipe_global_init(d);
            /* every walked group's candidates get their range[1], the groups spread over threads */
            int nt = o->nw > 1 ? o->nw : 1;
            if (nt > o->n_claims) nt = o->n_claims ? o->n_claims : 1;
            apply_job* aj = xcalloc((size_t)nt, sizeof(apply_job));
            for (int i = 0; i < nt; i++) {
                aj[i].o = o; aj[i].first = i; aj[i].step = nt; aj[i].rd = *d;
                aj[i].rd.rg_last_val = NULL; aj[i].rd.rg_last_name[0] = 0; aj[i].rd.gpu_pending = 0;
                if (pthread_create(&aj[i].th, NULL, apply_thread, &aj[i]) != 0) fatalf("cannot start a thread");
            }
            for (int i = 0; i < nt; i++) pthread_join(aj[i].th, NULL);
            free(aj);
            phase_time("insert lengths applied: candidates' ranges");
        }
        /* the flush points of every walked group (the read counter in front of each piece is known now), and the groups walked
         * for other ranks' contigs made ready for the trip */
        for (int ci = 0; ci < o->n_claims; ci++) {
            if (g_mg->claim_walker[ci] != g_mg->rank) continue;
            claim_t* c = &o->claims[ci];
            const int ship = g_mg->claim_owner[ci] != g_mg->rank || g_mg_self_ship;
            if (!c->G) { if (ship) package_pack(ci, NULL, 1); continue; }
            if (o->pieces[c->first].tid >= mg_abort_tid) continue;      /* nobody stages it */
            int64_t nr = 0;
            group_flush_points(c->G, &nr);
            c->G->from_package = 1;
            if (ship) {
                package_pack(ci, c->G, 0);      /* the host part as one block; the device arrays stay parked until mg_ship_groups */
                group_free(c->G); free(c->G);
                c->G = NULL;
            }
        }
    }
    if (g_onepass && !g_mg) {
        const char* sp = getenv("INDELMINER_SPECULATE");
        /* Worth it, and likely to hold, on large inputs only: the largest proper-pair insert size is the aligner's cut-off, which a
         * library of hundreds of millions of pairs meets within its first per cent, a small one perhaps never in its first
         * claims -- and a run taken twice costs more than the tail it was to hide.  INDELMINER_SPECULATE=1 / =0 override. */
        int64_t bam_bytes = 0;
        for (int32_t t = 0; t < d->hdr->n_targets; t++) bam_bytes += bai_contig_bytes(d->idx, t);
        const int wanted = sp ? strcmp(sp, "0") != 0 : bam_bytes >= ((int64_t)4 << 30);
        speculate = t_out != NULL && nrep > 0 && wanted && !getenv("INDELMINER_NO_HANDOFF");
        const int first_k = speculate ? (o->n_claims < o->nw ? o->n_claims : o->nw) : o->n_claims;
        pthread_mutex_lock(&o->mu);
        for (int ci = 0; ci < first_k; ci++) while (!o->claims[ci].walked) pthread_cond_wait(&o->cv, &o->mu);
        pthread_mutex_unlock(&o->mu);
        if (!speculate || first_k == o->n_claims) {
            speculate = 0;
            for (int i = 0; i < o->nw; i++) pthread_join(o->w[i].th, NULL);
            phase_time("the walk of all pieces (inflate + count + insert-length extrema; triage on the device)");
        }
        int aborted = 0;
        for (int ci = 0; ci < first_k; ci++) aborted |= o->claims[ci].aborted;
        if (aborted && g_handoff_pool) pipeline_handoff();      /* nothing is out yet: the record-at-a-time run prints it all */
        if (!aborted) {
            for (int ci = 0; ci < first_k; ci++) spec_collect(&spec, d, o->claims[ci].G);
            spec.n_prov = onepass_table(d, &spec, speculate ? 0 : 1);
            pipe_global_init(d);
            if (speculate) {
                spec_first = first_k;
                pthread_mutex_lock(&o->mu); g_spec_active = 1; pthread_cond_broadcast(&o->cv); pthread_mutex_unlock(&o->mu);
                phase_time("provisional insert lengths from the first claims");
            } else {
                onepass_counted_groups_known(d, o, o->n_claims);      /* a counted read of a group the table does not know: the reference dies at it */
                g_header_held = 0;
                print_output_header();
                /* every group's candidates get their range[1], the groups spread over threads */
                int nt = o->nw > 1 ? o->nw : 1;
                if (nt > o->n_claims) nt = o->n_claims ? o->n_claims : 1;
                apply_job* aj = xcalloc((size_t)nt, sizeof(apply_job));
                for (int i = 0; i < nt; i++) {
                    aj[i].o = o; aj[i].first = i; aj[i].step = nt; aj[i].rd = *d;
                    aj[i].rd.rg_last_val = NULL; aj[i].rd.rg_last_name[0] = 0; aj[i].rd.gpu_pending = 0;
                    if (pthread_create(&aj[i].th, NULL, apply_thread, &aj[i]) != 0) fatalf("cannot start a thread");
                }
                for (int i = 0; i < nt; i++) pthread_join(aj[i].th, NULL);
                free(aj);
                phase_time("insert lengths applied: candidates' ranges");
            }
        }
    }
    /* Replay workers: the replay of a group (evidence objects, paired-read components, merge, print) is the longest serial
     * piece of a run once the walks overlap; groups are independent of each other, so several are replayed at once, each
     * into a buffer that is written out when every group before it has been.  The numbered blocks of -o detailed, annotate
     * mode (one known-variant list) keep the replay on this thread; a multi-GPU run's jobs go to their contigs' part files (job_emit). */
    replayer_t* rp = nrep ? xcalloc((size_t)nrep, sizeof(replayer_t)) : NULL;
    o->jobs = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof(rjob_t));
    for (int i = 0; i < nrep; i++) {
        rp[i].pool = o; rp[i].rd = *d; rp[i].rd.gpu_pending = 0;
        if (pthread_create(&rp[i].th, NULL, replay_thread, &rp[i]) != 0) fatalf("cannot start a replay thread");
    }
    o->printed = 0;
    int64_t numread = d->numread;
    int floor_ = d->marker_floor;
    carry_t C;
    memset(&C, 0, sizeof C);
    C.frozen_min = ~0ull; C.tid = -1;
    /* groups of the contig being worked on: their replays start when the contig's depth array is complete (its last piece is in) */
    pgroup** held = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof(pgroup*));
    int n_held = 0;
    pgroup* chain = NULL;                   /* the same groups, for freeing them together */
    int n_freeable = 0;
    struct { pgroup* chain; int last_job; } *dead = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof *dead);
    /* multi-GPU with pieces of a contig walked by several ranks: no rank's depth array is complete before all ranks have walked
     * all their pieces -- the contigs' replays wait for the sum (im_depth_allreduce) */
    struct { pgroup** held; int n_held; pgroup* chain; } *late = (g_mg && g_mg->split) ? xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof *late) : NULL;
    volatile int n_late = 0;                /* read behind a longjmp (g_mg_split_abort) */
    pgroup** arrived = NULL;                /* multi-GPU, pieces over several ranks: the groups other ranks walked for this rank's contigs */
    if (late) {
        /* ONE exchange: what each claim's walk left (all-gather), and the walked groups to their contigs' owners device to device
         * (one RCCL send / receive group).  Staging follows, in file order. */
        uint8_t* wab = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), 1);
        for (int ci = 0; ci < o->n_claims; ci++) wab[ci] = (uint8_t)(o->claims[ci].aborted != 0);
        arrived = xcalloc((size_t)(o->n_claims ? o->n_claims : 1), sizeof(pgroup*));
        const int fa = mg_ship_groups(g_mg, d, o->n_claims, wab, arrived);
        free(wab);
        if (fa < o->n_claims && o->pieces[o->claims[fa].first].tid < mg_abort_tid) mg_abort_tid = o->pieces[o->claims[fa].first].tid;
        phase_time("walked groups exchanged between the ranks (RCCL send / receive)");
        /* a record the reference dies on, met while this rank stages: the rank stops staging there but still joins the sum of
         * the depth arrays below -- the other ranks are on their way into it (pipeline_handoff jumps here) */
        g_mg_split_armed = 1;
        if (setjmp(g_mg_split_abort)) goto claims_done;
    }
    for (int ci = 0; ci < o->n_claims; ci++) {
        claim_t* c = &o->claims[ci];
        if (g_mg && g_mg->claim_owner[ci] != g_mg->rank) continue;        /* another rank's contig */
        g_mg_cur_tid = o->pieces[c->first].tid;
        if (g_mg && g_mg_cur_tid >= mg_abort_tid) {      /* every rank stops in front of that contig */
            if (late) { g_mg->abort_tid = mg_abort_tid; break; }
            pipeline_handoff();                          /* contigs walked by their owners: the rank reports and is done (no collective follows) */
        }
        if (g_mg && (g_mg->claim_walker[ci] != g_mg->rank || (g_mg_self_ship && !o->serial))) {
            c->G = arrived[ci];
            c->walked = 1; c->aborted = c->G == NULL;
            if (c->aborted) pipeline_handoff();
        } else if (o->serial) {
            /* walked here, after the replay of the previous contig let go of the known-variant list */
            const int32_t tid = o->pieces[c->first].tid;
            known_free(&g_known);
            read_variants(g_vcfname, tid, d->hdr->target_name[tid], &g_known);
            if (g_known.n == 0) { pthread_mutex_lock(&o->mu); o->staged = ci + 1; pthread_mutex_unlock(&o->mu); continue; }   /* src/indelminer.c:788 */
            g_main_in_walk = g_handoff_pool != NULL;
            c->G = walk_claim(&o->w[0], o, c);
            g_main_in_walk = 0;
            c->walked = 1;
        } else {
            pthread_mutex_lock(&o->mu);
            while (!c->walked) pthread_cond_wait(&o->cv, &o->mu);
            pthread_mutex_unlock(&o->mu);
            if (c->aborted) pipeline_handoff();
        }
        phase_time("waited for the walk (inflate + count; triage on the device)");
        pgroup* G = c->G;
        G->seq = ci;
        if (speculate) {
            if (ci >= spec_first) spec_collect(&spec, d, G);                 /* the first claims' went into the provisional table */
            for (int k = 0; k < G->n_crg; k++) if (!qhash_lookup(d->insertlengths, G->crg[k], (int)strlen(G->crg[k]))) spec.unknown_counted_rg = 1;
            if (!G->sv_range) G->sv_range = group_ranges(d, G);               /* walked before that table was there */
        }
        const int first_of_contig = G->ctg[0].first, last_of_contig = G->ctg[G->n_ctg - 1].last;
        const int floor_of_contig = (g_mg && first_of_contig) ? g_mg->floor[G->ctg[0].tid] : floor_;
        static int contig_floor;            /* the floor all pieces of the contig in hand are measured against */
        if (first_of_contig) contig_floor = floor_of_contig;
        stage_take_front(G, &C);
        g_main_in_walk = g_handoff_pool != NULL;        /* a record the reference dies on inside the pair table: hand the run over */
        group_pair_table(d, G);
        g_main_in_walk = 0;
        if (!g_mg && group_meets_earlier_contigs(G)) {
            if (g_handoff_pool) pipeline_handoff();
            fatalf("read names are shared between contigs (the reference pairs them across contigs in its one pair table): run with INDELMINER_PIPELINE=host");
        }
        if (!G->from_package) group_flush_points(G, &numread);
        group_resolve_flushes(G, &floor_);
        stage_run_group(&S, G);
        stage_leftovers(G, &C, contig_floor);
        pthread_mutex_lock(&o->mu); o->staged = ci + 1; pthread_cond_broadcast(&o->cv); pthread_mutex_unlock(&o->mu);
        G->next_of_contig = chain; chain = G;
        held[n_held++] = G;
        if (!last_of_contig) continue;
        if (late) {
            late[n_late].held = xmalloc(sizeof(pgroup*) * (size_t)n_held);
            memcpy(late[n_late].held, held, sizeof(pgroup*) * (size_t)n_held);
            late[n_late].n_held = n_held; late[n_late].chain = chain; n_late++;
            n_held = 0; chain = NULL;
            continue;
        }
        /* the contig (or the run of small contigs) is complete: its depth array, then its groups' replays */
        for (int k = 0; k < n_held; k++)
            for (int cj = 0; cj < held[k]->n_ctg; cj++)
                if (held[k]->ctg[cj].last && g_region_tid < 0) GPU2(d, im_depth_scan(d->gpu, held[k]->ctg[cj].tid, S.stream));
        GPU2(d, im_stream_sync(d->gpu, S.stream));
        if (nrep) {
            pthread_mutex_lock(&o->mu);
            for (int k = 0; k < n_held; k++) {
                rjob_t* J = &o->jobs[o->n_jobs];
                J->G = held[k]; J->buf = NULL; J->len = 0; J->done = 0; J->last_of_contig = k == n_held - 1;
                o->n_jobs++;
            }
            dead[n_freeable].chain = chain; dead[n_freeable].last_job = o->n_jobs - 1; n_freeable++;
            pthread_cond_broadcast(&o->cv);
            /* whatever is complete at the head of the queue goes out now */
            while (o->printed < o->n_jobs && o->jobs[o->printed].done) {
                rjob_t* P = &o->jobs[o->printed++];
                pthread_mutex_unlock(&o->mu);
                if (P->len && speculate) spec_keep_output(&spec, P->buf, P->len);
                else job_emit(P);
                free(P->buf);
                pthread_mutex_lock(&o->mu);
            }
            /* contigs whose every replay is done: their groups go */
            for (int k = 0; k < n_freeable; k++) {
                if (!dead[k].chain) continue;
                int all = 1;
                for (int j = k ? dead[k - 1].last_job + 1 : 0; j <= dead[k].last_job; j++) all &= o->jobs[j].done;
                if (all) { pgroup* ch = dead[k].chain; dead[k].chain = NULL; pthread_mutex_unlock(&o->mu); groups_free_chain(ch); pthread_mutex_lock(&o->mu); }
            }
            pthread_mutex_unlock(&o->mu);
        } else {
            for (int k = 0; k < n_held; k++) group_replay(d, held[k]);
            phase_time("replay (variants, merge, print)");
            groups_free_chain(chain);
        }
        n_held = 0; chain = NULL;
    }
claims_done:
    g_mg_split_armed = 0;
    if (late) {
        /* every rank has walked what it walks and staged what it owns */
        free(arrived);
        mg_arm("the sum of the depth arrays");
        GPU2(d, im_depth_allreduce(d->gpu, g_mg->comm));
        mg_disarm();
        phase_time("depth arrays summed over the ranks");
        for (int k = 0; k < n_late; k++) {
            for (int j = 0; j < late[k].n_held; j++)
                for (int cj = 0; cj < late[k].held[j]->n_ctg; cj++)
                    if (late[k].held[j]->ctg[cj].last) GPU2(d, im_depth_scan(d->gpu, late[k].held[j]->ctg[cj].tid, S.stream));
            GPU2(d, im_stream_sync(d->gpu, S.stream));
            if (nrep) {
                pthread_mutex_lock(&o->mu);
                for (int j = 0; j < late[k].n_held; j++) {
                    rjob_t* J = &o->jobs[o->n_jobs];
                    J->G = late[k].held[j]; J->buf = NULL; J->len = 0; J->done = 0; J->last_of_contig = j == late[k].n_held - 1;
                    o->n_jobs++;
                }
                dead[n_freeable].chain = late[k].chain; dead[n_freeable].last_job = o->n_jobs - 1; n_freeable++;
                pthread_cond_broadcast(&o->cv);
                pthread_mutex_unlock(&o->mu);
            } else {
                for (int j = 0; j < late[k].n_held; j++) group_replay(d, late[k].held[j]);
                groups_free_chain(late[k].chain);
            }
            free(late[k].held);
        }
        free(late);
        phase_time("replay (variants, merge, print)");
    }
    if (nrep) {
        pthread_mutex_lock(&o->mu);
        o->jobs_closed = 1;
        pthread_cond_broadcast(&o->cv);
        while (o->printed < o->n_jobs) {
            while (!o->jobs[o->printed].done) pthread_cond_wait(&o->cv, &o->mu);
            rjob_t* P = &o->jobs[o->printed++];
            pthread_mutex_unlock(&o->mu);
            if (P->len && speculate) spec_keep_output(&spec, P->buf, P->len);
            else job_emit(P);
            free(P->buf);
            pthread_mutex_lock(&o->mu);
        }
        pthread_mutex_unlock(&o->mu);
        for (int i = 0; i < nrep; i++) pthread_join(rp[i].th, NULL);
        if (speculate) {
            /* every piece is in and replayed: does the table of the whole file say what the provisional one said? */
            for (int i = 0; i < o->nw; i++) pthread_join(o->w[i].th, NULL);
            if (!spec_holds(d, &spec) || spec.unknown_counted_rg) spec_fallback("the whole file has other read groups or larger insert sizes");
            g_spec_active = 0;
            onepass_print_tables(d, &spec);
            g_header_held = 0;
            print_output_header();
            if (spec.out_len && fwrite(spec.out, 1, spec.out_len, OUT) != spec.out_len) fatalf("write to stdout failed");
            free(spec.out); spec.out = NULL;
        }
        fflush(OUT);
        fflush(stdout);
        phase_time("replay workers drained");
        cpu_report();
        if (getenv("INDELMINER_TIDY_EXIT")) for (int k = 0; k < n_freeable; k++) if (dead[k].chain) groups_free_chain(dead[k].chain);
        free(rp);
    }
    free(o->jobs); o->jobs = NULL; free(held); free(dead);
    d->numread = numread;
    if (g_handoff_pool) { g_handoff_pool = NULL; if (t_out) { fflush(t_out); fclose(t_out); t_out = NULL; } }
    for (int i = 0; i < o->nw && !o->serial && !g_onepass && !g_mg; i++) pthread_join(o->w[i].th, NULL);      /* one pass, several ranks: joined above */
    /* the walkers' pinned rings and device arrays go with the process unless a tidy exit is asked for (leak checkers):
     * un-pinning and freeing them costs more than the whole device stage of a run */
    if (getenv("INDELMINER_TIDY_EXIT")) {
        for (int i = 0; i < o->nw; i++) {
            walker_t* W = &o->w[i];
            pipe_destroy(&W->P);
            bam_header_free(W->hdr);
            bgzf_close(W->r);
        }
        pipe_destroy(&S);
        pair_table_clear(d);
        free(C.live.v); free(C.frozen.v);
        free(o->w); free(o->claims); free(o->pieces);
        free(o);
    }
}

/* main thread, at the first group the reference does not survive: the groups in front go out, then the child takes over */
static void pipeline_handoff(void)
{
    walkpool_t* o = g_handoff_pool;
    if (g_mg && g_mg_split_armed) {
        /* pieces of contigs over several ranks: the other ranks are heading for the sum of the depth arrays -- this rank stops
         * staging here, joins that sum, replays the contigs it has complete and reports the contig when it is done (mg_finish) */
        g_mg->abort_tid = g_mg_cur_tid;
        longjmp(g_mg_split_abort, 1);
    }
    if (g_mg) {
        /* this rank's parts in front of the claim it is working on are complete (the replay workers' jobs first); the flag names
         * the claim's first contig and rank 0, once every rank has reported, prints what lies in front of the smallest such
         * contig and hands over */
        if (o && o->jobs) {
            pthread_mutex_lock(&o->mu);
            while (o->printed < o->n_jobs) {
                while (!o->jobs[o->printed].done) pthread_cond_wait(&o->cv, &o->mu);
                rjob_t* P = &o->jobs[o->printed++];
                pthread_mutex_unlock(&o->mu);
                job_emit(P);
                pthread_mutex_lock(&o->mu);
            }
            pthread_mutex_unlock(&o->mu);
        }
        g_mg->abort_tid = g_mg_cur_tid;
        mg_finish(g_mg, g_mg_driver);           /* rank 0 does not come back from this */
        fflush(stderr);
        _exit(EXIT_SUCCESS);
    }
    if (g_spec_active) { g_spec_active = 0; handoff_to_host_child(); }      /* a speculative one-pass run has printed nothing but the header */
    if (o->jobs) {
        pthread_mutex_lock(&o->mu);
        while (o->printed < o->n_jobs) {
            while (!o->jobs[o->printed].done) pthread_cond_wait(&o->cv, &o->mu);
            rjob_t* P = &o->jobs[o->printed++];
            pthread_mutex_unlock(&o->mu);
            if (P->len && fwrite(P->buf, 1, P->len, OUT) != P->len) fatalf("write to stdout failed");
            pthread_mutex_lock(&o->mu);
        }
        pthread_mutex_unlock(&o->mu);
    }
    handoff_to_host_child();
}
