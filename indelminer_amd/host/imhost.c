/*
 * imhost.c -- the indelminer host driver: same CLI, config file and VCF output as the
 * reference (src/indelminer.c), with the split-read realignment and the split-read
 * clustering done on the GPU through the C ABI of include/indelminer_amd.h.
 *
 * Host side = what BASELINE.json's north_star keeps on the host: BAM decode, the
 * fetch_func dispatch rules, insert-length estimation, paired-read evidence, variant
 * merge / filter / emit.  There is no CPU implementation of the two GPU seams in this
 * program: without the device it stops with the library's error.
 *
 * Two passes per contig:
 *   pass A  walk the BAM records, apply fetch_func's rules (src/indelminer.c:339-615),
 *           collect candidate reads, CIGAR-derived and paired-read evidence in arrival
 *           order, and record every READCHUNK flush point with its marker (617-623)
 *   GPU     one im_realign_batch over the contig's candidates
 *   pass B  replay: evidence enters the pending list in arrival order; at every flush
 *           point process_evidence -> sort -> merge -> print exactly as the reference
 */
#define _GNU_SOURCE            /* fopencookie */
#define _POSIX_C_SOURCE 200809L
#include "imhost.h"

#include <ctype.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <dirent.h>
#include <sys/wait.h>
#include <setjmp.h>
#include <spawn.h>

#define INDELMINER_VERSION 0.2          /* src/indelminer.c:26 */

#define OP_M 0
#define OP_I 1
#define OP_D 2
#define OP_N 3
#define OP_S 4
#define OP_H 5
#define OP_P 6
#define OP_EQ 7
#define OP_X 8
#define CIG_OP(c)  ((int)((c) & 15u))
#define CIG_LEN(c) ((int)((c) >> 4))

static im_options O;
static time_t t0;

static void out_flush_on_exit(void);
static void walker_bails_out(void);
static void mg_rank_failed(void);
static void fatalf(const char* fmt, ...)
{
    /* src/errors.c:15-27: message on stderr, exit(1) */
    va_list ap;
    walker_bails_out();         /* a walker thread of the pipeline does not come back from this (see handoff_to_host_child) */
    mg_rank_failed();           /* multi-GPU: rank 0 stops waiting for this rank's output */
    va_start(ap, fmt);
    out_flush_on_exit();
    fflush(stdout);
    fprintf(stderr, "indelminer: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    exit(EXIT_FAILURE);
}
/* Everything the driver prints as output goes through printf.  A replay worker (run_pipeline) points its own t_out at a
 * memory buffer, so that several groups can be replayed at once and still come out in contig order; every other thread
 * prints to stdout. */
static __thread FILE* t_out;
#define OUT (t_out ? t_out : stdout)
#define printf(...) fprintf(OUT, __VA_ARGS__)
static pthread_mutex_t g_query_mu = PTHREAD_MUTEX_INITIALIZER;     /* depth queries share the context's workspace and stream */

/* Runs the reference aborts.  The device pipeline finds the record the reference would die on during the walk (or in the device
 * stage), when only the groups in front of it have been printed; the reference has by then also printed the flushes of that
 * group in front of the record.  Rather than unpick a half-walked group, the pipeline hands the run over: it lets the groups in
 * front go out, counts the bytes it has printed (the main thread prints through a counting stream), and starts this program
 * again as a child in its record-at-a-time mode, told to drop that many bytes of its output.  The child prints the rest exactly
 * as the reference does and dies at the record with the reference's message and status; the parent exits with its status. */
static int64_t g_out_bytes = 0;             /* parent: bytes the main thread has written to stdout */
static int64_t g_out_skip = 0;              /* child: bytes of its output still to drop */
static int g_real_stderr = -1;              /* child: stderr is silent until it has something new to say */
static char** g_argv = NULL;
static __thread int t_is_main = 0;
static __thread jmp_buf* t_abort_jmp = NULL;    /* a walker thread's way out of a walk that met such a record */
static ssize_t out_cookie_write(void* c, const char* buf, size_t n)
{
    (void)c;
    size_t at = 0;
    if (g_out_skip > 0) { at = (size_t)g_out_skip < n ? (size_t)g_out_skip : n; g_out_skip -= (int64_t)at; }
    while (at < n) { const ssize_t w = write(STDOUT_FILENO, buf + at, n - at); if (w <= 0) return 0; at += (size_t)w; g_out_bytes += w; }
    return (ssize_t)n;
}
static FILE* out_cookie_open(void)
{
    cookie_io_functions_t io = { NULL, out_cookie_write, NULL, NULL };
    FILE* f = fopencookie(NULL, "w", io);
    if (f) setvbuf(f, NULL, _IOFBF, 1 << 16);
    return f;
}
/* Whatever ends the run inside a walker's walk -- the triage's error classes at harvest, the host's own checks on discordant
 * mates and read groups as the records go by -- may not be the FIRST thing the reference dies of (errors are found chunk by
 * chunk, contigs in parallel): the walker leaves the walk, and the record-at-a-time child finds the first in record order. */
static int g_main_in_walk = 0;              /* annotate mode walks on the main thread: the same, without the jump */
static void pipeline_handoff(void);
static void walker_bails_out(void)
{
    if (t_abort_jmp) { jmp_buf* j = t_abort_jmp; t_abort_jmp = NULL; longjmp(*j, 1); }
    if (t_is_main && g_main_in_walk) { g_main_in_walk = 0; pipeline_handoff(); }
}
static void out_flush_on_exit(void)
{
    if (t_out) fflush(t_out);
    if (g_real_stderr >= 0) { fflush(stderr); dup2(g_real_stderr, STDERR_FILENO); g_real_stderr = -1; }
}
extern char** environ;
static void handoff_to_host_child(void)
{
    if (t_out) fflush(t_out);
    fflush(stdout);
    char skip[32];
    snprintf(skip, sizeof skip, "%lld", (long long)g_out_bytes);
    setenv("INDELMINER_PIPELINE", "host", 1);
    setenv("INDELMINER_SKIP_STDOUT", skip, 1);
    pid_t pid;
    if (getenv("INDELMINER_DEBUG_HANDOFF")) fprintf(stderr, "[handoff] %lld bytes printed, starting the child\n", (long long)g_out_bytes);
    if (posix_spawn(&pid, "/proc/self/exe", NULL, NULL, g_argv, environ) != 0) { fprintf(stderr, "indelminer: cannot start the record-at-a-time run\n"); _exit(EXIT_FAILURE); }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) { }
    if (getenv("INDELMINER_DEBUG_HANDOFF")) fprintf(stderr, "[handoff] child status 0x%x\n", status);
    _exit(WIFEXITED(status) ? WEXITSTATUS(status) : EXIT_FAILURE);
}

#define forceassert(e) do { if (!(e)) { walker_bails_out(); out_flush_on_exit(); fprintf(stderr, "Assertion failed: %s file %s line %d\n", #e, __FILE__, __LINE__); exit(EXIT_FAILURE); } } while (0)

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int g_timing = 0;
static double g_t_last = 0;
static void phase_time(const char* what)
{
    if (!g_timing) return;
    const double t = now_ms();
    fprintf(stderr, "[timing] %-34s %9.2f ms\n", what, t - g_t_last);
    g_t_last = t;
}

static void timestamp(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, " : %ld sec. elapsed\n", (long)(time(0) - t0));
    va_end(ap);
}

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s ? s : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void* xrealloc(void* p, size_t n) { p = realloc(p, n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static char* xstrdup(const char* s) { size_t l = strlen(s); char* d = xmalloc(l + 1); memcpy(d, s, l + 1); return d; }

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
    e->qname = xstrdup(qname);
    e->aln = seglist_copy(whole);
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
static uint32_t region_depth_from_bam(driver* d, int32_t tid, int32_t start, int32_t stop);

static uint32_t region_depth(driver* d, int32_t tid, int32_t start, int32_t stop)
{
    if (stop <= start) return 0;
    if (d->depth_tid == tid) {
        /* the device holds the contig's depth array (im_depth_build in run_contig) */
        uint32_t sum = 0;
        gpu_wait(d);
        pthread_mutex_lock(&g_query_mu);
        const int qrc = d->pipe_mode ? im_depth_query_tid(d->gpu, tid, 1, &start, &stop, &sum) : im_depth_query(d->gpu, 1, &start, &stop, &sum);
        pthread_mutex_unlock(&g_query_mu);
        if (qrc != IM_OK)
            fatalf("im_depth_query: %s", im_last_error(d->gpu));
        return (uint32_t)floor(sum * 1.0 / (uint32_t)(stop - start));
    }
    /* region runs (-c): the reference pileups the whole BAM around the variant, which can reach
     * outside the analysed region -- go to the file like it does */
    return region_depth_from_bam(d, tid, start, stop);
}

static uint32_t region_depth_from_bam(driver* d, int32_t tid, int32_t start, int32_t stop)
{
    /* pileup semantics (bam_pileup.c:67-143,238-265): records with flag & (0x4|0x100|0x200|0x400)
     * or tid < 0 are skipped; a position counts a read iff its covering op is M/=/X */
    if (stop <= start) return 0;
    uint32_t* cov = xcalloc((size_t)(stop - start), sizeof(uint32_t));
    bgzf_reader* r = bgzf_open(d->bam_name);
    if (!r) fatalf("error in opening the file %s", d->bam_name);
    bam_header* h = bam_header_load(r);
    bam_region_iter it;
    bam_record b; memset(&b, 0, sizeof b);
    if (h && bam_region_begin(&it, r, d->idx, tid, start, stop) == 0) {
        while (bam_region_next(&it, &b) == 1) {
            if (b.tid < 0 || (b.flag & (0x4 | 0x100 | 0x200 | 0x400))) continue;
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
        }
    }
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
            const int qrc = d->pipe_mode ? im_depth_query_tid(d->gpu, d->depth_tid, m, beg, end, sum) : im_depth_query(d->gpu, m, beg, end, sum);
            pthread_mutex_unlock(&g_query_mu);
            if (qrc != IM_OK)
                fatalf("im_depth_query: %s", im_last_error(d->gpu));
            for (int q = 0; q < m; q++) {
                variant_t* v = out.v[who[q]];
                v->dp_cached = (int32_t)(uint32_t)floor(sum[q] * 1.0 / (uint32_t)(end[q] - beg[q]));
                v->dp_valid = 1;
            }
        }
        free(beg); free(end); free(sum); free(who);
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
        if (im_support_batch(d->gpu, nt, tg, to, qs, qo, res) != IM_OK) fatalf("im_support_batch: %s", im_last_error(d->gpu));
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
    if (g_mg_rank > 0) return;                          /* multi-GPU: rank 0 prints the header */
    if (g_mg_header_path[0] && !freopen(g_mg_header_path, "w", stdout)) fatalf("cannot write %s", g_mg_header_path);   /* ... as the first part */
    if (strncmp(O.outputformat, "vcf", 3) == 0) print_vcf_preamble();
    if (g_vcfname != NULL)
        printf("##INFO=<ID=%s,Number=0,Type=Flag,Description=\"The variant is also present in this sample\">\n", g_sample_name);
    if (strncmp(O.outputformat, "vcf", 3) == 0) printf("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n");
    fflush(stdout);
    /* the header part is complete; whatever a library prints on stdout from here on (librccl's banner) is not VCF */
    if (g_mg_header_path[0] && !freopen("/dev/stderr", "w", stdout)) { }
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
    while (bam_region_next(&it, &b) == 1) {
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
} mgpu;

static mgpu* g_mg = NULL;
static driver* g_mg_driver = NULL;
static void mg_finish(mgpu* m, driver* d);
static int g_mg_cur_tid = -1;          /* the first contig of the claim the main thread is working on */

static void mg_path(const mgpu* m, char* out, size_t cap, const char* what, int idx) { snprintf(out, cap, "%s/%s.%d", m->dir, what, idx); }

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

#define PIPE_CHUNK_BYTES   (32u << 20)
#define PIPE_CHUNK_RECS    (PIPE_CHUNK_BYTES / 64u)
#define PIPE_NCHUNK        4

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
    int32_t *cn_rec, *cn_pos; int64_t n_cn, cap_cn;
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
    void *order, *clkey, *clfirst, *clcount, *counts, *gscratch, *fdesc, *fgscratch; size_t gscratch_bytes, fgscratch_bytes;
    /* confirmed by harvested chunks / still in flight */
    int32_t conf_cand, conf_err; int64_t conf_bytes, fly_recs, fly_seq;
    im_triage_params tp;
    int ready, own_stream;
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
    /* a stream per walker: its uploads and triage launches, and the device stage of its groups.  INDELMINER_STREAMS=shared
     * puts every walker on the context's stream instead (the GPU then sees the run exactly as with one walker: a
     * debugging aid -- it is how the group-by scratch bug of profiles/README.md r02 was told apart from a device race) */
    P->own_stream = !(getenv("INDELMINER_STREAMS") && strcmp(getenv("INDELMINER_STREAMS"), "shared") == 0);
    if (P->own_stream) GPU(im_stream_create(d->gpu, &P->stream));
    else P->stream = im_ctx_stream(d->gpu);
    for (int i = 0; i < PIPE_NCHUNK && with_chunks; i++) {
        pchunk* c = &P->ck[i];
        GPU(im_host_alloc(d->gpu, PIPE_CHUNK_BYTES, (void**)&c->h_raw));
        GPU(im_host_alloc(d->gpu, 4 * ((size_t)PIPE_CHUNK_RECS + 1), (void**)&c->h_off));
        GPU(im_host_alloc(d->gpu, 64, (void**)&c->h_cnt));
        c->d_raw = pdev_alloc(P, PIPE_CHUNK_BYTES + 64);
        c->d_off = pdev_alloc(P, 4 * ((size_t)PIPE_CHUNK_RECS + 1));
        c->d_class = pdev_alloc(P, PIPE_CHUNK_RECS);
        c->scratch_bytes = im_dev_triage_scratch_bytes((int32_t)PIPE_CHUNK_RECS);
        c->d_scratch = pdev_alloc(P, c->scratch_bytes);
        GPU(im_dev_triage_scratch_init(d->gpu, (int32_t)PIPE_CHUNK_RECS, c->d_scratch, c->scratch_bytes, P->stream));
        GPU(im_event_create(d->gpu, &c->done));
    }
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
}

static void pipe_destroy(ppipe* P)
{
    if (!P->ready) return;
    for (int i = 0; i < PIPE_NCHUNK && P->ck[i].h_raw; i++) {
        pchunk* c = &P->ck[i];
        im_host_free(P->d->gpu, c->h_raw); im_host_free(P->d->gpu, c->h_off); im_host_free(P->d->gpu, c->h_cnt);
        im_dev_free(P->d->gpu, c->d_raw); im_dev_free(P->d->gpu, c->d_off); im_dev_free(P->d->gpu, c->d_class); im_dev_free(P->d->gpu, c->d_scratch);
        im_event_destroy(c->done);
    }
    pipe_free_cands(P);
    if (P->rstat) { im_dev_free(P->d->gpu, P->rstat); im_dev_free(P->d->gpu, P->rslot); im_dev_free(P->d->gpu, P->rcompact); }
    if (P->rcount) im_dev_free(P->d->gpu, P->rcount);
    im_dev_free(P->d->gpu, P->counters); im_dev_free(P->d->gpu, P->counts); im_dev_free(P->d->gpu, P->cut); im_dev_free(P->d->gpu, P->fdesc);
    if (P->own_stream) im_stream_destroy(P->d->gpu, P->stream);
    P->ready = 0;
}

static void group_free(pgroup* G)
{
    if (G->phantom && G->pe && G->n_pe_front > 0 && G->pe[G->n_pe_front - 1] && G->pe[G->n_pe_front - 1]->type == EV_PHANTOM) evidence_free(G->pe[G->n_pe_front - 1]);
    free(G->ctg); free(G->fl); free(G->fp); free(G->pe); free(G->pe_rec); free(G->cand_rec); free(G->craw_off); free(G->craw);
    free(G->cn_rec); free(G->cn_pos); free(G->lm_rec); free(G->lm_val);
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
    P->oldest = (P->oldest + 1) % PIPE_NCHUNK;
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
    P->cur = (P->cur + 1) % PIPE_NCHUNK;
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
static void host_rg_stat(pgroup* G, const bam_record* b, int64_t rec_in_contig)
{
    const int flag = b->flag;
    if (!((flag & 0x1) && !(flag & 0x4) && (flag & 0x2) && !(flag & (0x100 | 0x200 | 0x400)) &&
          b->isize >= 0 && b->mpos - b->pos >= 0 && b->isize >= b->mpos - b->pos)) return;
    const uint8_t* rg = bam_aux_find(b, "RG");
    const char* rgname = "generic";
    if (rg) { forceassert(rg[0] == 'Z'); rgname = bam_aux_str(rg); }
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
/* Reads beyond 255 bases take the realign kernels' second launch (im_expect_read_length, include/indelminer_amd.h): the context
 * hears of the longest read so far the moment a walker meets it, i.e. before the group that holds it is launched. */
static volatile int g_longest_read = 255;
static void note_long_read(driver* d, int l_seq)
{
    if (l_seq > IM_MAX_READ || O.numgaps != 0) return;                 /* the kernel reports such a candidate, the run stops with its name */
    if (im_expect_read_length(d->gpu, l_seq) != IM_OK) fatalf("im_expect_read_length: %s", im_last_error(d->gpu));
    int cur = __atomic_load_n(&g_longest_read, __ATOMIC_RELAXED);
    while (l_seq > cur && !__atomic_compare_exchange_n(&g_longest_read, &cur, l_seq, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

static void pipe_host_record(driver* d, pgroup* G, const bam_record* b)
{
    const int flag = b->flag;
    if (g_onepass) {
        host_rg_stat(G, b, ((int64_t)(b->pos < 0 ? 0 : b->pos) << 32) | (G->n_rec - 1 - G->ctg[G->cur_ctg].rec0));
        if (!G->cov.sum) cov_init(&G->cov, d->hdr->n_targets);
        cov_record(&G->cov, b);
    }
    if (flag & (0x100 | 0x200 | 0x400 | 0x800)) return;
    if ((flag & 0x1) == 0) return;
    if (b->l_seq > g_longest_read) note_long_read(d, b->l_seq);
    const int is_aligned = (flag & 0x4) == 0, is_mate_aligned = (flag & 0x8) == 0;
    if (is_aligned && is_mate_aligned && b->tid != b->mtid) return;
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
    if (G->n_cn == G->cap_cn) {
        G->cap_cn = G->cap_cn ? G->cap_cn * 2 : (1 << 20);
        G->cn_rec = xrealloc(G->cn_rec, sizeof(int32_t) * (size_t)G->cap_cn);
        G->cn_pos = xrealloc(G->cn_pos, sizeof(int32_t) * (size_t)G->cap_cn);
    }
    G->cn_rec[G->n_cn] = (int32_t)G->n_rec; G->cn_pos[G->n_cn] = b->pos; G->n_cn++;
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
            G->fp[G->n_fp].rec = G->cn_rec[cg->cn0 + k]; G->fp[G->n_fp].pos = G->cn_pos[cg->cn0 + k]; G->n_fp++;
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
            pipe_submit(P, G);
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
    pipe_submit(P, G);
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
    /* record numbers: the front's, then the own ones counted on from n_virt; the front's slots */
    {
        int32_t* t = xmalloc(4 * ((size_t)nc + 1) + 12 * nK * IM_MAX_EV + 64);
        for (int32_t q = 0; q < K; q++) t[q] = G->front_virt[q];
        for (int32_t i = 0; i < n_own; i++) t[K + i] = G->cand_rec[i] + G->n_virt;
        if (nc) GPU(im_dev_upload(g, P->cand_rec, t, 4 * (size_t)nc));
        if (K) {
            int32_t *c = t + nc + 1, *x1 = c + nK * IM_MAX_EV, *x2 = x1 + nK * IM_MAX_EV;
            for (int32_t q = 0; q < K; q++)
                for (int k = 0; k < IM_MAX_EV; k++) { c[q * IM_MAX_EV + k] = G->front[q].cls[k]; x1[q * IM_MAX_EV + k] = G->front[q].b1[k]; x2[q * IM_MAX_EV + k] = G->front[q].b2[k]; }
            GPU(im_dev_upload(g, P->cls, c, 16 * nK)); GPU(im_dev_upload(g, P->b1, x1, 16 * nK)); GPU(im_dev_upload(g, P->b2, x2, 16 * nK));
        }
        free(t);
        int32_t cnt[16] = { 0 };
        cnt[0] = nc;
        GPU(im_dev_upload(g, P->counters, cnt, 64));
    }
    if (G->n_pe > 0) {
        /* paired-read entries (class 2) behind the split-read slots: pending ones of earlier pieces first; an entry without an
         * evidence object stands for the entries that wait for the contig's end (stage_leftovers) and carries their smallest key */
        int32_t* t = xmalloc(sizeof(int32_t) * 3 * (size_t)G->n_pe);
        for (int32_t i = 0; i < G->n_pe; i++) { t[i] = 2; t[G->n_pe + i] = G->pe[i]->b1; t[2 * (size_t)G->n_pe + i] = G->pe[i]->b2; }
        GPU(im_dev_upload(g, (char*)P->cls + 4 * pe_base, t, 4 * (size_t)G->n_pe));
        GPU(im_dev_upload(g, (char*)P->b1 + 4 * pe_base, t + G->n_pe, 4 * (size_t)G->n_pe));
        GPU(im_dev_upload(g, (char*)P->b2 + 4 * pe_base, t + 2 * (size_t)G->n_pe, 4 * (size_t)G->n_pe));
        free(t);
    }
    /* The flush list of the group, in file order, and the split-read group-by.  Within a contig the markers never decrease
     * (find_marker is a minimum over pair-table entries that leave the table or enter it at the current position of a
     * coordinate-sorted walk), so which flush consumes an entry needs no history: three chip-wide launches do the whole
     * list and the group-by (im_dev_flush_groupby).  A BAM whose positions run backwards inside a contig can break that;
     * such a group takes the sequential forms: one workgroup walking the list, or one launch pair per flush when no
     * mid-contig flush consumes anything and the pending ranges grow long. */
    im_flush_desc* fd = xcalloc((size_t)(G->n_fl ? G->n_fl : 1), sizeof(im_flush_desc));
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
    if (!per_flush && G->n_fl) GPU(im_dev_upload(g, P->fdesc, fd, sizeof(im_flush_desc) * (size_t)G->n_fl));    /* synchronous: complete before the launches below */
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
    free(fd);
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
    int32_t n_evd = 0;
    int32_t* rstat = xmalloc(4 * (size_t)(n_own ? n_own : 1));
    if (n_own > 0) GPU(im_dev_download(g, &n_evd, P->rcount, 4));
    G->res = xrealloc(G->res, sizeof(im_read_result) * (size_t)(n_evd ? n_evd : 1));
    G->res_slot = xrealloc(G->res_slot, 4 * (size_t)(n_own ? n_own : 1));
    G->s_cls = xrealloc(G->s_cls, 4 * nn * IM_MAX_EV); G->s_b1 = xrealloc(G->s_b1, 4 * nn * IM_MAX_EV); G->s_b2 = xrealloc(G->s_b2, 4 * nn * IM_MAX_EV);
    G->cons_sr = xrealloc(G->cons_sr, 4 * nn * IM_MAX_EV);
    G->cons_pe = xrealloc(G->cons_pe, 4 * (size_t)(G->n_pe ? G->n_pe : 1));
    int32_t counts[2] = { 0, 0 };
    if (n_own > 0) {
        GPU(im_dev_download(g, rstat, P->rstat, 4 * nO));
        GPU(im_dev_download(g, G->res_slot, P->rslot, 4 * nO));
        if (n_evd > 0) GPU(im_dev_download(g, G->res, P->rcompact, sizeof(im_read_result) * (size_t)n_evd));
    }
    if (nc > 0) {
        GPU(im_dev_download(g, G->s_cls, P->cls, 4 * (size_t)nc * IM_MAX_EV));
        GPU(im_dev_download(g, G->s_b1, P->b1, 4 * (size_t)nc * IM_MAX_EV));
        GPU(im_dev_download(g, G->s_b2, P->b2, 4 * (size_t)nc * IM_MAX_EV));
        GPU(im_dev_download(g, G->cons_sr, P->consumed, 4 * (size_t)nc * IM_MAX_EV));
    }
    if (G->n_pe > 0) GPU(im_dev_download(g, G->cons_pe, (char*)P->consumed + 4 * pe_base, 4 * (size_t)G->n_pe));
    GPU(im_dev_download(g, counts, P->counts, 8));
    G->n_cl = counts[0]; G->n_nodes = counts[1];
    G->cl_key = xrealloc(G->cl_key, 16 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_first = xrealloc(G->cl_first, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_count = xrealloc(G->cl_count, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->cl_sorted = xrealloc(G->cl_sorted, 4 * (size_t)(G->n_cl ? G->n_cl : 1));
    G->order = xrealloc(G->order, 4 * (size_t)(G->n_nodes ? G->n_nodes : 1));
    if (G->n_cl > 0) {
        GPU(im_dev_download(g, G->cl_key, P->clkey, 16 * (size_t)G->n_cl));
        GPU(im_dev_download(g, G->cl_first, P->clfirst, 4 * (size_t)G->n_cl));
        GPU(im_dev_download(g, G->cl_count, P->clcount, 4 * (size_t)G->n_cl));
        GPU(im_dev_download(g, G->order, P->order, 4 * (size_t)G->n_nodes));
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
        if (st == IM_ST_UNSUPPORTED) fatalf("im_dev_realign: read %d: longer than IM_MAX_READ=%d", i, O.numgaps ? 255 : IM_MAX_READ);
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
        char* bases = decode_bases(&b);
        char strand = is_rc ? '-' : '+';
        uint8_t qual;
        if (!is_aligned) {
            qual = (uint8_t)mate_mapq(&b, 1);
            if (!is_mate_rc) { revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
        } else {
            qual = b.mapq;
            if (is_rc == is_mate_rc) { revcomp_inplace(bases); strand = (strand == '+') ? '-' : '+'; }
        }
        seglist whole;
        whole.ref_start = r->ref_start; whole.n = r->n_ops; whole.ops = (uint32_t*)r->ops; whole.bases = bases;
        for (int k = 0; k < r->n_ev && k < IM_MAX_EV; k++) {
            const im_evidence* ge = &r->ev[k];
            evidence_t* e = xcalloc(1, sizeof *e);
            e->type = EV_SPLIT_READ; e->cls = ge->cls; e->strand = strand; e->qual = qual;
            e->qname = xstrdup(qname);
            e->aln = seglist_copy(&whole);
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

static void group_replay(driver* d, pgroup* G)
{
    int32_t cursor = 0;
    for (int ci = 0; ci < G->n_ctg; ci++) {
        const gcontig* cg = &G->ctg[ci];
        const int32_t tid = cg->tid;
        d->depth_tid = g_region_tid < 0 ? tid : -1;     /* -c: the depth of a variant is taken from the file (it reaches outside the stretch) */
        if (g_mg) {                                     /* one VCF part per contig, concatenated by rank 0 in contig order */
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
    char path[512], tmp[520];
    snprintf(path, sizeof path, "%s/rccl_id", m->dir);
    uint8_t id[IM_COMM_ID_BYTES];
    /* what the ranks of ONE run share and no other run has: the launcher's run id, port and process (or what the caller says) */
    char token[96];
    memset(token, 0, sizeof token);
    if (getenv("INDELMINER_RUN_TOKEN")) snprintf(token, sizeof token, "%s", getenv("INDELMINER_RUN_TOKEN"));
    else snprintf(token, sizeof token, "%s:%s:%ld", getenv("TORCHELASTIC_RUN_ID") ? getenv("TORCHELASTIC_RUN_ID") : "", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", (long)getppid());
    if (m->rank == 0) {
        /* no fork() here: the GPU helper thread is inside the HIP runtime's start-up */
        if (mkdir(m->dir, 0700) != 0 && errno != EEXIST) fatalf("cannot create the rendezvous directory %s", m->dir);
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
        gpu_wait(d);                                    /* the HIP runtime is up before librccl is asked for anything */
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
    gpu_wait(d);
    mg_arm("communicator bring-up");
    if (im_comm_init(d->gpu, id, m->rank, m->world, &m->comm) != IM_OK) fatalf("im_comm_init: %s", im_comm_last_error());
    mg_disarm();
}

/* every rank contributes `bytes` bytes (a multiple of 4); all[] receives world * bytes */
static void mg_allgather(mgpu* m, driver* d, const void* mine, void* all, size_t bytes)
{
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
    while (bam_region_next(&it, &b) == 1) {
        const int flag = b.flag;
        const int32_t this_rec = rec++;
        if (cov) cov_record(cov, &b);
        if (estimate && (flag & 0x1) && !(flag & 0x4) && (flag & 0x2) && !(flag & (0x100 | 0x200 | 0x400)) &&
            b.isize >= 0 && b.mpos - b.pos >= 0 && b.isize >= b.mpos - b.pos) {
            const uint8_t* rg = bam_aux_find(&b, "RG");
            const char* rgname = "generic";
            if (rg) { forceassert(rg[0] == 'Z'); rgname = bam_aux_str(rg); }
            mg_rg* g = &rgs[mg_rg_index(rgs, pn_rg, rgname)];
            if (!g->seen) { g->seen = 1; g->min = g->max = b.isize; g->first_tid = t; g->first_rec = ((int64_t)(b.pos < 0 ? 0 : b.pos) << 32) | (uint32_t)this_rec; }
            else { if (g->min > b.isize) g->min = b.isize; if (g->max < b.isize) g->max = b.isize; }
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
    free(b.data);
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

/* exchange + merge: the insert-length table (when estimated), the counter prefix and the marker floor of every contig */
static void mg_exchange(mgpu* m, driver* d, int estimate, const piece_t* pieces, int n_pieces, const int32_t* piece_walker)
{
    const int32_t nt = d->hdr->n_targets;
    mgbuf mine = { NULL, 0, 0 };
    mg_prewalk(m, d, estimate, &mine, pieces, n_pieces, piece_walker);
    phase_time("pre-walk of this rank's contigs (count, pair-table events, insert lengths)");
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
    for (int pi = 0; pi < n_pieces; pi++) {
        const int32_t t = pieces[pi].tid;
        m->piece_prefix[pi] = run;
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
typedef struct { pgroup* G; char* buf; size_t len; int done; int last_of_contig; } rjob_t;
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
 * the candidates' record numbers and BAM records, and the parked device arrays.  One file per claim in the rendezvous directory,
 * written under another name and renamed when complete; `aborted` = the walk met a record the reference dies on. */
#define PKG_MAGIC 0x504b4733
typedef struct { int32_t magic, aborted, n_ctg, n_fp, n_npp, n_cand, sv_n; int64_t n_rec, npp_len, craw_len, sv_bytes; } pkg_head;

static void pkg_put(FILE* fp, const void* p, size_t bytes, const char* path) { if (bytes && fwrite(p, 1, bytes, fp) != bytes) fatalf("cannot write %s", path); }
static void pkg_get(FILE* fp, void* p, size_t bytes, const char* path) { if (bytes && fread(p, 1, bytes, fp) != bytes) fatalf("%s is cut short", path); }

static void package_write(const mgpu* m, int ci, ppipe* P, pgroup* G, int aborted)
{
    char path[512], tmp[520];
    mg_path(m, path, sizeof path, "pkg", ci);
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE* fp = fopen(tmp, "wb");
    if (!fp) fatalf("cannot write %s", tmp);
    pkg_head h;
    memset(&h, 0, sizeof h);
    h.magic = PKG_MAGIC; h.aborted = aborted;
    if (!aborted) {
        h.n_ctg = G->n_ctg; h.n_fp = G->n_fp; h.n_npp = G->n_npp; h.n_cand = G->n_cand; h.sv_n = G->sv_n;
        h.n_rec = G->n_rec; h.npp_len = G->npp_len; h.craw_len = G->craw_len; h.sv_bytes = G->sv_bytes;
    }
    pkg_put(fp, &h, sizeof h, tmp);
    if (!aborted) {
        pkg_put(fp, G->ctg, sizeof(gcontig) * (size_t)G->n_ctg, tmp);
        pkg_put(fp, G->fp, sizeof(gfpoint) * (size_t)G->n_fp, tmp);
        pkg_put(fp, G->npp_off, sizeof(int64_t) * ((size_t)G->n_npp + (G->n_npp ? 1 : 0)), tmp);
        pkg_put(fp, G->npp_rec, sizeof(int32_t) * (size_t)G->n_npp, tmp);
        pkg_put(fp, G->npp_raw, (size_t)G->npp_len, tmp);
        pkg_put(fp, G->cand_rec, sizeof(int32_t) * (size_t)G->n_cand, tmp);
        pkg_put(fp, G->craw_off, sizeof(int64_t) * ((size_t)G->n_cand + (G->n_cand ? 1 : 0)), tmp);
        pkg_put(fp, G->craw, (size_t)G->craw_len, tmp);
        const size_t n = (size_t)G->sv_n, ns = n * IM_MAX_EV;
        const size_t bytes[10] = { (size_t)G->sv_bytes, 8 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * ns, 4 * ns, 4 * ns, 4 * n };
        size_t most = 0;
        for (int k = 0; k < 10; k++) if (bytes[k] > most) most = bytes[k];
        uint8_t* t = xmalloc(most + 8);
        for (int k = 0; k < 10; k++) {
            if (!bytes[k]) continue;
            GPU(im_dev_download(P->d->gpu, t, G->sv[k], bytes[k]));
            pkg_put(fp, t, bytes[k], tmp);
        }
        free(t);
    }
    if (fclose(fp) != 0 || rename(tmp, path) != 0) fatalf("cannot publish %s", path);
}

/* the owner's side: waits for the file, rebuilds the group, parks its arrays on this rank's device; NULL = the walk was aborted */
static pgroup* package_read(const mgpu* m, int ci, ppipe* P)
{
    char path[512];
    mg_path(m, path, sizeof path, "pkg", ci);
    FILE* fp = NULL;
    mg_arm("a piece another rank walks");
    while (!(fp = fopen(path, "rb"))) { struct timespec ts = { 0, 2 * 1000 * 1000 }; nanosleep(&ts, NULL); }
    mg_disarm();
    pkg_head h;
    pkg_get(fp, &h, sizeof h, path);
    if (h.magic != PKG_MAGIC) fatalf("%s is not a group of this run", path);
    if (h.aborted) { fclose(fp); unlink(path); return NULL; }
    pgroup* G = xcalloc(1, sizeof(pgroup));
    G->from_package = 1;
    G->n_ctg = G->cap_ctg = h.n_ctg; G->n_fp = G->cap_fp = h.n_fp; G->n_npp = G->cap_npp = h.n_npp; G->n_cand = G->cap_cand = h.n_cand; G->sv_n = h.sv_n;
    G->n_rec = h.n_rec; G->npp_len = G->npp_cap = h.npp_len; G->craw_len = G->craw_cap = h.craw_len; G->sv_bytes = h.sv_bytes;
    G->ctg = xmalloc(sizeof(gcontig) * (size_t)(h.n_ctg ? h.n_ctg : 1)); pkg_get(fp, G->ctg, sizeof(gcontig) * (size_t)h.n_ctg, path);
    G->fp = xmalloc(sizeof(gfpoint) * (size_t)(h.n_fp ? h.n_fp : 1)); pkg_get(fp, G->fp, sizeof(gfpoint) * (size_t)h.n_fp, path);
    G->npp_off = xmalloc(sizeof(int64_t) * ((size_t)h.n_npp + 1)); pkg_get(fp, G->npp_off, sizeof(int64_t) * ((size_t)h.n_npp + (h.n_npp ? 1 : 0)), path);
    G->npp_rec = xmalloc(sizeof(int32_t) * (size_t)(h.n_npp ? h.n_npp : 1)); pkg_get(fp, G->npp_rec, sizeof(int32_t) * (size_t)h.n_npp, path);
    G->npp_raw = xmalloc((size_t)h.npp_len + 1); pkg_get(fp, G->npp_raw, (size_t)h.npp_len, path);
    G->cand_rec = xmalloc(sizeof(int32_t) * (size_t)(h.n_cand ? h.n_cand : 1)); pkg_get(fp, G->cand_rec, sizeof(int32_t) * (size_t)h.n_cand, path);
    G->craw_off = xmalloc(sizeof(int64_t) * ((size_t)h.n_cand + 1)); pkg_get(fp, G->craw_off, sizeof(int64_t) * ((size_t)h.n_cand + (h.n_cand ? 1 : 0)), path);
    G->craw = xmalloc((size_t)h.craw_len + 1); pkg_get(fp, G->craw, (size_t)h.craw_len, path);
    const size_t n = (size_t)G->sv_n, ns = n * IM_MAX_EV;
    const size_t bytes[10] = { (size_t)G->sv_bytes, 8 * n, 4 * n, 4 * n, 4 * n, 4 * n, 4 * ns, 4 * ns, 4 * ns, 4 * n };
    size_t total = 0, most = 0;
    for (int k = 0; k < 10; k++) { total += (bytes[k] + 255) & ~(size_t)255; if (bytes[k] > most) most = bytes[k]; }
    char* slab = pdev_alloc(P, total);
    uint8_t* t = xmalloc(most + 8);
    for (int k = 0; k < 10; k++) {
        G->sv[k] = slab;
        if (bytes[k]) { pkg_get(fp, t, bytes[k], path); GPU(im_dev_upload(P->d->gpu, G->sv[k], t, bytes[k])); }
        slab += (bytes[k] + 255) & ~(size_t)255;
    }
    free(t);
    fclose(fp);
    unlink(path);
    return G;
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
    pipe_submit(&W->P, G);
    pipe_drain(&W->P, G);
    group_park_device(&W->P, G);
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
    for (;;) {
        pthread_mutex_lock(&o->mu);
        /* walked groups wait for the main thread with their logs and parked arrays: stay a bounded number of claims ahead of it */
        while (!g_onepass && !g_mg && o->next_claim < o->n_claims && o->next_claim >= o->staged + 2 * o->nw + 4) pthread_cond_wait(&o->cv, &o->mu);
        while (g_mg && o->next_claim < o->n_claims && g_mg->claim_walker[o->next_claim] != g_mg->rank) o->next_claim++;      /* another rank walks it */
        const int ci = o->next_claim < o->n_claims ? o->next_claim++ : -1;
        pthread_mutex_unlock(&o->mu);
        if (ci < 0) break;
        claim_t* c = &o->claims[ci];
        const int ship = g_mg && g_mg->claim_owner[ci] != g_mg->rank;
        if (g_handoff_pool) {
            /* a record the reference dies on ends this walker: the claim is published as it is, marked */
            W->cur_claim = c;
            if (setjmp(W->abort_jmp)) {
                const int cj = (int)(W->cur_claim - o->claims);
                if (g_mg && g_mg->claim_owner[cj] != g_mg->rank) package_write(g_mg, cj, &W->P, NULL, 1);      /* its owner hands the run over */
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
        if (g_mg) { int64_t nr = 0; group_flush_points(G, &nr); G->from_package = 1; }      /* the counter in front of every piece is known (mg_exchange) */
        if (ship) {
            package_write(g_mg, ci, &W->P, G, 0);
            im_dev_free(W->P.d->gpu, G->sv[0]);
            group_free(G); free(G);
            G = NULL;
        }
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
        G->sv_range = group_ranges(&j->rd, G);
    }
    return NULL;
}

/* groups of a contig are freed together: pending evidence points back at the groups it came from */
static void groups_free_chain(pgroup* G)
{
    while (G) { pgroup* n = G->next_of_contig; group_free(G); free(G); G = n; }
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
        if (j < 0) break;
        rjob_t* J = &o->jobs[j];
        {   /* test hook: every other replay takes this much longer, so that replays finish out of order on any machine */
            const char* dl = getenv("INDELMINER_DEBUG_REPLAY_DELAY_MS");
            if (dl && (j & 1) == 0) { struct timespec ts = { atoi(dl) / 1000, (long)(atoi(dl) % 1000) * 1000000L }; nanosleep(&ts, NULL); }
        }
        t_out = open_memstream(&J->buf, &J->len);
        if (!t_out) fatalf("cannot buffer the output of a group");
        group_replay(&R->rd, J->G);
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
    /* pieces: a contig is cut where its compressed bytes cross multiples of the piece size -- about 1/(8 walkers) of the file, at
     * least 8 MB of it (a stage and a replay have fixed costs per group), so that large contigs spread over all walkers */
    int64_t piece_bytes = total_bytes / (8 * (int64_t)nw);
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
    gpu_wait(d);                    /* the reference is on the device */
    pipe_global_init(d);
    if (o->serial) { walker_setup(&o->w[0], d); walker_adopt_driver(&o->w[0], d); }
    pthread_mutex_lock(&o->mu); o->go = 1; pthread_cond_broadcast(&o->cv); pthread_mutex_unlock(&o->mu);
    /* the main thread's own pipeline: the stage of every group */
    driver sd = *d;
    ppipe S;
    pipe_init(&S, &sd, 0);
    if (g_onepass) {
        /* ONE pass over the BAM: the walk above runs without insert lengths (which records are candidates does not depend on
         * them; the triage leaves range_max open), collecting the extrema per read group as estimate_insertlengths would
         * (src/bamoperations.c:15-86).  When every piece is in, the table is made -- read groups in the order one process
         * meets them -- and the stage of every group follows. */
        pthread_mutex_lock(&o->mu);
        for (int ci = 0; ci < o->n_claims; ci++) while (!o->claims[ci].walked) pthread_cond_wait(&o->cv, &o->mu);
        pthread_mutex_unlock(&o->mu);
        for (int i = 0; i < o->nw; i++) pthread_join(o->w[i].th, NULL);
        phase_time("the walk of all pieces (inflate + count + insert-length extrema; triage on the device)");
        int aborted = 0;
        for (int ci = 0; ci < o->n_claims; ci++) aborted |= o->claims[ci].aborted;
        if (aborted && g_handoff_pool) pipeline_handoff();      /* nothing is out yet: the record-at-a-time run prints it all */
        if (!aborted) {
            mg_rg* all = xcalloc((size_t)(o->n_claims ? o->n_claims : 1) * MG_MAX_RG, sizeof(mg_rg));
            int n_all = 0;
            for (int ci = 0; ci < o->n_claims; ci++) {
                const pgroup* G = o->claims[ci].G;
                for (int k = 0; k < G->n_rgs; k++) {
                    mg_rg* m = &all[n_all++];
                    snprintf(m->name, sizeof m->name, "%s", G->rgs[k].name);
                    m->min = G->rgs[k].min; m->max = G->rgs[k].max; m->first_tid = G->rgs[k].first_tid; m->first_rec = G->rgs[k].first_rec; m->seen = 1;
                }
            }
            mg_rg* merged = xcalloc((size_t)(n_all ? n_all : 1), sizeof(mg_rg));
            const int n = merge_rgs(all, n_all, merged);
            fprintf(stderr, "\nRead-group\tMin-value\tMax-value (estimated during the walk)\n");
            for (int j = 0; j < n; j++) rg_table_enter(d, &merged[j]);
            for (int j = 0; j < g_rg_n; j++) fprintf(stderr, "%s\t%d\t%d\n", g_rg_name[j], g_rg_range[j][0], g_rg_range[j][1]);
            free(all); free(merged);
            {
                covlist** ls = xmalloc(sizeof(covlist*) * (size_t)(o->n_claims ? o->n_claims : 1));
                int nl = 0;
                for (int ci = 0; ci < o->n_claims; ci++) if (o->claims[ci].G->cov.sum) ls[nl++] = &o->claims[ci].G->cov;
                cov_means_of_lists(d->hdr->n_targets, ls, nl);
                free(ls);
                cov_print_table(d->hdr);
            }
            pipe_global_init(d);
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
    /* Replay workers: the replay of a group (evidence objects, paired-read components, merge, print) is the longest serial
     * piece of a run once the walks overlap; groups are independent of each other, so several are replayed at once, each
     * into a buffer that is written out when every group before it has been.  The numbered blocks of -o detailed, annotate
     * mode (one known-variant list) and the per-contig part files of a multi-GPU run keep the replay on this thread. */
    const char* re = getenv("INDELMINER_REPLAYERS");
    int nrep = re ? atoi(re) : (g_onepass ? 8 : 3);      /* one-pass: every replay comes after the walk, nothing else wants the cores */
    if (o->serial || g_mg || strcmp(O.outputformat, "vcf") != 0 || nrep < 2) nrep = 0;
    if (nrep > 8) nrep = 8;
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
    int n_late = 0;
    for (int ci = 0; ci < o->n_claims; ci++) {
        claim_t* c = &o->claims[ci];
        if (g_mg && g_mg->claim_owner[ci] != g_mg->rank) continue;        /* another rank's contig */
        g_mg_cur_tid = o->pieces[c->first].tid;
        if (g_mg && g_mg->claim_walker[ci] != g_mg->rank) {
            c->G = package_read(g_mg, ci, &S);
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
                if (P->len && fwrite(P->buf, 1, P->len, OUT) != P->len) fatalf("write to stdout failed");
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
    if (late) {
        /* every rank has walked what it walks (its walkers are done: the packages are out) and staged what it owns */
        for (int i = 0; i < o->nw && !o->serial; i++) pthread_join(o->w[i].th, NULL);
        mg_arm("the sum of the depth arrays");
        GPU2(d, im_depth_allreduce(d->gpu, g_mg->comm));
        mg_disarm();
        phase_time("depth arrays summed over the ranks");
        for (int k = 0; k < n_late; k++) {
            for (int j = 0; j < late[k].n_held; j++)
                for (int cj = 0; cj < late[k].held[j]->n_ctg; cj++)
                    if (late[k].held[j]->ctg[cj].last) GPU2(d, im_depth_scan(d->gpu, late[k].held[j]->ctg[cj].tid, S.stream));
            GPU2(d, im_stream_sync(d->gpu, S.stream));
            for (int j = 0; j < late[k].n_held; j++) group_replay(d, late[k].held[j]);
            groups_free_chain(late[k].chain);
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
            if (P->len && fwrite(P->buf, 1, P->len, OUT) != P->len) fatalf("write to stdout failed");
            free(P->buf);
            pthread_mutex_lock(&o->mu);
        }
        pthread_mutex_unlock(&o->mu);
        for (int i = 0; i < nrep; i++) pthread_join(rp[i].th, NULL);
        fflush(OUT);
        fflush(stdout);
        phase_time("replay workers drained");
        if (getenv("INDELMINER_TIDY_EXIT")) for (int k = 0; k < n_freeable; k++) if (dead[k].chain) groups_free_chain(dead[k].chain);
        free(rp);
    }
    free(o->jobs); o->jobs = NULL; free(held); free(dead);
    d->numread = numread;
    if (g_handoff_pool) { g_handoff_pool = NULL; if (t_out) { fflush(t_out); fclose(t_out); t_out = NULL; } }
    for (int i = 0; i < o->nw && !o->serial && !g_onepass && !(g_mg && g_mg->split); i++) pthread_join(o->w[i].th, NULL);
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
    if (g_mg) {
        /* this rank's parts in front of the claim it is working on are complete; the flag names the claim's first contig and
         * rank 0, once every rank has reported, prints what lies in front of the smallest such contig and hands over */
        g_mg->abort_tid = g_mg_cur_tid;
        mg_finish(g_mg, g_mg_driver);           /* rank 0 does not come back from this */
        fflush(stderr);
        _exit(EXIT_SUCCESS);
    }
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

/* -------------------------------------------------------------------- main -- */

static void print_help(FILE* file)
{
    /* src/indelminer.c:883-923 */
    fprintf(file, "\n");
    fprintf(file, "Program: indelminer (Call/Tag indels from a clean BAM file)\n");
    fprintf(file, "Version: %2.2f\n\n", INDELMINER_VERSION);
    fprintf(file, "Usage:\n");
    fprintf(file, "\tindelminer [options] ref.fa [indels.vcf] sample=aln.bam\n");
    fprintf(file, "where the options are\n");
    fprintf(file, "\t-h   print help and return\n");
    fprintf(file, "\n");
    fprintf(file, "\t-i, read the configuration from this file\n");
    fprintf(file, " \t-c, only analyze this chromosomal region [ALL]\n");
    fprintf(file, " \t-t, do not call indels on 3' regions of the read\n");
    fprintf(file, " \t-q, do not call indels from reads with MQ < INT [10]\n");
    fprintf(file, " \t-n, disallow indel within INT bp towards the ends [10]\n");
    fprintf(file, " \t    We ignore the 3' soft-clipping, since that is where\n");
    fprintf(file, " \t    we expect the low quality region on the reads\n");
    fprintf(file, "\t-a, in case of overlapping indels, call all of them\n");
    fprintf(file, "\t    Default is to call the indels with most support\n");
    fprintf(file, "\t-e, minimum support for an indel [2]\n");
    fprintf(file, "\t-o, output format. vcf/detailed [vcf]\n");
    fprintf(file, "\t-s, maximum size of deletion reported using split reads [1 kbp]\n");
    fprintf(file, "\t-p, maximum size of deletion reported using PE reads [1 Mbp]\n");
    fprintf(file, "\n");
    fprintf(file, "\t-k, length of the kmers to be used in alignments[6]\n");
    fprintf(file, "\t-g, number of gaps allowed in the alignments[0]\n");
    fprintf(file, "\t-f, number of differences allowed in an alignment[6]\n");
    fprintf(file, "\t-b, require at least one read with these bases on \n");
    fprintf(file, "\t    either side of the indel[30]\n");
    fprintf(file, "\n");
    fprintf(file, "Assumptions:\n");
    fprintf(file, "\tThe BAM file is coordinate sorted\n");
    fprintf(file, "\tUnless specified in a config file, insertlengths for\n");
    fprintf(file, "\treadgroups, as well as average coverage per chromosome\n");
    fprintf(file, "\tare estimated from the BAM file (which can be slow!!!),\n");
    fprintf(file, "\tas well as lead to false negatives as some of the PE\n");
    fprintf(file, "\tevidence which is accounted for in one sample,might not\n");
    fprintf(file, "\tbe accounted for in the other\n");
}

static void free_range(void* p) { free(p); }

/* The realignment kernels take reads of up to IM_MAX_READ bases, 255 with -g > 0 (include/indelminer_amd.h; the reference has
 * no such bound, src/readaln.c:242-267).  A library of longer reads is turned away here, before any work, rather than at its
 * first long candidate somewhere inside a contig; a stray long read later on still stops the run with its name. */
static void check_read_lengths(const char* bam_name)
{
    bgzf_reader* r = bgzf_open(bam_name);
    if (!r) return;
    bam_header* h = bam_header_load(r);
    if (h) {
        bam_record b; memset(&b, 0, sizeof b);
        for (int i = 0; i < 20000 && bam_read_record(r, &b) == 1; i++)
            if (b.l_seq > (O.numgaps ? 255 : IM_MAX_READ) && (b.flag & (0x100 | 0x800)) == 0)
                fatalf("%s holds reads of %d bases (%s): this build realigns reads of up to %d bases (IM_MAX_READ, "
                       "include/indelminer_amd.h)%s", bam_name, (int)b.l_seq, BAMR_QNAME(&b), O.numgaps ? 255 : IM_MAX_READ,
                       O.numgaps ? " when -g is not 0" : "");
        free(b.data);
        bam_header_free(h);
    }
    bgzf_close(r);
}

/* annotate mode: im_support_batch aligns a read against its reference span widened by the indel's size on both sides with
 * the variant applied (check_for_indel, src/variant.c:1427-1556), at most IM_MAX_SW_TARGET bytes.  A variant file with a
 * larger split-read indel is turned away before any work. */
static void check_known_variants(const char* vcfname)
{
    FILE* fp = fopen(vcfname, "r");
    if (!fp) return;            /* the reference finds that out when it reads the first contig's variants, its header already printed: so here */
    size_t cap = 2;
    char* line = xmalloc(cap);
    while (im_getline(&line, &cap, fp) != -1) {
        if (line[0] == '#') continue;
        const char* f = line;
        size_t flen[5] = {0, 0, 0, 0, 0};
        for (int c = 0; c < 5 && *f; c++) {                 /* CHROM POS ID REF ALT */
            while (*f == ' ' || *f == '\t') f++;
            const char* e = f;
            while (*e && *e != ' ' && *e != '\t' && *e != '\n') e++;
            flen[c] = (size_t)(e - f);
            f = e;
        }
        if (strstr(line, "SPLIT_READ") == NULL) continue;      /* only split-read variants are realigned (src/variant.c:1655) */
        const size_t rl = flen[3], al = flen[4];
        const size_t indel = rl > al ? rl - al : al - rl;
        if ((size_t)IM_MAX_READ + 2 * indel + al + 8 > (size_t)IM_MAX_SW_TARGET) {
            line[flen[0] + flen[1] + 2 < 80 ? flen[0] + flen[1] + 2 : 80] = 0;
            fatalf("%s: the indel of %zu bases at %s is beyond what annotate mode realigns against (windows of up to %d bytes, "
                   "IM_MAX_SW_TARGET in include/indelminer_amd.h)", vcfname, indel, line, IM_MAX_SW_TARGET);
        }
    }
    free(line);
    fclose(fp);
}

int main(int argc, char** argv)
{
    t_is_main = 1;
    g_argv = xcalloc((size_t)argc + 1, sizeof(char*));          /* as given: the parsing below cuts the sample argument in two */
    for (int i = 0; i < argc; i++) g_argv[i] = xstrdup(argv[i]);
    {
        /* the record-at-a-time child of a pipeline run the reference aborts: its first bytes are on stdout already, and so is
         * everything it has to say on stderr until something goes wrong */
        const char* sk = getenv("INDELMINER_SKIP_STDOUT");
        if (sk) {
            g_out_skip = atoll(sk);
            unsetenv("INDELMINER_SKIP_STDOUT");
            t_out = out_cookie_open();
            if (!getenv("INDELMINER_DEBUG_HANDOFF")) {
                g_real_stderr = dup(STDERR_FILENO);
                if (!freopen("/dev/null", "w", stderr)) { }
            }
        }
    }
    O.maxdelsize = 1000; O.maxpedelsize = 1000000; O.minsupport = 2; O.klength = 6; O.numgaps = 0;
    O.outputformat = "vcf"; O.qthreshold = 10; O.ethreshold = 10; O.ethreshold_vcfcheck = 10;
    O.call_all_indels = 0; O.maxdiffsallowed = 6; O.minbalance = 30;
    const char* tie_env = getenv("INDELMINER_TIE_ORDER");       /* "expected": SURVEY.md 0.2 */
    O.tie_desc = (tie_env && strcmp(tie_env, "expected") == 0) ? 1 : 0;

    int c;
    while ((c = getopt(argc, argv, "dl:hc:e:o:k:g:x:i:s:p:tn:q:af:b:")) != -1) {
        switch (c) {
        case 'd': O.debug = 1; break;
        case 'l': break;
        case 'h': print_help(stdout); return EXIT_SUCCESS;
        case 'c': O.region = optarg; break;
        case 'e': if (sscanf(optarg, "%u", &O.minsupport) != 1) fatalf("incorrect option for -e: %s\n", optarg); break;
        case 'o': O.outputformat = optarg; break;
        case 'k': if (sscanf(optarg, "%u", &O.klength) != 1) fatalf("incorrect option for -k: %s\n", optarg); break;
        case 'f': if (sscanf(optarg, "%u", &O.maxdiffsallowed) != 1) fatalf("incorrect option for -f: %s\n", optarg); break;
        case 'g': if (sscanf(optarg, "%u", &O.numgaps) != 1) fatalf("incorrect option for -g: %s\n", optarg); break;
        case 'x': break;                                            /* accepted, unused (src/indelminer.c:793-794) */
        case 'i': O.configfile = optarg; break;
        case 's': if (sscanf(optarg, "%u", &O.maxdelsize) != 1) fatalf("incorrect option for -s: %s\n", optarg); break;
        case 'p': if (sscanf(optarg, "%u", &O.maxpedelsize) != 1) fatalf("incorrect option for -p: %s\n", optarg); break;
        case 't': break;                                            /* stored, never read (src/indelminer.c:775) */
        case 'n':
            if (sscanf(optarg, "%u", &O.ethreshold) != 1) fatalf("incorrect option for -n: %s\n", optarg);
            if (O.ethreshold < O.klength) O.ethreshold = O.klength;
            O.ethreshold_vcfcheck = O.ethreshold;
            break;
        case 'q': if (sscanf(optarg, "%d", &O.qthreshold) != 1) fatalf("incorrect option for -q: %s\n", optarg); break;
        case 'a': O.call_all_indels = 1; break;
        case 'b': if (sscanf(optarg, "%u", &O.minbalance) != 1) fatalf("incorrect option for -b: %s\n", optarg); break;
        case '?': break;
        default: print_help(stderr); return EXIT_FAILURE;
        }
    }
    forceassert(O.maxdelsize > 0);
    forceassert(O.klength > 1 && O.klength < 16);
    forceassert(strcmp(O.outputformat, "vcf") == 0 || strcmp(O.outputformat, "detailed") == 0);
    if (argc == optind) { print_help(stderr); return EXIT_FAILURE; }
    forceassert(argc - optind > 1);
    t0 = time(0);
    g_timing = getenv("INDELMINER_TIMING") != NULL;
    g_t_last = now_ms();

    const char* fasta_reference = argv[optind++];
    char* ptr = argv[optind++];
    if (strchr(ptr, '=') == NULL) {                 /* a VCF: tag its indels only (src/indelminer.c:1046-1053) */
        g_vcfname = ptr;
        O.minsupport = 1;
        O.outputformat = "vcf";
        ptr = argv[optind++];
    }
    char* samplename = ptr;
    while (*ptr != '=') ptr++;
    *ptr = 0;
    const char* bam_name = ++ptr;
    g_sample_name = samplename;
    if (g_vcfname != NULL) O.ethreshold_vcfcheck = 0;   /* src/indelminer.c:1074 */

    fprintf(stderr, "Reference fasta file: %s\n", fasta_reference);
    fprintf(stderr, "Chromosomal region  : %s\n", O.region == NULL ? "ALL" : O.region);
    fprintf(stderr, "BAM file            : %s\n", bam_name);
    if (g_vcfname != NULL) fprintf(stderr, "VCF file            : %s\n", g_vcfname);

    driver d;
    memset(&d, 0, sizeof d);
    d.depth_tid = -1;
    d.bam_name = bam_name;
    bgzf_reader* r = bgzf_open(bam_name);
    if (!r) fatalf("error in opening the file %s", bam_name);
    d.hdr = bam_header_load(r);
    if (!d.hdr) fatalf("%s is not a BAM file", bam_name);
    d.idx = bai_load(bam_name);
    if (!d.idx) fatalf("BAM indexing file is not available.");
    check_read_lengths(bam_name);
    if (g_vcfname != NULL) check_known_variants(g_vcfname);
    d.insertlengths = qhash_new(4);
    d.readpairs = qhash_new(20);

    int chromid = -1, chromstart = -1, chromstop = -1;
    if (O.region) bam_parse_region_str(d.hdr, O.region, &chromid, &chromstart, &chromstop);

    /* one process per GPU (torch.distributed.run's environment): contigs are sharded over the ranks */
    mgpu mg;
    memset(&mg, 0, sizeof mg);
    {
        const char* ws = getenv("WORLD_SIZE");
        const int world = ws ? atoi(ws) : 1;
        const char* pl = getenv("INDELMINER_PIPELINE");
        if ((world > 1 || getenv("INDELMINER_FORCE_MGPU")) && chromid == -1 && !(pl && strcmp(pl, "host") == 0)) {
            mg.world = world > 0 ? world : 1;
            mg.rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
            mg.local_rank = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : mg.rank;
            forceassert(mg.rank >= 0 && mg.rank < mg.world);
            g_mg = &mg; g_mg_rank = mg.rank; g_mg_local = mg.local_rank; g_mg_parts = 1;
            /* librccl prints a banner on descriptor 1: the VCF goes through part files and the saved descriptor, and
             * descriptor 1 points at stderr for the whole run */
            fflush(stdout);
            mg.out_fd = dup(1);
            if (mg.out_fd < 0 || dup2(2, 1) < 0) fatalf("cannot redirect stdout");
            mg.abort_tid = -1;
            if (g_vcfname != NULL) {
                /* annotate mode walks only the contigs the variant file names (src/indelminer.c:788) */
                mg.skip = xcalloc((size_t)d.hdr->n_targets, 1);
                for (int32_t i = 0; i < d.hdr->n_targets; i++) {
                    known_free(&g_known);
                    read_variants(g_vcfname, i, d.hdr->target_name[i], &g_known);
                    mg.skip[i] = g_known.n == 0;
                }
            }
            mg_plan(&mg, &d);                   /* before the walkers are planned: they take this rank's contigs */
        }
    }
    d.marker_floor = INT_MAX;

    /* the GPU: one context, opened by a helper thread while this thread reads the BAM (insert lengths) and the FASTA --
     * HIP start-up is 0.15-0.3 s of nothing but waiting */
    pthread_mutex_init(&d.gpu_mu, NULL); pthread_cond_init(&d.gpu_cv, NULL);
    d.gpu_pending = 1;
    if (pthread_create(&d.gpu_thread, NULL, gpu_open_thread, &d) != 0) fatalf("cannot start the GPU helper thread");
    /* the device pipeline: its walkers set their buffers up from now on.  A region run (-c) is the same pipeline over the pieces of
     * one stretch of one contig (its first piece also takes the records that reach into it from the front, as bam_fetch does;
     * mates outside the stretch and the depth around a variant are looked up in the file, like the reference does).
     * INDELMINER_PIPELINE=host is the record-at-a-time path, kept for runs the reference aborts (handoff_to_host_child). */
    walkpool_t* pool = NULL;
    {
        const char* pl0 = getenv("INDELMINER_PIPELINE");
        if (!(pl0 && strcmp(pl0, "host") == 0)) {
            if (chromid != -1) {
                g_region_tid = chromid; g_region_beg = chromstart; g_region_end = chromstop;
                if (g_region_end > d.hdr->target_len[chromid]) g_region_end = d.hdr->target_len[chromid];
                if (g_region_end < g_region_beg) g_region_end = g_region_beg;
            }
            /* no config file: the insert lengths are estimated by the walk itself (run_pipeline) instead of by a pass of their own;
             * multi-GPU runs and annotate mode keep the pre-pass (the shard summaries / the serial walk need the table up front) */
            {
                const char* op = getenv("INDELMINER_ONEPASS");          /* INDELMINER_ONEPASS=0: the pre-pass of the reference's layout */
                g_onepass = O.configfile == NULL && !g_mg && g_vcfname == NULL && chromid == -1 && !(op && strcmp(op, "0") == 0);
            }
            pool = walkpool_start(&d);
        }
    }

    if (O.configfile) read_configuration(O.configfile, d.insertlengths, d.hdr);
    else if (g_onepass) { }
    else if (!g_mg) { if (chromid == -1 && pool && !getenv("INDELMINER_ESTIMATE_SERIAL")) estimate_insertlengths_threads(&d, pool->pieces, pool->n_pieces); else estimate_insertlengths(&d, chromid); }
    fprintf(stderr, "\nRead-group\tMin-value\tMax-value\n----------\t---------\t---------\n");
    for (uint32_t i = 0; i <= d.insertlengths->mask; i++)
        for (qbin* it = d.insertlengths->bins[i]; it; it = it->next)
            fprintf(stderr, "%s\t%d\t%d\n", it->name, ((int32_t*)it->val)[0], ((int32_t*)it->val)[1]);
    fprintf(stderr, "----------\t---------\t---------\n\n");
    if (!g_onepass && !(g_mg && O.configfile == NULL)) cov_print_table(d.hdr);      /* one-pass and multi-rank estimates: printed when every record has been seen */
    timestamp("Read insertlengths for the BAM file");
    phase_time("open BAM, index, insert lengths");

    const int nseq = fasta_load(fasta_reference, d.hdr->n_targets, &d.sequences, &d.seqlen, chromid);
    if (nseq < 0) fatalf("error in opening the file %s", fasta_reference);
    forceassert(nseq == d.hdr->n_targets);
    timestamp("Read the reference sequence");
    phase_time("read FASTA");

    /* the reference is in: the GPU helper (started before the insert-length pass) uploads it */
    pthread_mutex_lock(&d.gpu_mu);
    d.seq_ready = 1;
    pthread_cond_broadcast(&d.gpu_cv);
    pthread_mutex_unlock(&d.gpu_mu);


    const char* pl = getenv("INDELMINER_PIPELINE");
    const int use_pipeline = !(pl && strcmp(pl, "host") == 0);
    if (g_mg) {
        mg_rendezvous(&mg, &d);
        mg_exchange(&mg, &d, O.configfile == NULL, pool->pieces, pool->n_pieces, mg.piece_walker);
        if (mg.cross) {
            /* A first mate left waiting in one contig meets a record of the same name in a later one: the reference's one
             * pair table pairs them across contigs (readpairs is never reset, src/indelminer.c), so the contigs of this
             * input are not independent.  Every rank sees that in the exchanged logs; the run goes to ONE process that
             * serves one table record by record, the other ranks have nothing to add. */
            if (mg.rank != 0) { im_comm_destroy(mg.comm); fflush(stderr); _exit(EXIT_SUCCESS); }
            fprintf(stderr, "indelminer: read names are shared between contigs (pairs across contigs in the one pair table): one process takes the run\n");
            mg_discard_dir(&mg);
            mg_restore_stdout(&mg);
            handoff_to_host_child();
        }
        if (O.configfile == NULL && mg.rank == 0) {
            fprintf(stderr, "\nRead-group\tMin-value\tMax-value (estimated over all ranks' contigs)\n");
            for (int j = 0; j < g_rg_n; j++) fprintf(stderr, "%s\t%d\t%d\n", g_rg_name[j], g_rg_range[j][0], g_rg_range[j][1]);
            cov_print_table(d.hdr);
        }
    }
    if (use_pipeline) run_pipeline(&d, pool);
    if (g_mg) mg_finish(&mg, &d);
    /* the reference prints its header before it reads the first record (src/indelminer.c:745-754 in front of 756-): a run it
     * aborts on some record has the header on stdout.  Here the header waits for the GPU context (gpu_wait). */
    if (!use_pipeline) gpu_wait(&d);
    for (int32_t i = 0; i < d.hdr->n_targets && !use_pipeline; i++) {
        if (chromid != -1 && i != chromid) continue;
        if (g_vcfname != NULL) {
            known_free(&g_known);
            read_variants(g_vcfname, i, d.hdr->target_name[i], &g_known);
            if (g_known.n == 0) continue;           /* src/indelminer.c:788 */
        }
        if (chromid == -1) run_contig(&d, i, 0, d.hdr->target_len[i], r);
        else run_contig(&d, i, chromstart, chromstop, r);
    }

    gpu_wait(&d);
    if (t_out) fflush(t_out);
    if (!getenv("INDELMINER_TIDY_EXIT")) {
        /* everything is printed: the GPU context, the pinned rings and the device arrays go with the process -- tearing the
         * HIP runtime down in order costs about as long as the whole device work of a small run (leak checkers: INDELMINER_TIDY_EXIT=1) */
        fflush(stdout);
        fflush(stderr);
        _exit(EXIT_SUCCESS);
    }
    im_ctx_destroy(d.gpu);
    bgzf_close(r);
    bai_free(d.idx);
    qhash_free(d.insertlengths, free_range);
    return EXIT_SUCCESS;
}
