/*
 * ref_batch.c -- TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * A batch loop around the REAL reference's public attempt_pe_alignment
 * (src/alignment.h:21-25), linked against oracle/_ref/libimref.so (the
 * reference compiled in place by oracle/Makefile).  Used by bench.py's
 * cpu_baseline leg ("kind": "reference") so that the CPU number next to the
 * GPU one is the reference's own code, allocator churn and all, without
 * Python call overhead in the timed loop.
 *
 * Nothing of the reference is copied: the three struct layouts below restate
 * src/readaln.h:13-32 and src/evidence.h:20-36 so that the objects this file
 * builds are the ones the reference's functions expect.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct rb_readseg {             /* src/readaln.h:13-20 */
    struct rb_readseg* next;
    char* sequence;
    uint32_t oplen : 28, op : 4;
    int32_t start, end;
} rb_readseg;

typedef struct rb_readaln {             /* src/readaln.h:24-32 */
    char* qname;
    int32_t tid;
    char strand, index;
    uint8_t qual;
    rb_readseg* segments;
} rb_readaln;

typedef struct rb_evidence {            /* src/evidence.h:20-36 */
    struct rb_evidence* next;
    int type, variantclass;
    char strand;
    uint8_t qual;
    char* qname;
    rb_readseg *aln1, *aln2, *aln3;
    int32_t b1, b2, mindelsize, max;
    int isused;
} rb_evidence;

/* the reference's own symbols */
extern rb_evidence* attempt_pe_alignment(char** sequences, int32_t tid, int32_t position,
                                         const int* range, rb_readaln* rln);
extern rb_evidence* free_used_evidence(rb_evidence* all);   /* src/evidence.c:73-93 */
extern void free_readsegs(rb_readseg** prs);                /* src/readaln.c:316-327 */
extern unsigned klength, numgaps, maxdelsize, ethreshold;
extern uint32_t seed_mask;
extern int debug_flag;

void rb_set_params(unsigned k, unsigned g, unsigned maxdel, unsigned eth)
{
    klength = k; numgaps = g; maxdelsize = maxdel; ethreshold = eth;
    seed_mask = (1u << (2 * (k - 1))) - 1u;                 /* src/indelminer.c:1071 */
    debug_flag = 0;
}

/*
 * Runs attempt_pe_alignment for reads [0,n).  contigs[] are NUL-terminated.
 * Writes, per read, the number of evidence records and the first record's
 * (class,b1,b2) into out4[4*i..] (class = -1 when NULL was returned).
 * Returns wall seconds spent in the loop.
 */
double rb_run(char** contigs, int32_t n, const uint8_t* bases, const int64_t* off,
              const int32_t* tid, const int32_t* anchor, const int32_t* range_max, int32_t* out4)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int32_t i = 0; i < n; i++) {
        const size_t len = (size_t)(off[i + 1] - off[i]);
        /* new_unaligned_readaln's object (src/readaln.c:242-267) */
        rb_readseg* seg = calloc(1, sizeof *seg);
        seg->sequence = calloc(len + 1, 1);
        memcpy(seg->sequence, bases + off[i], len);
        seg->oplen = (uint32_t)len; seg->op = 4; seg->start = -1; seg->end = -1;
        rb_readaln rln;
        memset(&rln, 0, sizeof rln);
        rln.qname = calloc(2, 1); rln.qname[0] = 'q';
        rln.tid = -1; rln.strand = '+'; rln.index = '1'; rln.qual = 60; rln.segments = seg;
        int range[2] = { 0, range_max[i] };
        rb_evidence* ev = attempt_pe_alignment(contigs, tid[i], anchor[i], range, &rln);
        int cnt = 0;
        out4[4 * i + 1] = -1; out4[4 * i + 2] = 0; out4[4 * i + 3] = 0;
        if (ev) {
            /* list is in reverse segment order (src/alignment.c:465,471): report the leftmost */
            for (rb_evidence* e = ev; e; e = e->next) {
                cnt++;
                out4[4 * i + 1] = e->variantclass; out4[4 * i + 2] = e->b1; out4[4 * i + 3] = e->b2;
                e->isused = 1;
            }
            free_used_evidence(ev);
        } else {
            free_readsegs(&rln.segments);                    /* caller frees on NULL (src/indelminer.c:420,511) */
        }
        out4[4 * i] = cnt;
        free(rln.qname);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
