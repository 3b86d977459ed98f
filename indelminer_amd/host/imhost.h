/*
 * imhost.h -- data model of the indelminer host driver (C).
 *
 * The driver keeps the reference's control flow (fetch_func dispatch, READCHUNK
 * flushes, merge / filter / VCF emit) on the host and calls the gfx950 library
 * through include/indelminer_amd.h for the two data-parallel seams.  Objects
 * are flat arrays instead of the reference's linked lists; every ordering rule
 * that leaks into the output is restated explicitly.
 */
#ifndef IMHOST_H
#define IMHOST_H

#include <stdint.h>
#include <stdio.h>

#include "hostio.h"
#include "indelminer_amd.h"

#define READCHUNK 100000            /* src/indelminer.c:28 */

enum { EV_SPLIT_READ = 0, EV_PAIRED_READ = 1, EV_COMPOSITE = 2 };   /* src/evidence.h:6-11 */
enum { CLS_INSERTION = 0, CLS_DELETION = 1 };                       /* src/evidence.h:13-17 */

typedef struct {
    uint32_t klength, numgaps, maxdelsize, maxpedelsize;
    uint32_t ethreshold, ethreshold_vcfcheck;
    int      qthreshold;
    uint32_t maxdiffsallowed, minbalance, minsupport;
    int      call_all_indels;
    const char* outputformat;       /* "vcf" | "detailed" */
    const char* region;             /* -c */
    const char* configfile;         /* -i */
    int      tie_desc;              /* --tie-order expected */
    int      debug;
} im_options;

/* a read's alignment as a segment list = packed ops + start (readseg list, src/readaln.h:13-20) */
typedef struct {
    int32_t   ref_start;
    int32_t   n;
    uint32_t* ops;                  /* len<<4 | op */
    char*     bases;                /* read bases the segments slice, NUL terminated */
} seglist;

typedef struct {
    int      type;                  /* EV_SPLIT_READ | EV_PAIRED_READ */
    int      cls;                   /* CLS_* */
    char     strand;
    uint8_t  qual;
    char*    qname;
    int32_t  b1, b2, mindelsize, max;
    seglist  aln;                   /* SR: the whole read; PE: the left read (aln1) */
    int32_t  seg;                   /* SR: index of the indel segment in aln */
    seglist  aln3;                  /* PE: the right read */
    int32_t  lflank, rflank, nd_print, nd_filter;   /* src/variant.c:217-290,704-775 */
    int      used;
    int64_t  arrival;               /* position in arrival order */
    int32_t  live_slot;             /* paired-read evidence waiting for its second mate: index in the driver's live list */
    int64_t  when;                  /* paired-read evidence: where in the contig it was completed (piece << 32 | record), see carry_item */
} evidence_t;

typedef struct {
    int       type;                 /* CLS_* */
    int       evdnctype;            /* EV_* */
    int32_t   tid;
    uint32_t  start, stop, lw, rw;
    uint32_t  support;
    evidence_t** evidence;
    /* ordering key among variants of one process_evidence call (see cluster_evidence) */
    int64_t   rep_b1, rep_b2, rep_arrival;
    /* DP= of the variant when print_variants asked the device for a whole flush at once */
    int32_t   dp_cached;
    int       dp_valid;
} variant_t;

typedef struct {
    variant_t** v;
    int n, cap;
} variant_list;

/* string-keyed table with the reference's hashtable semantics (src/hashtable.c): chains are
 * prepend order, lookup is a strncmp PREFIX match that returns the LAST hit of the chain,
 * hash = DJB2 over the bytes back to front (src/hashfunc.c:23-30) */
typedef struct qbin { struct qbin* next; char* name; void* val; } qbin;
typedef struct { int po2; uint32_t mask; qbin** bins; } qhash;
qhash* qhash_new(int po2size);
void   qhash_add(qhash* h, const char* name, int len, void* val);
qbin*  qhash_lookup(qhash* h, const char* name, int len);
void*  qhash_remove(qhash* h, const char* name, int len);
void   qhash_free(qhash* h, void (*free_val)(void*));

#endif
