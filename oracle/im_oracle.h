/*
 * im_oracle.h -- CPU restatement of indelMINER's split-read hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path is the HIP
 * library behind include/indelminer_amd.h and it fails loudly without a GPU.
 *
 * Parity status: PINNED.  Every function here is checked against the real
 * reference compiled in place (oracle/_ref/libimref.so, oracle/Makefile) on
 * the reference's own test_data and on seeded synthetic inputs
 * (tests/test_oracle_vs_ref.py), and against the committed golden vectors
 * those runs produced (tests/golden/).
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef IM_ORACLE_H
#define IM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BAM CIGAR op codes used by the path (samtools bam.h + src/readaln.h:10-11) */
#define IMO_OP_M   0
#define IMO_OP_I   1
#define IMO_OP_D   2
#define IMO_OP_S   4
#define IMO_OP_EQ  7
#define IMO_OP_X   8

#define IMO_MAX_OPS 256     /* packed CIGAR words per alignment / segment list */
#define IMO_MAX_EV  32      /* indel segments per realigned read */

/* status codes returned by imo_realign */
#define IMO_NONE    0       /* reference returns NULL (no evidence)          */
#define IMO_OK      1       /* segment list valid, n_ev >= 0                 */
#define IMO_ABORT  -1       /* the reference would hit a forceassert/exit    */
#define IMO_OVERFLOW -2     /* an IMO_MAX_* bound was exceeded               */

/* the globals the reference path reads (src/alignment.c:3-9) */
typedef struct {
    uint32_t klength;       /* -k, default 6  (src/indelminer.c:933)  */
    uint32_t numgaps;       /* -g, default 0  (src/indelminer.c:934)  */
    uint32_t maxdelsize;    /* -s, default 1000 (src/indelminer.c:930)*/
    uint32_t ethreshold;    /* -n, default 10 (src/indelminer.c:940)  */
} imo_params;

/* one banded alignment = attempt_band_alignment's outputs (src/alignment.c:343-391) */
typedef struct {
    int32_t  r1, r2, q1, q2;        /* 0-based half-open, contig / read coords */
    int32_t  n_ops;
    uint32_t ops[IMO_MAX_OPS];      /* len<<4 | op */
    int32_t  low, up;               /* band handed to local_align */
    int32_t  mismatches;            /* fetch_cigar's return value */
} imo_band_aln;

/* one D/I segment of the final list = one evidence (src/evidence.c:4-34) */
typedef struct {
    int32_t cls;            /* 0 = INSERTION, 1 = DELETION (src/evidence.h:13-17) */
    int32_t b1, b2;         /* segment start / end, 0-based                */
    int32_t seg;            /* index of the segment in ops[]               */
    int32_t read_off;       /* read offset of the segment's first base     */
    int32_t lflank, rflank; /* sum of =/X/M/I lengths in aln1 / aln3 (src/variant.c:217-274) */
    int32_t nd_print;       /* X+I+D lengths in aln1+aln3 (src/variant.c:230-241) */
    int32_t nd_filter;      /* nd_print + S lengths (src/variant.c:718-765)*/
} imo_evidence;

/* what attempt_pe_alignment leaves behind for one read */
typedef struct {
    int32_t  status;                /* IMO_* */
    int32_t  ref_start;             /* contig coord of the first segment */
    int32_t  n_ops;
    uint32_t ops[IMO_MAX_OPS];      /* final segment list (update_readsegs) */
    int32_t  n_ev;
    imo_evidence ev[IMO_MAX_EV];    /* in segment order (left to right) */
    /* bookkeeping for the roofline harness */
    int32_t  n_band;                /* band searches performed (1 or 2) */
    int32_t  win_bytes[2];          /* W of each find_best_band */
    int32_t  piece_bytes[2];        /* read-piece length of each */
    imo_band_aln piece[2];          /* the two raw band alignments */
} imo_result;

/* K1: find_best_band (src/alignment.c:393-447).  anchor is a contig coordinate
 * (may be < zstart1).  Returns 0, or IMO_ABORT when the reference asserts. */
int imo_find_best_band(const imo_params* P,
                       const char* ref, uint32_t zstart1, uint32_t end1, uint32_t anchor,
                       const char* read, uint32_t zstart2, uint32_t end2,
                       int* plow, int* pup, int* pindex, int* pcount);

/* K2+K3: attempt_band_alignment = local_align + ALIGN + fetch_cigar
 * (src/alignment.c:343-391, src/localalign.c:15-196, src/globalalign.c:333-401,507-604) */
int imo_band_alignment(const imo_params* P,
                       const char* ref, uint32_t zstart1, uint32_t end1,
                       const char* read, uint32_t zstart2, uint32_t end2,
                       int low, int up, imo_band_aln* out);

/* a2+a8..a11: attempt_pe_alignment (src/alignment.c:764-799) for one read.
 * contig/contig_len: sequences[tid] and its strlen; anchor: mate position;
 * range_max: range[1] of the read group; read: ASCII bases, already
 * reverse-complemented as the caller decided (src/indelminer.c:404-409,479-484). */
int imo_realign(const imo_params* P,
                const char* contig, int32_t contig_len,
                int32_t anchor, int32_t range_max,
                const char* read, int32_t readlen,
                imo_result* out);

/* batch form used by bench.py's cpu_baseline leg and by the parity tests.
 * bases: concatenated reads, off[n+1] offsets into it.  tid indexes contigs[]. */
int imo_realign_batch(const imo_params* P,
                      int32_t n_contigs, const char* const* contigs, const int32_t* contig_len,
                      int32_t n, const uint8_t* bases, const int64_t* off,
                      const int32_t* tid, const int32_t* anchor, const int32_t* range_max,
                      imo_result* out);

/* K5: the split-read part of process_evidence (src/indelminer.c:117-209 with
 * src/graph.c:122-127): sort by (b1,b2), take the prefix with b2 < marker,
 * group identical (cls,b1,b2).  Inputs are parallel arrays in ARRIVAL order
 * (oldest first).  Outputs: order[] = evidence indices grouped cluster by
 * cluster, clusters in ascending (b1,b2), members in ascending arrival order
 * (tie_desc=0, glibc stable qsort) or descending (tie_desc=1, the order
 * test_data/indelminer.expected.vcf was produced with);
 * cl_first[c]/cl_count[c] index into order[].  used[i] is set for every
 * evidence that became a node.  Returns the number of clusters. */
int32_t imo_cluster_sr(int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                       int32_t marker, int32_t tie_desc,
                       int32_t* order, int32_t* cl_first, int32_t* cl_count,
                       uint8_t* used);

/* K7: the alignment inside realign_with_indel (src/variant.c:1272-1424): full affine local SW of
 * query (len2) against the mutated reference window target (len1), match +2, mismatch -1, gap
 * open 4, extend 1; traceback from the first strictly-greatest cell while the score stays > 0;
 * counts substitutions, inserted+deleted bases and "aligned" bases exactly as the reference's
 * final loop does (it also counts the terminating NUL position, hence aligned >= 1). */
void imo_sw_indel(const char* target, int32_t len1, const char* query, int32_t len2,
                  int32_t* subs, int32_t* indels, int32_t* aligned);

/* a1 / fetch_func (src/indelminer.c:339-521) for one BAM record (the 32-byte core + variable part, no
 * block_size word): im_oracle_triage.c.  cls uses the IM_REC_* numbering of include/indelminer_amd.h. */
typedef struct {
    int32_t cls;
    int32_t revcomp;
    int32_t range_max;          /* range[1] of the record's read group */
    int32_t qual;               /* evidence qual: mate MQ (unmapped read) or own MAPQ (proper pair) */
    int32_t strand;             /* '+' / '-' after the flip */
    int32_t tid, anchor, l_seq; /* arguments of attempt_pe_alignment */
    int32_t n_ev;               /* check_variants (285-337), segment order */
    int32_t ev_cls[IMO_MAX_EV], ev_b1[IMO_MAX_EV], ev_b2[IMO_MAX_EV];
    int32_t want;               /* 2 / 3 once the record is a candidate -- also when cls then became an error found while the candidate
                                 * is written out (19, 20, 21); 0 when the record never got that far (new_readaln's errors included) */
} imo_triage;

void imo_triage_record(const uint8_t* rec, uint32_t len,
                       int32_t n_rg, const char* const* rg_names, const int32_t* rg_range_max,
                       int32_t qthreshold, uint32_t ethreshold_vcfcheck, uint32_t maxpedelsize,
                       imo_triage* out, char* bases_out);
void imo_depth_add(const uint8_t* rec, uint32_t len, int32_t tid, int32_t* depth, int64_t clen);
int32_t imo_flush_cut(int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                      int32_t marker, int32_t flush_id);
int32_t imo_flush_nohistory(int32_t n_fl, const int32_t* marker, const int32_t* id, const int32_t* last,
                            int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2, const int32_t* arr, int32_t* consumed);

#ifdef __cplusplus
}
#endif
#endif
