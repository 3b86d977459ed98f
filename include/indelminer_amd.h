/*
 * indelminer_amd.h -- C ABI of the MI355X (gfx950) split-read hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point names the reference interface it replaces (file:line under
 * ratan-lab/indelMINER).  The library is libindelminer_amd.so, built from
 * indelminer_amd/csrc/ with hipcc --offload-arch=gfx950.  There is NO CPU
 * fallback: every compute entry point needs a GPU and returns IM_E_NOGPU /
 * IM_E_HIP loudly without one.
 *
 * Two levels:
 *   im_realign_batch / im_cluster_sr      host buffers in, host buffers out
 *                                         (what the C host driver calls);
 *   im_dev_*                              same kernels on caller-owned device
 *                                         buffers and a caller-owned stream
 *                                         (bench.py, multi-GPU plumbing).
 */
#ifndef INDELMINER_AMD_H
#define INDELMINER_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IM_ABI_VERSION 2

/* ---- return codes ------------------------------------------------------ */
#define IM_OK             0
#define IM_E_ARG         -1     /* bad argument                                      */
#define IM_E_NOGPU       -2     /* no HIP device / wrong architecture                */
#define IM_E_HIP         -3     /* a HIP runtime call failed (see im_last_error)     */
#define IM_E_UNSUPPORTED -4     /* parameters outside what the kernels implement     */
#define IM_E_ABORT       -5     /* a read hit a condition on which the reference
                                   exits (forceassert, src/asserts.h:9-19)           */
#define IM_E_OVERFLOW    -6     /* a fixed bound (IM_MAX_OPS / IM_MAX_EV) exceeded   */

/* ---- per-read status (im_read_result.status) --------------------------- */
#define IM_ST_NONE        0     /* attempt_pe_alignment returned NULL                */
#define IM_ST_EVIDENCE    1     /* segment list + n_ev >= 1 evidence records valid   */
#define IM_ST_ABORT      -1     /* the reference would have exited on this read      */
#define IM_ST_OVERFLOW   -2
#define IM_ST_UNSUPPORTED -3    /* a read beyond 255 bases when im_expect_read_length was not told of it; base_off not a multiple of 4 */

/* CIGAR op codes in packed words (len<<4|op): samtools bam.h + src/readaln.h:10-11 */
#define IM_OP_M  0
#define IM_OP_I  1
#define IM_OP_D  2
#define IM_OP_S  4
#define IM_OP_EQ 7
#define IM_OP_X  8

#define IM_MAX_READ  1020       /* longest read of the LAID-OUT realign kernels (numgaps == 0; 255 with numgaps > 0); longer
                                   reads -- the reference takes any, src/readaln.c:242-267 -- run in a general pass behind them,
                                   see im_expect_read_length */
#define IM_MAX_SW_TARGET 4095   /* longest annotate-mode window (reference span + variant) of im_support_batch's LDS form; longer
                                   windows and queries beyond IM_MAX_READ run in its second form */
#define IM_MAX_OPS   64         /* packed segment words per realigned read           */
#define IM_MAX_EV    4          /* indel segments (= evidence) per realigned read    */

#define IM_CLS_INSERTION 0      /* varianttype, src/evidence.h:13-17                 */
#define IM_CLS_DELETION  1

typedef struct im_ctx im_ctx;
typedef struct im_comm im_comm;         /* an RCCL communicator (multi-GPU section below) */

/* The globals the reference path reads (src/alignment.c:3-9), set from the CLI
 * (src/indelminer.c:930-944): -k, -g, -s, -n. */
typedef struct im_params {
    uint32_t klength;           /* 2..15, default 6    */
    uint32_t numgaps;           /* default 0           */
    uint32_t maxdelsize;        /* default 1000        */
    uint32_t ethreshold;        /* default 10          */
} im_params;

/* One evidence record = one D/I segment of a realigned read
 * (new_evidence, src/evidence.c:4-34) plus the per-evidence reductions
 * print_variants / print_vcf_output take over aln1/aln3 (src/variant.c:217-290,704-775). */
typedef struct im_evidence {
    int32_t cls;                /* IM_CLS_*                                          */
    int32_t b1, b2;             /* segment [start,end) on the contig, 0-based        */
    int32_t seg;                /* index of the segment in ops[]                     */
    int32_t read_off;           /* read offset of the segment's first base           */
    int32_t lflank, rflank;     /* M/=/X/I bases left / right of the segment         */
    int32_t nd_print;           /* X+I+D bases in aln1+aln3 (DF=)                    */
    int32_t nd_filter;          /* nd_print + soft-clipped bases (-f filter)         */
} im_evidence;

/* One raw band alignment (attempt_band_alignment, src/alignment.c:343-391). */
typedef struct im_band_aln {
    int32_t r1, r2, q1, q2;     /* 0-based half-open contig / read coordinates       */
    int32_t low;                /* diagonal chosen by find_best_band                 */
    int32_t votes;              /* k-mer votes on that band                          */
    int32_t win_bytes;          /* reference window bytes scanned (roofline book-keeping) */
    int32_t piece_bytes;        /* read piece bytes                                  */
} im_band_aln;

/* What attempt_pe_alignment (src/alignment.c:764-799) leaves behind for one read. */
typedef struct im_read_result {
    int32_t status;             /* IM_ST_*                                           */
    int32_t ref_start;          /* contig coordinate of the first segment            */
    int32_t n_ops;              /* words valid in ops[]                              */
    int32_t n_ev;               /* records valid in ev[]                             */
    int32_t n_band;             /* band searches done (0, 1 or 2)                    */
    int32_t reserved[7];        /* pads the record to 512 bytes */
    im_band_aln band[2];
    im_evidence ev[IM_MAX_EV];  /* in segment order, left to right                   */
    uint32_t ops[IM_MAX_OPS];   /* final segment list, update_readsegs (src/readaln.c:348-458) */
} im_read_result;

/* A batch of candidate reads, struct-of-arrays.  Replaces the per-read
 * arguments of attempt_pe_alignment(sequences, tid, position, range, rln)
 * (src/alignment.h:21-25, call sites src/indelminer.c:411,486). */
typedef struct im_read_batch {
    int32_t        n;           /* reads                                             */
    const uint8_t* bases;       /* ASCII read bases, concatenated; already reverse-
                                   complemented where the caller decided so
                                   (src/indelminer.c:404-409,479-484)                */
    const int64_t* base_off;    /* n+1 offsets into bases                            */
    const int32_t* tid;         /* mate contig  (core.mtid)                          */
    const int32_t* anchor;      /* mate position (core.mpos)                         */
    const int32_t* range_max;   /* range[1] of the read group                        */
} im_read_batch;

/* ---- context ----------------------------------------------------------- */

/* Open HIP device `device` (must be gfx950).  One context per GPU. */
int  im_ctx_create(int device, im_ctx** out);
void im_ctx_destroy(im_ctx* ctx);
/* Last error text of this context (or of im_ctx_create when ctx == NULL). */
const char* im_last_error(const im_ctx* ctx);
int  im_abi_version(void);

/* Make the reference contigs resident in HBM.  seqs[i] is contig i exactly as
 * read_reference keeps it (upper-cased ASCII, src/shared.c:46-82); lens[i] its
 * length.  Replaces the `char** sequences` argument of attempt_pe_alignment. */
int im_set_reference(im_ctx* ctx, int32_t n_contigs,
                     const char* const* seqs, const int64_t* lens);

/* ---- seam 1: split-read realignment ------------------------------------ */

/* Realign every read of the batch: per read, the exact result of
 * attempt_pe_alignment (src/alignment.c:764-799).  out[] has batch->n entries.
 * Returns IM_OK, or IM_E_ABORT / IM_E_OVERFLOW / IM_E_UNSUPPORTED if any read
 * ended in the corresponding status (all results are still written). */
int im_realign_batch(im_ctx* ctx, const im_params* params,
                     const im_read_batch* batch, im_read_result* out);

/* ---- seam 2: split-read clustering -------------------------------------- */

/* The split-read part of process_evidence (src/indelminer.c:117-209 with the
 * SR rule of add_node, src/graph.c:122-127): evidence arrives as parallel
 * arrays in ARRIVAL order; it is sorted by (b1,b2), cut at the first
 * b2 >= marker, and grouped by identical (cls,b1,b2).
 *   order[n]     evidence indices, cluster after cluster, clusters ascending
 *                in (b1,b2); members ascending in arrival (tie_desc = 0, what
 *                glibc's stable qsort yields) or descending (tie_desc = 1, the
 *                order test_data/indelminer.expected.vcf was made with)
 *   cl_first/cl_count[n]  slice of order[] per cluster
 *   used[n]      1 for every evidence that became a graph node
 *   n_clusters   number of clusters
 */
int im_cluster_sr(im_ctx* ctx, int32_t n,
                  const int32_t* cls, const int32_t* b1, const int32_t* b2,
                  int32_t marker, int32_t tie_desc,
                  int32_t* order, int32_t* cl_first, int32_t* cl_count,
                  uint8_t* used, int32_t* n_clusters);

/* ---- seam 3: region depth (DP=) ------------------------------------------------- */

/* Replaces calculate_cov_params(bam_name, tid, start, stop) (src/shared.c:178-212), which the
 * reference calls once per printed variant and which re-opens the BAM, reloads the index and
 * pileups the region each time.  im_depth_build takes, once per contig, the M/=/X segments
 * (contig start, length) of every record samtools' pileup would count (not unmapped, secondary,
 * QC-fail or duplicate; bam_pileup.c:171-172) and leaves the per-position depth resident on the
 * device; im_depth_query returns, per query, the SUM of depths over [beg, end) -- the caller
 * divides by the length and floors (src/shared.c:205). */
int im_depth_build(im_ctx* ctx, int64_t contig_len, int32_t n_seg,
                   const int32_t* seg_start, const int32_t* seg_len);
int im_depth_query(im_ctx* ctx, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out);

/* ---- seam 4: annotate mode, "is this known indel supported by this read?" ------- */

/* Replaces the Smith-Waterman inside realign_with_indel (src/variant.c:1246-1424), called from
 * check_for_indel (1427-1556) for every read overlapping a known split-read variant that the
 * discovery pass did not re-find.  Task i aligns query i (the aligned part of a read,
 * read[qstart,qstop)) against target i (the reference window with the variant applied, built by
 * the caller exactly as 1260-1272 do) and returns the three counts check_for_indel compares with
 * the read's existing alignment (1549-1553): substitutions, inserted+deleted bases, aligned bases.
 * targets/queries: concatenated bytes with n+1 offsets.  out: n x 4 int32 {subs, indels, aligned,
 * status (IM_ST_EVIDENCE = valid; IM_ST_UNSUPPORTED only for a query beyond 2^20 bases)}.  Targets and queries of any
 * length: within IM_MAX_SW_TARGET / IM_MAX_READ the task runs with its boundary row in LDS, beyond in device memory. */
int im_support_batch(im_ctx* ctx, int32_t n,
                     const uint8_t* targets, const int64_t* t_off,
                     const uint8_t* queries, const int64_t* q_off, int32_t* out);

/* ---- device-resident level --------------------------------------------- */

/* Device buffers of one realign batch.  All pointers are device pointers owned
 * by the caller (hipMalloc or a torch tensor's data_ptr).  bases must hold
 * each read at a 4-byte aligned offset (base_off[i] % 4 == 0) and 8 spare
 * bytes after the last read. */
typedef struct im_dev_batch {
    int32_t        n;
    const uint8_t* bases;
    const int64_t* base_off;    /* n   start of read i in bases, multiple of 4 */
    const int32_t* read_len;    /* n   */
    const int32_t* tid;
    const int32_t* anchor;
    const int32_t* range_max;
    im_read_result* out;        /* n   */
    /* optional evidence SLOT arrays, n * IM_MAX_EV entries each (NULL = not wanted):
     * slot i*IM_MAX_EV+k carries evidence k of read i, cls = -1 marks an empty slot.
     * Slot order is arrival order, so the arrays feed im_dev_cluster_slots directly. */
    int32_t* ev_cls;
    int32_t* ev_b1;
    int32_t* ev_b2;
} im_dev_batch;

/* Launch the realign kernel on `stream` (a hipStream_t, NULL = default stream).
 * Asynchronous: returns after the launch. */
int im_dev_realign(im_ctx* ctx, const im_params* params,
                   const im_dev_batch* batch, void* stream);

/* The reference realigns reads of any length with a band of any width (src/readaln.c:242-267, src/indelminer.c:934,948).
 * Here reads of up to 255 bases run in the kernels laid out for them (four read positions per lane; numgaps <= 60: a lane
 * per band diagonal); reads of 256 .. IM_MAX_READ bases at numgaps == 0 (2 x 300 chemistry) take a second launch with
 * sixteen positions per lane; reads beyond IM_MAX_READ, reads beyond 255 bases with numgaps > 0, and every read when
 * numgaps > 60 take a general pass (one lane per read, its state in device memory; it synchronises the stream once and
 * cannot be captured into a launch graph).  Every im_dev_realign* call issues the later launches behind the first once the
 * context has been told that such reads occur: max_len = the longest read seen so far (the value only ever grows; callable
 * from any thread).  im_realign_batch (host buffers) calls it by itself.  Without the call a read beyond 255 bases comes
 * back IM_ST_UNSUPPORTED. */
int im_expect_read_length(im_ctx* ctx, int32_t max_len);

/* The results' trip to the host.  attempt_pe_alignment returns NULL for most candidates (src/alignment.c:764-799), and the
 * caller reads a candidate's 512-byte record only when it holds realigned evidence (src/indelminer.c:494-502): this packs the
 * records with status == IM_ST_EVIDENCE and n_ev > 0 into compact[] (capacity n records, order unspecified) and leaves per
 * read its status in status[i] and its place in compact[] -- or -1 -- in slot[i]; *count (a device int32, set by the call)
 * receives the number of packed records.  n_dev: device-resident batch size (NULL: n).  Asynchronous. */
int im_dev_compact_results(im_ctx* ctx, const im_read_result* res, int32_t n, const int32_t* n_dev,
                           int32_t* status, int32_t* slot, im_read_result* compact, int32_t* count, void* stream);

/* Bytes of device scratch im_dev_cluster_sr needs for n evidence records. */
size_t im_dev_cluster_scratch_bytes(int32_t n);

/* Device form of im_cluster_sr.  The record count is read from device memory
 * (*n_dev, e.g. the n_out of im_dev_gather_evidence) so that no host round trip
 * sits between the stages; n_cap is the capacity the arrays and the scratch
 * were sized for.  n_clusters is a device int32.  Asynchronous. */
int im_dev_cluster_sr(im_ctx* ctx, int32_t n_cap, const int32_t* n_dev,
                      const int32_t* cls, const int32_t* b1, const int32_t* b2,
                      int32_t marker, int32_t tie_desc,
                      int32_t* order, int32_t* cl_first, int32_t* cl_count,
                      uint8_t* used, int32_t* n_clusters,
                      void* scratch, size_t scratch_bytes, void* stream);

/* Single-launch form for evidence SLOT arrays (see im_dev_batch): compacts the live
 * slots (cls >= 0) in arrival order and clusters them in one workgroup.  Holds up
 * to im_dev_cluster_slots_max() live records (a READCHUNK flush of the reference is
 * a few thousand); with more, counts[0] is set to -1 and the caller takes the
 * im_dev_gather_evidence + im_dev_cluster_sr path instead.  order[] lists slot
 * indices; used[] (n_slots, may be NULL) marks slots that became graph nodes;
 * counts (device int32[2]) = {clusters, live records}.  Asynchronous. */
int im_dev_cluster_slots(im_ctx* ctx, int32_t n_slots,
                         const int32_t* cls, const int32_t* b1, const int32_t* b2,
                         int32_t marker, int32_t tie_desc,
                         int32_t* order, int32_t* cl_first, int32_t* cl_count,
                         uint8_t* used, int32_t* counts, void* stream);
int im_dev_cluster_slots_max(void);

/* Breakpoint-histogram form (the default for evidence SLOT arrays): hash the live slots into
 * a table of distinct (b1,b2,class) breakpoints with their support, sort only the distinct
 * breakpoints, drop every record into its cluster's slice and order the slice by arrival.
 * Four launches, any n_slots.  Same outputs as im_dev_cluster_slots.  Limits: 8192 distinct
 * breakpoints and 1024 records per breakpoint per call; beyond them counts[0] = -1 and the
 * caller takes the im_dev_gather_evidence + im_dev_cluster_sr path (after a distinct-key
 * overflow the scratch must be re-initialised).  scratch: im_dev_cluster_hist_scratch_bytes(n_slots)
 * bytes, prepared ONCE with im_dev_cluster_hist_init; every call leaves it clean.  The scratch is laid out for
 * the n_slots it was prepared with: call with THAT n_slots every time (pad the slot arrays with cls = -1), or prepare it
 * again.  Asynchronous. */
size_t im_dev_cluster_hist_scratch_bytes(int32_t n_slots);
int im_dev_cluster_hist_init(im_ctx* ctx, int32_t n_slots, void* scratch, size_t scratch_bytes, void* stream);
int im_dev_cluster_hist(im_ctx* ctx, int32_t n_slots,
                        const int32_t* cls, const int32_t* b1, const int32_t* b2,
                        int32_t marker, int32_t tie_desc,
                        int32_t* order, int32_t* cl_first, int32_t* cl_count,
                        uint8_t* used, int32_t* counts,
                        void* scratch, size_t scratch_bytes, void* stream);

/* Gather the evidence records of a realigned batch into dense SoA arrays
 * (arrival order = read order, then segment order), the input format of
 * im_dev_cluster_sr.  n_out is a device int32 (number of records written);
 * src[] receives read_index*IM_MAX_EV + k for each record.  cap = capacity of
 * the output arrays.  Asynchronous. */
int im_dev_gather_evidence(im_ctx* ctx, const im_read_result* res, int32_t n,
                           int32_t* cls, int32_t* b1, int32_t* b2, int32_t* src,
                           int32_t cap, int32_t* n_out, void* scratch, size_t scratch_bytes,
                           void* stream);
size_t im_dev_gather_scratch_bytes(int32_t n);

/* ---- seam 0: record triage -- fetch_func's candidate rules on the device ------- */

/* What fetch_func (src/indelminer.c:339-515) decides per delivered BAM record, done for a whole
 * chunk of records at once: the flag / pairing filters (348-366), the read-group lookup (369-376),
 * the three candidate cases with their mapping-quality gates (384-515), new_unaligned_readaln's
 * 4-bit -> ASCII decode and the reverse complement (src/readaln.c:242-267, src/indelminer.c:404-409,
 * 479-484), check_variants' CIGAR-derived evidence (285-337), and the read filter + match segments
 * of the DP= pileup (src/shared.c:160-176, bam_pileup.c:171-172).  Discordant pairs (516-615) stay
 * with the host's pair table; the kernel only labels them.
 *
 * Record classes (rec_class[], one byte per record): */
#define IM_REC_SKIP          0  /* returned before the read was counted (348-366)                */
#define IM_REC_COUNTED       1  /* counted (reaches 617), nothing to do                          */
#define IM_REC_CAND_UNMAPPED 2  /* unmapped read, mapped mate: realign (384-424)                 */
#define IM_REC_CAND_PROPER   3  /* proper pair with S/I/D: realign (425-515)                     */
#define IM_REC_PE            4  /* not a proper pair, passes 519-521: the host's readpairs path  */
#define IM_REC_ERR_RG       16  /* read group not in the insert-length table (must_find_hashtable exits) */
#define IM_REC_ERR_MQ       17  /* MQ tag of an unmapped read is not an integer (forceassert 395-397) */
#define IM_REC_ERR_CIGAR    18  /* N/H/P or unknown CIGAR op in a proper pair (new_readseg_bam exits) */
#define IM_REC_ERR_CLIP     19  /* soft clip inside the CIGAR (forceassert in check_variants, 325) */
#define IM_REC_ERR_BASE     20  /* base code outside A,C,G,T,N in a candidate (bit2char exits)   */
#define IM_REC_ERR_LIMIT    21  /* more than IM_MAX_EV CIGAR-derived evidence, or a malformed record */

/* the read-group -> range[1] table with the reference's hashtable semantics (16 bins, chains in
 * prepend order, prefix match, last hit wins: src/hashtable.c:62-81, src/hashfunc.c:23-30).
 * names[] in the ORDER THEY WERE ADDED to the table (config file order / first-seen order). */
int im_set_insert_ranges(im_ctx* ctx, int32_t n, const char* const* names, const int32_t* range_max);

/* A chunk of delivered records on the device: each record is the 32-byte BAM core followed by its
 * variable part (qname, cigar, seq, qual, aux) exactly as in the file, WITHOUT the block_size word,
 * starting at a 4-byte aligned offset.  A record whose core carries bin = 0xFFFF (no bin of any record: bins end at 37449)
 * comes WITHOUT its l_seq quality bytes -- (qname, cigar, seq, aux): nothing on the path reads qualities, and they are half
 * of every record's bytes over PCIe and out of HBM.  The deliverer may only leave them out when the CIGAR's read bases do not
 * exceed l_seq (the reference reads on behind the packed bases otherwise, src/readaln.c:186-240).  rec_off has n + 1 entries; record i spans rec_off[i]..rec_off[i+1], which may
 * include up to three bytes of alignment padding behind the aux area: the tag walk treats a tail shorter than the
 * smallest possible field (tag, type, one value byte = 4 bytes) as the end of the record.  The buffer behind `raw` must stay
 * readable for 64 bytes past the last record (the kernels read whole 16-byte pieces and a short look-ahead). */
typedef struct im_dev_records {
    int32_t         n;
    const uint8_t*  raw;
    const uint32_t* rec_off;
    int32_t         rec_base;   /* index of record 0 in the caller's numbering (goes into cand_rec) */
} im_dev_records;

typedef struct im_triage_params {
    int32_t  qthreshold;            /* -q */
    uint32_t ethreshold_vcfcheck;   /* -n (0 in annotate mode, src/indelminer.c:1074) */
    uint32_t maxpedelsize;          /* -p */
    int32_t  want_depth;            /* scatter the pileup match segments into the genome-wide difference array */
    int32_t  defer_ranges;          /* 1: the insert-length table is not known yet (it is being estimated in the same pass over the BAM):
                                     * no read-group look-up, range_max[] is left 0 for the caller to fill in, and EVERY pair that passes the
                                     * other tests of the discordant rule is labelled IM_REC_PE (the caller applies |isize| > range[1]) */
    int32_t  restart;               /* 1: this call opens a new batch -- counters[0..4] count as zero whatever they hold (saves the
                                     * caller a memset launch per batch); 0: the call appends to the running counters */
} im_triage_params;

/* Candidate batch under construction.  Candidates are APPENDED in record order: counters[0] = candidates
 * so far, counters[1] = read bytes so far (both updated by every call), counters[2] = records counted,
 * counters[3] = records with an IM_REC_ERR_* class.  batch holds the device arrays (capacity cap_cand
 * reads / cap_bases bytes); batch.n is ignored.  The evidence slots of a candidate receive its
 * CIGAR-derived evidence (check_variants); im_dev_realign later REPLACES them when the realignment
 * finds evidence (src/indelminer.c:494-512) -- launch it with keep_slots = 1 (im_dev_realign_keep). */
typedef struct im_dev_cands {
    im_dev_batch batch;
    int32_t*     cand_rec;      /* cap_cand: record index (rec_base + i) of every candidate      */
    int32_t*     counters;      /* device int32[8]; zero it before the first call (im_dev_alloc does not) */
    uint8_t*     rec_class;     /* n records of the chunk (may be NULL)                          */
    int32_t      cap_cand;
    int64_t      cap_bases;
    int32_t*     consumed;      /* optional: the flush marks of the evidence slots (im_dev_flush_*); a new candidate's
                                   IM_MAX_EV marks are cleared here, so that no separate fill is needed per batch    */
} im_dev_cands;

size_t im_dev_triage_scratch_bytes(int32_t n_records);
/* once per scratch buffer, before its first use (asynchronous; the launches leave it ready for the next one) */
int im_dev_triage_scratch_init(im_ctx* ctx, int32_t n_records, void* scratch, size_t scratch_bytes, void* stream);
int im_dev_triage(im_ctx* ctx, const im_triage_params* tp, const im_dev_records* recs,
                  const im_dev_cands* out, void* scratch, size_t scratch_bytes, void* stream);

/* im_dev_realign that leaves the evidence slots of reads WITHOUT realigned evidence untouched (the
 * CIGAR-derived evidence im_dev_triage put there survives, as at src/indelminer.c:504-510). */
int im_dev_realign_keep(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, void* stream);
/* the same with the batch size read from DEVICE memory (*n_dev, e.g. counters[0] of im_dev_cands), so that no host
 * round trip sits between triage and realignment; batch->n is the upper bound the launch is sized for. */
int im_dev_realign_n(im_ctx* ctx, const im_params* params, const im_dev_batch* batch, const int32_t* n_dev, int32_t keep_slots, void* stream);

/* ---- seam 2, streaming form: the READCHUNK flushes on the device ---------------- */

/* process_evidence's node selection (src/indelminer.c:123-146) for one flush: among the PENDING
 * entries (cls >= 0, consumed[] == 0) of up to two slot ranges, the entries that sort before the
 * first (b1,b2)-sorted entry with b2 >= marker become graph nodes: consumed[slot] = flush_id (> 0).
 * Range A = split-read evidence slots, range B = the host's paired-read evidence (cls = 2), which
 * takes part in the cut but is clustered on the host.  cut_word: one device uint64 PER FLUSH, set to
 * all ones beforehand (it receives the (b1,b2) of the cutting entry).  Asynchronous. */
int im_dev_flush_cut(im_ctx* ctx, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                     int32_t a0, int32_t a1, int32_t b0, int32_t b1_end,
                     int32_t marker, int32_t flush_id, uint64_t* cut_word, void* stream);
/* the same with range A given as RECORD bounds [rec0, rec1): the slots of the candidates whose record index
 * (cand_rec[], ascending, *n_cand_dev of them, both on the device as im_dev_triage left them) lies inside --
 * the flush points are known to the host as record counts, the candidate list only to the device. */
int im_dev_flush_cut_rec(im_ctx* ctx, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         int32_t rec0, int32_t rec1, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap,
                         int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id, uint64_t* cut_word, void* stream);

/* Every flush of a batch of contigs in ONE launch, in list order (the order of the file): desc[f] gives flush f's
 * record bounds [rec0, rec1) (records of its contig up to the flush point), its paired-read entries [pe0, pe1)
 * relative to slot pe_base, its marker and its id (> 0).  desc lives on the device.  A flush sees a few thousand
 * pending slots at the reference's READCHUNK, so one workgroup walks the list; callers with very long pending
 * ranges (no mid-contig flush ever consumes anything) use im_dev_flush_cut_rec per flush instead.
 * cls, b1, b2 and consumed are read a candidate (IM_MAX_EV slots = 16 bytes) at a time: their bases must be 16-byte aligned
 * (any im_dev_alloc / hipMalloc pointer is). */
typedef struct im_flush_desc {
    int32_t rec0, rec1, pe0, pe1, marker, id;
    int32_t last;               /* index (in the list) of the LAST flush of the same contig; read by im_dev_flush_groupby only */
    int32_t reserved;
} im_flush_desc;
int im_dev_flush_cuts(im_ctx* ctx, const im_flush_desc* desc_dev, int32_t n_flushes,
                      const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                      const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap,
                      int32_t pe_base, int32_t pe_count /* their marks are cleared first */, void* stream);

/* The flush list AND the split-read group-by of a group of contigs in three chip-wide launches (the product's device
 * stage; im_dev_flush_cuts + im_dev_cluster_groupby are the sequential form of the same thing).
 *
 * Why no history is needed: find_marker (src/indelminer.c:211-233) is a minimum over the pair table, whose entries only
 * leave it or enter it at the current read position of a coordinate-sorted walk, and the marker of a flush is
 * min(that, current position) (622-623) -- so WITHIN A CONTIG THE MARKERS NEVER DECREASE (the end-of-contig flush has
 * INT_MAX, 806).  An entry a flush f' consumed sorted in front of f' s cutting entry, hence had b2 < marker(f') <=
 * marker(f) for every later flush f of the contig: it could not be f's cutting entry even if it were still pending.
 * So the cut of flush f is simply  cut(f) = min{ (b1,b2) of e : e arrived before f's bounds, b2(e) >= marker(f) }  over
 * ALL entries of the contig, and  consumed(e) = the first flush f at or after e's arrival with (b1,b2)(e) < cut(f).
 * Launch 1 gives every entry's key to the cuts of the (contiguous, usually empty) run of flushes it is a candidate
 * for; launch 2 marks every entry and enters the consumed split-read slots into the cluster table; launch 3 writes
 * each cluster's record and its members in arrival order.
 *
 * THE CALLER GUARANTEES: desc[] lists the flushes in file order, rec1 and pe1 never decrease along the list, the flushes
 * of one contig are consecutive, carry the index of the contig's last flush in `last`, and their markers never
 * decrease (true for every coordinate-sorted BAM; a caller that finds otherwise uses im_dev_flush_cuts).  Entries
 * that no flush of their contig consumes keep consumed = 0.  Outputs as im_dev_cluster_groupby; counts must be 8-byte
 * aligned; the slot arrays 16-byte aligned.  The scratch is prepared once (im_dev_flushgroup_scratch_init) for a slot
 * and a flush capacity and every call leaves it ready for the next one. */
size_t im_dev_flushgroup_scratch_bytes(int32_t n_slots_cap, int32_t n_flushes_cap);
int im_dev_flushgroup_scratch_init(im_ctx* ctx, int32_t n_slots_cap, int32_t n_flushes_cap, void* scratch, size_t scratch_bytes, void* stream);
int im_dev_flush_groupby(im_ctx* ctx, const im_flush_desc* desc_dev, int32_t n_flushes,
                         const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap /* also the launch bound */,
                         int32_t pe_base, int32_t pe_count, int32_t tie_desc,
                         int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                         void* scratch, size_t scratch_bytes, void* stream);

/* The split-read rule of add_node (src/graph.c:122-127) over every consumed slot of [0, n_slots):
 * one cluster per distinct (consumed flush, class, b1, b2).  Output: cl_key[4 * c] = {flush_id, cls, b1,
 * b2}, cl_first[c], cl_count[c] in no particular cluster order (the host orders the few clusters; the
 * per-evidence work is done here); order[] = slot indices, cluster after cluster, members ascending in
 * slot index (= arrival) or descending with tie_desc.  counts (device int32[2]) = {clusters, nodes}.
 * Entries with cls >= 2 are ignored. */
size_t im_dev_groupby_scratch_bytes(int32_t n_slots);
/* once per scratch buffer (asynchronous); every group-by call leaves the scratch ready for the next one.  The scratch is
 * laid out for THIS n_slots for its whole life: later calls may pass any n_slots up to it (the context remembers the
 * layout per scratch pointer; a scratch that was never initialised is refused) */
int im_dev_groupby_scratch_init(im_ctx* ctx, int32_t n_slots, void* scratch, size_t scratch_bytes, void* stream);
int im_dev_cluster_groupby(im_ctx* ctx, int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                           const int32_t* consumed, int32_t tie_desc,
                           int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                           void* scratch, size_t scratch_bytes, void* stream);
/* n_slots = IM_MAX_EV * *n_cand_dev (device), at most n_slots_cap */
int im_dev_cluster_groupby_n(im_ctx* ctx, int32_t n_slots_cap, const int32_t* n_cand_dev,
                             const int32_t* cls, const int32_t* b1, const int32_t* b2,
                             const int32_t* consumed, int32_t tie_desc,
                             int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                             void* scratch, size_t scratch_bytes, void* stream);

/* ---- seam 3, genome-wide form ----------------------------------------------------- */

/* One int32 per reference position for ALL contigs (4 bytes per base of HBM), filled by im_dev_triage
 * (want_depth) as a difference array; im_depth_scan turns contig tid into depths once all its records
 * have been through triage (one launch: depths local to 8192-position tiles plus a tile offset each, which
 * only im_depth_query_tid knows how to read); im_depth_query_tid sums [beg,end) like im_depth_query. */
int im_depth_enable(im_ctx* ctx);
int im_depth_scan(im_ctx* ctx, int32_t tid, void* stream);
/* contig tid's run back to zeros (asynchronous): a contig that is to go through triage + im_depth_scan AGAIN */
int im_depth_reset(im_ctx* ctx, int32_t tid, void* stream);
int im_depth_query_tid(im_ctx* ctx, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out);
/* the same, and per query the DEEPEST position of [beg - 1, end] (max_out, may be NULL).  The depth array counts every record; samtools'
 * pileup, which the reference's DP= comes from (src/shared.c:178-212), stops buffering records that start at the position it stands on
 * once 8000 are buffered (src/samtools-0.1.19/bam_pileup.c:172,244): the host driver asks the file, with that rule, about the queries
 * whose maximum says the rule may have applied. */
int im_depth_query_max_tid(im_ctx* ctx, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out, uint32_t* max_out);
/* Multi-GPU, pieces of one contig walked by several ranks: every rank's difference array holds the +-1 of the records IT
 * delivered; their sum (one RCCL all-reduce over the whole array, before any im_depth_scan) is the single run's array.  The
 * reference has no counterpart (calculate_cov_params re-reads the file per variant, src/shared.c:178-212).  Synchronous. */
int im_depth_allreduce(im_ctx* ctx, im_comm* comm);

/* ---- multi-GPU: one collective ------------------------------------------------ */

/* Contigs are independent (the reference's own parallel mode is one process per -c
 * region, src/indelminer.c:536-542), so ranks own disjoint contigs and exchange
 * nothing until each holds its cluster list; then ONE all-gather (RCCL over xGMI).
 * Rendezvous: rank 0 calls im_comm_unique_id and ships the IM_COMM_ID_BYTES to the
 * other ranks by any side channel (bench.py: torch.distributed/gloo broadcast). */
#define IM_COMM_ID_BYTES 128
int  im_comm_unique_id(void* id_bytes);
int  im_comm_init(im_ctx* ctx, const void* id_bytes, int rank, int world, im_comm** out);
/* every rank contributes bytes_per_rank bytes; recv_dev holds world * bytes_per_rank.  Asynchronous. */
int  im_comm_allgather(im_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
/* in place: buf[i] = sum over ranks of buf[i].  Asynchronous. */
int  im_comm_allreduce_sum_i32(im_comm* comm, int32_t* buf_dev, size_t count, void* stream);
/* Point-to-point traffic of one exchange step, device memory to device memory over RCCL (xGMI): n operations issued as ONE group.
 * Operation k SENDS bytes[k] bytes at dev[k] to rank peer[k] (dir[k] = 0) or RECEIVES them from it (dir[k] = 1).  Every rank passes
 * the operations it takes part in, all ranks in the same global order (the host driver: walked groups in claim order, a group's host
 * part in front of its device arrays), so that sends and receives pair up whatever the ranks' timing.  Asynchronous.
 * Replaces nothing in the reference (it has no communication: src/indelminer.c:536-542); it is how a group walked by one rank
 * reaches the rank that owns its contig. */
int  im_comm_exchange(im_comm* comm, int32_t n, const int32_t* dir, const int32_t* peer, void* const* dev, const size_t* bytes, void* stream);
void im_comm_destroy(im_comm* comm);
const char* im_comm_last_error(void);

/* One 16-byte record per cluster, the unit that is gathered: {tid, b1, b2, cls<<24 | support}.
 * recs[0] = {n_clusters, n_live_evidence, tid, 0}; cluster c at recs[1 + c]; cap = records the
 * buffer holds (clusters beyond cap-1 are dropped and recs[0].w is set to 1).  counts = the
 * device int32[2] written by im_dev_cluster_slots / n_clusters of im_dev_cluster_sr. */
int im_dev_cluster_records(im_ctx* ctx, int32_t tid, const int32_t* counts,
                           const int32_t* order, const int32_t* cl_first, const int32_t* cl_count,
                           const int32_t* cls, const int32_t* b1, const int32_t* b2,
                           int32_t* recs, int32_t cap, void* stream);

/* ---- device memory / timing plumbing for callers without a HIP binding --- */

/* hipMalloc / hipFree / hipMemcpy on the context's device.  im_dev_upload and
 * im_dev_download are synchronous. */
int  im_dev_alloc(im_ctx* ctx, size_t bytes, void** out);
int  im_dev_free(im_ctx* ctx, void* p);
int  im_dev_upload(im_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  im_dev_download(im_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int  im_dev_memset(im_ctx* ctx, void* dst_dev, int byte, size_t bytes, void* stream);     /* asynchronous */
/* Pinned host memory and asynchronous copies on a stream ("reads are pre-staged into pinned buffers and
 * hipMemcpyAsync'd"): the host driver inflates BAM records straight into im_host_alloc'd chunks. */
int  im_host_alloc(im_ctx* ctx, size_t bytes, void** out);
int  im_host_free(im_ctx* ctx, void* p);
int  im_dev_upload_async(im_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int  im_dev_download_async(im_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int  im_dev_copy_async(im_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream);
/* The context's own stream (a hipStream_t) and a wait for it. */
void* im_ctx_stream(im_ctx* ctx);
int   im_ctx_device(im_ctx* ctx);                               /* the HIP device index the context lives on */
int  im_stream_sync(im_ctx* ctx, void* stream);
/* HIP-event stopwatch on a stream: create, record start / stop on the stream the
 * kernels are launched on, read the elapsed milliseconds (synchronises on stop). */
typedef struct im_timer im_timer;
int  im_timer_create(im_ctx* ctx, im_timer** out);
void im_timer_destroy(im_timer* t);
int  im_timer_start(im_timer* t, void* stream);
int  im_timer_stop(im_timer* t, void* stream);
int  im_timer_elapsed_ms(im_timer* t, float* ms);

/* A second stream and stream-to-stream events, so that the cluster kernels of one flush can run while the
 * next flush's realign kernel does (two sets of realign output buffers; the reference has no counterpart,
 * its flushes are sequential host code, src/indelminer.c:617-640). */
typedef struct im_event im_event;
int  im_stream_create(im_ctx* ctx, void** out);
int  im_stream_destroy(im_ctx* ctx, void* stream);
int  im_event_create(im_ctx* ctx, im_event** out);
void im_event_destroy(im_event* ev);
int  im_event_record(im_event* ev, void* stream);
int  im_event_sync(im_event* ev);                               /* host waits */
int  im_stream_wait_event(im_ctx* ctx, void* stream, im_event* ev);
int  im_stream_follow(im_event* ev, void* from, void* to);      /* record on `from`, `to` waits */

/* Launch graphs.  One flush is a fixed sequence of dependent im_dev_* launches on one stream
 * (the reference has no counterpart: its flush, src/indelminer.c:617-640, is host code).  Between
 * im_capture_begin and im_capture_end the im_dev_* calls on `stream` are recorded instead of run;
 * im_graph_launch replays them with a single host call.  Device buffers and sizes are baked in:
 * capture again when they change. */
typedef struct im_graph im_graph;
int  im_capture_begin(im_ctx* ctx, void* stream);
int  im_capture_end(im_ctx* ctx, void* stream, im_graph** out);
int  im_graph_launch(im_graph* g, void* stream);
void im_graph_destroy(im_graph* g);

#ifdef __cplusplus
}
#endif
#endif
