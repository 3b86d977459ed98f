/*
 * hostio.h -- host-side file formats of the indelminer driver: BGZF, BAM, BAI,
 * FASTA, the IL/RC config file.  Own implementation (zlib only); samtools is not
 * used.  Semantics follow what the reference observes through libbam
 * (SURVEY.md A.12), cited per function in hostio.c.
 */
#ifndef IM_HOSTIO_H
#define IM_HOSTIO_H

#include <stdint.h>
#include <string.h>
#include <stdio.h>

/* ---- BGZF reader ---- */
typedef struct bgzf_reader bgzf_reader;
bgzf_reader* bgzf_open(const char* path);
void    bgzf_close(bgzf_reader* r);
void    bgzf_set_workers(bgzf_reader* r, int n);             /* inflate workers of this reader (before it has read far); default INDELMINER_THREADS or 4 */
int64_t bgzf_read(bgzf_reader* r, void* buf, int64_t n);    /* bytes read, < n at EOF, -1 on error */
int64_t bgzf_tell(const bgzf_reader* r);                     /* virtual offset */
int     bgzf_seek(bgzf_reader* r, int64_t voffset);

/* ---- BAM ---- */
typedef struct {
    int32_t  n_targets;
    char**   target_name;
    int32_t* target_len;
} bam_header;

typedef struct {
    int32_t  tid, pos;
    uint8_t  l_qname, mapq;
    uint16_t bin, n_cigar, flag;
    int32_t  l_seq, mtid, mpos, isize;
    uint8_t* data;          /* qname, cigar, seq, qual, aux */
    int32_t  l_data, m_data;
    int      no_qual;       /* a delivered record whose base qualities were left out (bin == BAM_BIN_NO_QUAL): aux follows the packed bases */
} bam_record;
/* The device pipeline's records travel without their base qualities (half of every record; nothing on the path reads them):
 * such a record says so in its bin field, which no part of the path uses.  See bam_region_next_raw. */
#define BAM_BIN_NO_QUAL 0xFFFFu

#define BAMR_QNAME(b)  ((char*)(b)->data)
/* CIGAR word k: the words follow the read name, at any byte alignment */
static inline uint32_t bamr_cigar_at(const uint8_t* cig8, int k) { uint32_t v; memcpy(&v, cig8 + 4 * (size_t)k, 4); return v; }
#define BAMR_CIGAR(b)  ((const uint8_t*)((b)->data + (b)->l_qname))
#define BAMR_SEQ(b)    ((b)->data + (b)->l_qname + 4 * (b)->n_cigar)
#define BAMR_AUX(b)    ((b)->data + (b)->l_qname + 4 * (b)->n_cigar + (((b)->l_seq + 1) >> 1) + ((b)->no_qual ? 0 : (b)->l_seq))
#define BAMR_SEQI(s, i) (((s)[(i) >> 1] >> ((~(i) & 1) << 2)) & 0xf)

bam_header* bam_header_load(bgzf_reader* r);
void bam_header_free(bam_header* h);
int  bam_read_record(bgzf_reader* r, bam_record* b);    /* 1 = record, 0 = EOF, -1 = error */
int32_t bam_record_end(const bam_record* b);            /* bam_calend: pos + reference-consuming ops */
const uint8_t* bam_aux_find(const bam_record* b, const char tag[2]);   /* pointer at the type byte, or NULL */
int32_t bam_aux_int(const uint8_t* s);                  /* c/C/s/S/i/I, anything else 0 (bam_aux.c:163-174) */
const char* bam_aux_str(const uint8_t* s);              /* after a 'Z' type byte */

/* ---- BAI ---- */
typedef struct bai_index bai_index;
bai_index* bai_load(const char* bam_path);              /* <bam>.bai ; NULL if missing */
void bai_free(bai_index* idx);
int64_t bai_contig_bytes(const bai_index* idx, int32_t tid);   /* compressed bytes holding the contig's records, 0 if none */

/* region iterator with bam_fetch's semantics (bam_index.c:571-576,715-726): every record of
 * tid with rend > beg && rbeg < end, in file order; rend = pos+1 for records without CIGAR */
typedef struct {
    bgzf_reader* r;
    int32_t tid, beg, end;
    int done;
    int32_t pending_size;   /* block_size of a record whose body has not been read yet (bam_region_next_raw) */
    int by_start;           /* 1: the records that START in [beg, end) (bam_piece_begin), not the ones that overlap it */
    int drop_qual;          /* bam_region_next_raw leaves the base qualities out (set by the caller after *_begin) */
} bam_region_iter;
int bam_region_begin(bam_region_iter* it, bgzf_reader* r, const bai_index* idx, int32_t tid, int32_t beg, int32_t end);
int bam_region_next(bam_region_iter* it, bam_record* b);   /* 1 = record, 0 = done, -1 = error */
/* A PIECE of a contig: every record of tid whose position lies in [beg, end), in file order -- consecutive pieces
 * [0, a), [a, b), ..., [z, length) deliver exactly the records, in the order, that bam_region_begin(tid, 0, length) does */
int bam_piece_begin(bam_region_iter* it, bgzf_reader* r, const bai_index* idx, int32_t tid, int32_t beg, int32_t end);
/* split points for pieces of about equal file size: positions (multiples of 16 kb) where the contig's compressed bytes
 * cross k * target; out[] receives up to cap of them, ascending, each in (0, length).  Returns how many. */
int bai_split_points(const bai_index* idx, int32_t tid, int32_t length, int64_t target_bytes, int32_t* out, int cap);
/* The same iteration, but the record's bytes (32-byte core + variable part, as in the file, without the
 * block_size word) land in the caller's buffer dst[0..cap) -- the host driver points it into a pinned chunk
 * that goes to the GPU as it is.  view receives the decoded core and view->data = dst + 32 (not owned).
 * Returns 1 = record (*len_out bytes written), 0 = done, -1 = error, -2 = the next record needs more than
 * cap bytes (nothing consumed: call again with a bigger / fresh buffer).
 * With it->drop_qual the l_seq quality bytes are not copied and the record's bin field reads BAM_BIN_NO_QUAL -- unless the CIGAR
 * asks for more read bases than l_seq: the reference then reads on into the bytes behind the packed bases (new_readaln,
 * src/readaln.c:186-240), so such a record keeps them. */
void bam_record_view(const uint8_t* rec, int32_t len, bam_record* view);   /* decode a raw record in place (view->data not owned) */
int bam_region_next_raw(bam_region_iter* it, uint8_t* dst, int64_t cap, int32_t* len_out, bam_record* view);

/* region string "chr", "chr:beg", "chr:beg-end" (bam_aux.c:107-161): returns 0 on success */
int bam_parse_region_str(const bam_header* h, const char* str, int* tid, int* beg, int* end);

/* ---- FASTA ---- */
/* contigs in file order, filtered by the reference's IUPAC table and upper-cased
 * (src/sequences.c:6-20,88-95; src/shared.c:66-69).  only != NULL keeps just that index. */
int fasta_load(const char* path, int32_t n_expected, char*** seqs_out, int64_t** lens_out, int only_index);

/* ---- line reader with the reference's getline quirk (src/files.c:17-55): a last line
 * without '\n' is dropped ---- */
long im_getline(char** lineptr, size_t* cap, FILE* fp);

#endif
