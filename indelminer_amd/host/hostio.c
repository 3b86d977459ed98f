/*
 * hostio.c -- BGZF / BAM / BAI / FASTA readers of the indelminer host driver.
 * Own code over zlib.  Formats: SAM/BAM specification sections 4.1-4.2, 5.
 * Behaviour the reference observes through libbam is cited as SURVEY.md A.12 items.
 */
#define _DEFAULT_SOURCE         /* madvise(MADV_HUGEPAGE) */
#define _POSIX_C_SOURCE 200809L
#include "hostio.h"

#include <ctype.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "iminflate.h"

/* ------------------------------------------------------------------ BGZF -- */
/*
 * Sequential reads run through a read-ahead ring: the reading thread fetches the compressed blocks
 * that follow (file I/O stays on one thread) and a few worker threads inflate them, so that the
 * decode passes of the driver (insert-length estimate, pass A) cost one thread's record parsing and
 * not one thread's zlib.  A seek empties the ring.  The pool starts only after a run of
 * consecutive blocks, so short region fetches never pay for threads.  INDELMINER_THREADS sets the
 * number of inflate workers (default 4, 0 = inflate on the reading thread as before).
 */

#define BGZF_MAX_BLOCK 65536
#define BGZF_RING 64
#define BGZF_WARMUP 8

enum { SLOT_FREE = 0, SLOT_LOADED, SLOT_BUSY, SLOT_DONE, SLOT_BAD };

typedef struct {
    uint8_t  cbuf[BGZF_MAX_BLOCK + 64];
    uint8_t  ubuf[BGZF_MAX_BLOCK + IM_INFLATE_SLACK];
    int64_t  coff;
    int32_t  total, hdr, ulen;
    int      state;
} bgzf_slot;

struct bgzf_reader {
    FILE*    fp;
    uint8_t  cbuf[BGZF_MAX_BLOCK + 64];
    uint8_t  ubuf[BGZF_MAX_BLOCK + IM_INFLATE_SLACK];
    int32_t  ulen, upos;
    int64_t  block_coff;        /* file offset of the block in ubuf */
    int64_t  next_coff;         /* file offset of the next block */
    int      eof;
    const uint8_t* map; int64_t map_size;   /* the file, mapped (shared by the readers of one path): blocks are inflated where they lie */
    int64_t  map_lo;            /* this reader has mapped pages in from here to next_coff since its last seek */
    /* read-ahead */
    int      nworkers, started, stop, run;
    pthread_t workers[16];
    pthread_mutex_t mu;
    pthread_cond_t  cv_work, cv_done;
    bgzf_slot* ring;            /* [BGZF_RING] */
    int      head, tail, next_job;      /* consume at head, fill at tail, inflate at next_job (ring indices, monotone) */
    int64_t  fill_coff;         /* file offset of the next block to fetch */
    int      fill_eof;
};

/* One mapping per file for all its readers (a run opens the BAM once per walker): with it a block's payload goes from the page
 * cache straight through the decoder -- no copy into the stdio buffer and from there into the reader's.  INDELMINER_BAM_MMAP=0
 * keeps the reads. */
static struct { pthread_mutex_t mu; char path[1024]; const uint8_t* map; int64_t size; } g_bam_map = { PTHREAD_MUTEX_INITIALIZER, "", NULL, 0 };
static void bgzf_map_file(bgzf_reader* r, const char* path)
{
    const char* e = getenv("INDELMINER_BAM_MMAP");
    if ((e && e[0] == '0') || strlen(path) >= sizeof g_bam_map.path) return;
    pthread_mutex_lock(&g_bam_map.mu);
    if (!g_bam_map.map || strcmp(g_bam_map.path, path) != 0) {
        struct stat sb;
        if (!g_bam_map.map && fstat(fileno(r->fp), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            void* m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_SHARED, fileno(r->fp), 0);
            if (m != MAP_FAILED) { g_bam_map.map = m; g_bam_map.size = sb.st_size; snprintf(g_bam_map.path, sizeof g_bam_map.path, "%s", path); }
        }
    }
    if (g_bam_map.map && strcmp(g_bam_map.path, path) == 0) { r->map = g_bam_map.map; r->map_size = g_bam_map.size; }
    pthread_mutex_unlock(&g_bam_map.mu);
}

bgzf_reader* bgzf_open(const char* path)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return NULL;
    bgzf_reader* r = calloc(1, sizeof *r);
    r->fp = fp;
    setvbuf(fp, NULL, _IOFBF, 256 << 10);           /* blocks are read one after the other: a few system calls per megabyte, not per block */
    const char* e = getenv("INDELMINER_THREADS");
    r->nworkers = e ? atoi(e) : 4;
    if (r->nworkers < 0) r->nworkers = 0;
    if (r->nworkers > 16) r->nworkers = 16;
    bgzf_map_file(r, path);
    return r;
}

/* the number of inflate workers of a reader that has not started its pool yet (the walkers of the driver share the cores out) */
void bgzf_set_workers(bgzf_reader* r, int n)
{
    if (!r || r->started) return;
    r->nworkers = n < 0 ? 0 : (n > 16 ? 16 : n);
}

/* reads the compressed block at coff into cbuf; returns total size, 0 at EOF, -1 on error; *phdr = header size */
static int bgzf_fetch(FILE* fp, int64_t coff, uint8_t* h, int* phdr)
{
    if (fseeko(fp, coff, SEEK_SET) != 0) return -1;
    size_t got = fread(h, 1, 18, fp);
    if (got == 0) return 0;
    if (got < 18 || h[0] != 31 || h[1] != 139 || h[2] != 8 || !(h[3] & 4)) return -1;
    const int xlen = h[10] | (h[11] << 8);
    /* find the BC subfield */
    int bsize = -1;
    if (xlen == 6 && h[12] == 'B' && h[13] == 'C') bsize = h[16] | (h[17] << 8);
    else {
        uint8_t* x = malloc((size_t)xlen);
        memcpy(x, h + 12, 6);
        if (xlen > 6 && fread(x + 6, 1, (size_t)xlen - 6, fp) != (size_t)xlen - 6) { free(x); return -1; }
        for (int i = 0; i + 4 <= xlen;) {
            const int slen = x[i + 2] | (x[i + 3] << 8);
            if (x[i] == 'B' && x[i + 1] == 'C' && slen == 2) bsize = x[i + 4] | (x[i + 5] << 8);
            i += 4 + slen;
        }
        free(x);
        if (fseeko(fp, coff + 18, SEEK_SET) != 0) return -1;
    }
    if (bsize < 0) return -1;
    const int total = bsize + 1;
    const int hdr = 12 + xlen;
    const int remain = total - 18;
    if (remain < 0 || total > BGZF_MAX_BLOCK + 64 || total - hdr - 8 < 0) return -1;
    if (fread(h + 18, 1, (size_t)remain, fp) != (size_t)remain) return -1;
    *phdr = hdr;
    return total;
}

/* raw-deflate payload of a fetched block -> ubuf (BGZF_MAX_BLOCK + IM_INFLATE_SLACK bytes); returns the inflated size or -1.
 * The driver's own decoder (iminflate.c); INDELMINER_INFLATE=zlib takes zlib's instead (a cross-check). */
static int g_use_zlib = -1;
static int bgzf_inflate(const uint8_t* h, int total, int hdr, uint8_t* ubuf)
{
    const int clen = total - hdr - 8;
    if (g_use_zlib < 0) { const char* e = getenv("INDELMINER_INFLATE"); g_use_zlib = e && strcmp(e, "zlib") == 0; }
    if (!g_use_zlib) {
        /* the block's trailer (CRC32, ISIZE) and the buffer's spare bytes follow the payload: the decoder's look-ahead stays inside */
        const uint8_t* tr = h + total - 8;
        const uint32_t isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize > BGZF_MAX_BLOCK) return -1;
        const int64_t got = im_inflate(h + hdr, (size_t)clen, ubuf, BGZF_MAX_BLOCK);
        return got == (int64_t)isize ? (int)got : -1;
    }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return -1;
    zs.next_in = (Bytef*)(h + hdr); zs.avail_in = (uInt)clen;
    zs.next_out = ubuf; zs.avail_out = BGZF_MAX_BLOCK;
    const int zr = inflate(&zs, Z_FINISH);
    const int ulen = (int)zs.total_out;
    inflateEnd(&zs);
    return zr == Z_STREAM_END ? ulen : -1;
}

static void* bgzf_worker(void* arg)
{
    bgzf_reader* r = arg;
    pthread_mutex_lock(&r->mu);
    for (;;) {
        while (!r->stop && !(r->next_job < r->tail && r->ring[r->next_job % BGZF_RING].state == SLOT_LOADED))
            pthread_cond_wait(&r->cv_work, &r->mu);
        if (r->stop) break;
        bgzf_slot* s = &r->ring[r->next_job % BGZF_RING];
        r->next_job++;
        s->state = SLOT_BUSY;
        pthread_mutex_unlock(&r->mu);
        const int ulen = bgzf_inflate(s->cbuf, s->total, s->hdr, s->ubuf);
        pthread_mutex_lock(&r->mu);
        s->ulen = ulen;
        s->state = ulen < 0 ? SLOT_BAD : SLOT_DONE;
        pthread_cond_broadcast(&r->cv_done);
    }
    pthread_mutex_unlock(&r->mu);
    return NULL;
}

static void bgzf_pool_start(bgzf_reader* r)
{
    r->ring = calloc(BGZF_RING, sizeof(bgzf_slot));
    pthread_mutex_init(&r->mu, NULL);
    pthread_cond_init(&r->cv_work, NULL);
    pthread_cond_init(&r->cv_done, NULL);
    r->head = r->tail = r->next_job = 0;
    r->stop = 0;
    int ok = 0;
    for (int i = 0; i < r->nworkers; i++)
        if (pthread_create(&r->workers[ok], NULL, bgzf_worker, r) == 0) ok++;
    r->nworkers = ok;
    r->started = ok > 0;
}

/* forget everything read ahead (a seek): wait for the inflates in flight, then empty the ring */
static void bgzf_pool_reset(bgzf_reader* r)
{
    if (!r->started) return;
    pthread_mutex_lock(&r->mu);
    for (int i = r->head; i < r->tail; i++)
        while (r->ring[i % BGZF_RING].state == SLOT_BUSY) pthread_cond_wait(&r->cv_done, &r->mu);
    for (int i = 0; i < BGZF_RING; i++) r->ring[i].state = SLOT_FREE;
    r->head = r->tail = r->next_job = 0;
    r->fill_eof = 0;
    pthread_mutex_unlock(&r->mu);
}

void bgzf_close(bgzf_reader* r)
{
    if (!r) return;
    if (r->started) {
        pthread_mutex_lock(&r->mu);
        r->stop = 1;
        pthread_cond_broadcast(&r->cv_work);
        pthread_mutex_unlock(&r->mu);
        for (int i = 0; i < r->nworkers; i++) pthread_join(r->workers[i], NULL);
        pthread_mutex_destroy(&r->mu);
        pthread_cond_destroy(&r->cv_work);
        pthread_cond_destroy(&r->cv_done);
    }
    free(r->ring);
    fclose(r->fp);
    free(r);
}

/* the read-ahead path of bgzf_load_block: block at r->next_coff from the ring */
static int bgzf_load_ahead(bgzf_reader* r)
{
    pthread_mutex_lock(&r->mu);
    if (r->head == r->tail) { r->fill_coff = r->next_coff; r->fill_eof = 0; }     /* ring empty: (re)start here */
    /* top the ring up (this thread does all the file I/O) */
    while (!r->fill_eof && r->tail - r->head < BGZF_RING) {
        bgzf_slot* s = &r->ring[r->tail % BGZF_RING];
        pthread_mutex_unlock(&r->mu);
        int hdr = 0;
        const int total = bgzf_fetch(r->fp, r->fill_coff, s->cbuf, &hdr);
        pthread_mutex_lock(&r->mu);
        s->coff = r->fill_coff;
        if (total <= 0) { s->total = total; s->state = total == 0 ? SLOT_DONE : SLOT_BAD; s->ulen = total == 0 ? -2 : -1; r->fill_eof = 1; }
        else { s->total = total; s->hdr = hdr; s->state = SLOT_LOADED; r->fill_coff += total; }
        r->tail++;
        pthread_cond_signal(&r->cv_work);
    }
    bgzf_slot* s = &r->ring[r->head % BGZF_RING];
    while (s->state == SLOT_LOADED || s->state == SLOT_BUSY) pthread_cond_wait(&r->cv_done, &r->mu);
    int rc;
    if (s->state == SLOT_BAD) rc = -1;
    else if (s->ulen == -2) { r->eof = 1; r->ulen = r->upos = 0; rc = 0; }
    else {
        memcpy(r->ubuf, s->ubuf, (size_t)s->ulen);
        r->block_coff = s->coff;
        r->next_coff = s->coff + s->total;
        r->ulen = s->ulen; r->upos = 0;
        rc = 1;
    }
    s->state = SLOT_FREE;
    r->head++;
    pthread_mutex_unlock(&r->mu);
    return rc;
}

/* loads the block at r->next_coff; returns 1, 0 at EOF, -1 on error */
static int bgzf_load_block(bgzf_reader* r)
{
    if (r->nworkers > 0 && ++r->run > BGZF_WARMUP) {      /* a run of consecutive blocks: read ahead from here on */
        if (!r->started) bgzf_pool_start(r);
        if (r->started) return bgzf_load_ahead(r);
    }
    if (r->map && r->next_coff + 18 <= r->map_size) {
        /* the usual block (one BC subfield), whole inside the mapping with the decoder's slack behind it: inflated where it lies */
        const uint8_t* h = r->map + r->next_coff;
        if (h[0] == 31 && h[1] == 139 && h[2] == 8 && (h[3] & 4) && h[10] == 6 && h[11] == 0 && h[12] == 'B' && h[13] == 'C') {
            const int total = (h[16] | (h[17] << 8)) + 1;
            if (total >= 18 + 8 && total <= BGZF_MAX_BLOCK + 64 && r->next_coff + total + IM_INFLATE_SLACK <= r->map_size) {
                const int ulen = bgzf_inflate(h, total, 18, r->ubuf);
                if (ulen < 0) return -1;
                r->block_coff = r->next_coff;
                r->next_coff += total;
                r->ulen = ulen; r->upos = 0;
                /* what lies behind is not read again: its page-table entries go now, 8 MB at a time (the pages stay in the
                 * page cache), so that the mapping never counts for more than the readers' working sets */
                if (r->map_lo > r->block_coff) r->map_lo = r->block_coff;
                if (r->block_coff - r->map_lo >= ((int64_t)8 << 20)) {
                    const int64_t a = (r->map_lo + 4095) & ~(int64_t)4095, z = r->block_coff & ~(int64_t)4095;
                    if (z > a) (void)madvise((void*)(r->map + a), (size_t)(z - a), MADV_DONTNEED);
                    r->map_lo = z;
                }
                return 1;
            }
        }
    }
    int hdr = 0;
    const int total = bgzf_fetch(r->fp, r->next_coff, r->cbuf, &hdr);
    if (total == 0) { r->eof = 1; r->ulen = r->upos = 0; return 0; }
    if (total < 0) return -1;
    const int ulen = bgzf_inflate(r->cbuf, total, hdr, r->ubuf);
    if (ulen < 0) return -1;
    r->block_coff = r->next_coff;
    r->next_coff += total;
    r->ulen = ulen; r->upos = 0;
    return 1;
}

int64_t bgzf_read(bgzf_reader* r, void* buf, int64_t n)
{
    uint8_t* out = buf;
    int64_t done = 0;
    while (done < n) {
        if (r->upos >= r->ulen) {
            if (r->eof) break;
            int rc = bgzf_load_block(r);
            if (rc < 0) return -1;
            if (rc == 0) break;
            if (r->ulen == 0) continue;     /* empty block (EOF marker) */
        }
        int64_t take = r->ulen - r->upos;
        if (take > n - done) take = n - done;
        memcpy(out + done, r->ubuf + r->upos, (size_t)take);
        r->upos += (int32_t)take;
        done += take;
    }
    return done;
}

int64_t bgzf_tell(const bgzf_reader* r)
{
    if (r->upos >= r->ulen) return r->next_coff << 16;      /* at a block boundary */
    return (r->block_coff << 16) | (int64_t)r->upos;
}

int bgzf_seek(bgzf_reader* r, int64_t voffset)
{
    bgzf_pool_reset(r);
    r->run = 0;
    r->eof = 0;
    if (r->map && r->next_coff > r->map_lo) {          /* the rest of what this reader mapped in before the seek */
        const int64_t a = (r->map_lo + 4095) & ~(int64_t)4095, z = (r->next_coff < r->map_size ? r->next_coff : r->map_size) & ~(int64_t)4095;
        if (z > a) (void)madvise((void*)(r->map + a), (size_t)(z - a), MADV_DONTNEED);
    }
    r->map_lo = voffset >> 16;
    r->next_coff = voffset >> 16;
    r->ulen = r->upos = 0;
    const int within = (int)(voffset & 0xffff);
    if (within > 0) {
        int rc = bgzf_load_block(r);
        if (rc <= 0) return -1;
        if (within > r->ulen) return -1;
        r->upos = within;
    }
    return 0;
}

/* ------------------------------------------------------------------- BAM -- */

static int rd_i32(bgzf_reader* r, int32_t* v)
{
    uint8_t b[4];
    if (bgzf_read(r, b, 4) != 4) return -1;
    *v = (int32_t)((uint32_t)b[0] | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16) | ((uint32_t)b[3] << 24));
    return 0;
}

bam_header* bam_header_load(bgzf_reader* r)
{
    char magic[4];
    if (bgzf_read(r, magic, 4) != 4 || memcmp(magic, "BAM\1", 4) != 0) return NULL;
    int32_t l_text, n_ref;
    if (rd_i32(r, &l_text)) return NULL;
    char* text = malloc((size_t)l_text + 1);
    if (bgzf_read(r, text, l_text) != l_text) { free(text); return NULL; }
    free(text);
    if (rd_i32(r, &n_ref)) return NULL;
    bam_header* h = calloc(1, sizeof *h);
    h->n_targets = n_ref;
    h->target_name = calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(char*));
    h->target_len = calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(int32_t));
    for (int32_t i = 0; i < n_ref; i++) {
        int32_t l_name;
        if (rd_i32(r, &l_name)) { bam_header_free(h); return NULL; }
        h->target_name[i] = calloc((size_t)l_name + 1, 1);
        if (bgzf_read(r, h->target_name[i], l_name) != l_name) { bam_header_free(h); return NULL; }
        if (rd_i32(r, &h->target_len[i])) { bam_header_free(h); return NULL; }
    }
    return h;
}

void bam_header_free(bam_header* h)
{
    if (!h) return;
    for (int32_t i = 0; i < h->n_targets; i++) free(h->target_name[i]);
    free(h->target_name); free(h->target_len); free(h);
}

int bam_read_record(bgzf_reader* r, bam_record* b)
{
    uint8_t c[36];
    int64_t got = bgzf_read(r, c, 4);
    if (got == 0) return 0;
    if (got != 4) return -1;
    const int32_t block_size = (int32_t)((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
    if (block_size < 32) return -1;
    if (bgzf_read(r, c, 32) != 32) return -1;
#define U32(o) ((uint32_t)c[o] | ((uint32_t)c[(o) + 1] << 8) | ((uint32_t)c[(o) + 2] << 16) | ((uint32_t)c[(o) + 3] << 24))
    b->tid = (int32_t)U32(0); b->pos = (int32_t)U32(4);
    b->l_qname = c[8]; b->mapq = c[9]; b->bin = (uint16_t)(c[10] | (c[11] << 8));
    b->n_cigar = (uint16_t)(c[12] | (c[13] << 8)); b->flag = (uint16_t)(c[14] | (c[15] << 8));
    b->l_seq = (int32_t)U32(16); b->mtid = (int32_t)U32(20); b->mpos = (int32_t)U32(24); b->isize = (int32_t)U32(28);
#undef U32
    b->l_data = block_size - 32;
    if (b->l_data + 8 > b->m_data) { b->m_data = b->l_data + 64; b->data = realloc(b->data, (size_t)b->m_data); }
    if (bgzf_read(r, b->data, b->l_data) != b->l_data) return -1;
    return 1;
}

int32_t bam_record_end(const bam_record* b)
{
    /* bam_calend (bam.c:17-39): M, D, N, =, X consume the reference */
    const uint8_t* cig = BAMR_CIGAR(b);
    int32_t end = b->pos;
    for (int k = 0; k < b->n_cigar; k++) {
        const int op = (int)(bamr_cigar_at(cig, k) & 15u);
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) end += (int32_t)(bamr_cigar_at(cig, k) >> 4);
    }
    return end;
}

static int aux_size(int type)
{
    switch (type) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'd': return 8;
    default: return 0;
    }
}

const uint8_t* bam_aux_find(const bam_record* b, const char tag[2])
{
    const uint8_t* s = BAMR_AUX(b);
    const uint8_t* end = b->data + b->l_data;
    while (s + 4 <= end) {        /* the shortest field is 4 bytes; views of padded records end with up to 3 spare ones */
        const int hit = (s[0] == (uint8_t)tag[0] && s[1] == (uint8_t)tag[1]);
        const int type = s[2];
        const uint8_t* v = s + 2;
        s += 3;
        if (hit) return v;
        if (type == 'Z' || type == 'H') { while (s < end && *s) s++; s++; }
        else if (type == 'B') {
            if (s + 5 > end) return NULL;
            const int sz = aux_size(s[0]);
            const uint32_t cnt = (uint32_t)s[1] | ((uint32_t)s[2] << 8) | ((uint32_t)s[3] << 16) | ((uint32_t)s[4] << 24);
            s += 5 + (size_t)sz * cnt;
        } else {
            const int sz = aux_size(type);
            if (sz == 0) return NULL;
            s += sz;
        }
    }
    return NULL;
}

int32_t bam_aux_int(const uint8_t* s)
{
    if (!s) return 0;
    const int type = *s++;
    switch (type) {
    case 'c': return (int32_t)(int8_t)s[0];
    case 'C': return (int32_t)s[0];
    case 's': return (int32_t)(int16_t)(s[0] | (s[1] << 8));
    case 'S': return (int32_t)(uint16_t)(s[0] | (s[1] << 8));
    case 'i': case 'I': return (int32_t)((uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16) | ((uint32_t)s[3] << 24));
    default: return 0;
    }
}

const char* bam_aux_str(const uint8_t* s) { return (const char*)(s + 1); }

/* ------------------------------------------------------------------- BAI -- */

typedef struct {
    int32_t   n_intv;
    uint64_t* ioffset;
    uint64_t  first_chunk;      /* smallest chunk start of any real bin, 0 if none */
    uint64_t  last_chunk;       /* largest chunk end of any real bin */
} bai_ref;

struct bai_index {
    int32_t  n_ref;
    bai_ref* ref;
};

static int frd(FILE* fp, void* p, size_t n) { return fread(p, 1, n, fp) == n ? 0 : -1; }

bai_index* bai_load(const char* bam_path)
{
    size_t l = strlen(bam_path);
    char* path = malloc(l + 8);
    sprintf(path, "%s.bai", bam_path);
    FILE* fp = fopen(path, "rb");
    if (!fp && l > 4 && strcmp(bam_path + l - 4, ".bam") == 0) {   /* x.bam -> x.bai */
        strcpy(path, bam_path);
        strcpy(path + l - 3, "bai");
        fp = fopen(path, "rb");
    }
    free(path);
    if (!fp) return NULL;
    char magic[4];
    int32_t n_ref;
    if (frd(fp, magic, 4) || memcmp(magic, "BAI\1", 4) != 0 || frd(fp, &n_ref, 4)) { fclose(fp); return NULL; }
    bai_index* idx = calloc(1, sizeof *idx);
    idx->n_ref = n_ref;
    idx->ref = calloc((size_t)(n_ref > 0 ? n_ref : 1), sizeof(bai_ref));
    for (int32_t i = 0; i < n_ref; i++) {
        int32_t n_bin;
        if (frd(fp, &n_bin, 4)) goto fail;
        uint64_t first = 0, last = 0;
        for (int32_t j = 0; j < n_bin; j++) {
            uint32_t bin; int32_t n_chunk;
            if (frd(fp, &bin, 4) || frd(fp, &n_chunk, 4)) goto fail;
            for (int32_t c = 0; c < n_chunk; c++) {
                uint64_t uv[2];
                if (frd(fp, uv, 16)) goto fail;
                if (bin != 37450 && (first == 0 || uv[0] < first)) first = uv[0];
                if (bin != 37450 && uv[1] > last) last = uv[1];
            }
        }
        idx->ref[i].first_chunk = first; idx->ref[i].last_chunk = last;
        if (frd(fp, &idx->ref[i].n_intv, 4)) goto fail;
        idx->ref[i].ioffset = calloc((size_t)(idx->ref[i].n_intv > 0 ? idx->ref[i].n_intv : 1), 8);
        if (idx->ref[i].n_intv > 0 && frd(fp, idx->ref[i].ioffset, 8 * (size_t)idx->ref[i].n_intv)) goto fail;
    }
    fclose(fp);
    return idx;
fail:
    fclose(fp);
    bai_free(idx);
    return NULL;
}

/* compressed bytes of the file that hold contig tid's records (0: none): what reading the contig costs */
int64_t bai_contig_bytes(const bai_index* idx, int32_t tid)
{
    if (!idx || tid < 0 || tid >= idx->n_ref || idx->ref[tid].first_chunk == 0) return 0;
    const int64_t a = (int64_t)(idx->ref[tid].first_chunk >> 16), b = (int64_t)(idx->ref[tid].last_chunk >> 16);
    return b > a ? b - a : 1;
}

void bai_free(bai_index* idx)
{
    if (!idx) return;
    for (int32_t i = 0; i < idx->n_ref; i++) free(idx->ref[i].ioffset);
    free(idx->ref); free(idx);
}

int bam_region_begin(bam_region_iter* it, bgzf_reader* r, const bai_index* idx, int32_t tid, int32_t beg, int32_t end)
{
    memset(it, 0, sizeof *it);
    it->r = r; it->tid = tid; it->beg = beg < 0 ? 0 : beg; it->end = end;
    /* an empty interval meets no bin (reg2bins returns 0 for beg >= end, bam_index.c:559): bam_fetch(beg == end) delivers nothing */
    if (!idx || tid < 0 || tid >= idx->n_ref || end <= it->beg) { it->done = 1; return 0; }
    const bai_ref* br = &idx->ref[tid];
    if (br->first_chunk == 0) { it->done = 1; return 0; }        /* no record on this contig */
    /* smallest offset of any record overlapping the 16 kb window of beg (bam_index.c:605-615);
     * every record that overlaps [beg,end) starts at or after it, the file is coordinate sorted */
    uint64_t min_off = 0;
    if (br->n_intv > 0) {
        int32_t w = it->beg >> 14;
        min_off = (w >= br->n_intv) ? br->ioffset[br->n_intv - 1] : br->ioffset[w];
        if (min_off == 0) {
            int32_t n = w > br->n_intv ? br->n_intv : w;
            for (int32_t i = n - 1; i >= 0; i--) if (br->ioffset[i] != 0) { min_off = br->ioffset[i]; break; }
        }
    }
    if (min_off < br->first_chunk) min_off = br->first_chunk;
    if (bgzf_seek(r, (int64_t)min_off) != 0) return -1;
    return 0;
}

int bam_piece_begin(bam_region_iter* it, bgzf_reader* r, const bai_index* idx, int32_t tid, int32_t beg, int32_t end)
{
    const int rc = bam_region_begin(it, r, idx, tid, beg, end);
    it->by_start = 1;
    return rc;
}

int bai_split_points(const bai_index* idx, int32_t tid, int32_t length, int64_t target_bytes, int32_t* out, int cap)
{
    if (!idx || tid < 0 || tid >= idx->n_ref || target_bytes <= 0) return 0;
    const bai_ref* br = &idx->ref[tid];
    if (br->first_chunk == 0 || br->n_intv < 2) return 0;
    const int64_t first = (int64_t)(br->first_chunk >> 16);
    int n = 0;
    int64_t next = target_bytes;
    for (int32_t w = 1; w < br->n_intv && n < cap; w++) {
        if (br->ioffset[w] == 0) continue;
        const int64_t at = (int64_t)(br->ioffset[w] >> 16) - first;
        const int64_t pos = (int64_t)w << 14;
        if (pos >= length) break;
        if (at >= next) { out[n++] = (int32_t)pos; next = at + target_bytes; }
    }
    return n;
}

int bam_region_next(bam_region_iter* it, bam_record* b)
{
    while (!it->done) {
        int rc = bam_read_record(it->r, b);
        if (rc <= 0) { it->done = 1; return rc; }
        if (b->tid != it->tid || b->pos >= it->end) { it->done = 1; return 0; }   /* bam_index.c:704-707 */
        const uint32_t rbeg = (uint32_t)b->pos;
        const uint32_t rend = b->n_cigar ? (uint32_t)bam_record_end(b) : (uint32_t)b->pos + 1u;   /* bam_index.c:571-576 */
        if (it->by_start ? (rbeg >= (uint32_t)it->beg && rbeg < (uint32_t)it->end) : (rend > (uint32_t)it->beg && rbeg < (uint32_t)it->end)) return 1;
    }
    return 0;
}

static void core_view(const uint8_t* c, bam_record* b)
{
#define U32(o) ((uint32_t)c[o] | ((uint32_t)c[(o) + 1] << 8) | ((uint32_t)c[(o) + 2] << 16) | ((uint32_t)c[(o) + 3] << 24))
    b->tid = (int32_t)U32(0); b->pos = (int32_t)U32(4);
    b->l_qname = c[8]; b->mapq = c[9]; b->bin = (uint16_t)(c[10] | (c[11] << 8));
    b->n_cigar = (uint16_t)(c[12] | (c[13] << 8)); b->flag = (uint16_t)(c[14] | (c[15] << 8));
    b->l_seq = (int32_t)U32(16); b->mtid = (int32_t)U32(20); b->mpos = (int32_t)U32(24); b->isize = (int32_t)U32(28);
#undef U32
    b->no_qual = b->bin == BAM_BIN_NO_QUAL;
}

void bam_record_view(const uint8_t* rec, int32_t len, bam_record* view)
{
    core_view(rec, view);
    view->data = (uint8_t*)rec + 32;
    view->l_data = len - 32;
    view->m_data = 0;
}

/* A record that lies whole in the inflated block at hand (all but the one or two that straddle a block's end) is taken
 * from the block's buffer in place: no call per field.  Returns the delivered size, 0 when the slow path has to take it,
 * -2 when it does not fit cap (nothing consumed). */
static inline int32_t record_from_block(bam_region_iter* it, uint8_t* dst, int64_t cap)
{
    bgzf_reader* r = it->r;
    const int32_t avail = r->ulen - r->upos;
    if (avail < 36) return 0;
    const uint8_t* p = r->ubuf + r->upos;
    const int32_t bs = (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
    if (bs < 32 || bs > avail - 4) return 0;
    if ((int64_t)bs > cap) return -2;
    p += 4;
    int32_t out = bs;
    if (it->drop_qual) {
        const int32_t l_qname = p[8], n_cigar = p[12] | (p[13] << 8);
        const int32_t l_seq = (int32_t)((uint32_t)p[16] | ((uint32_t)p[17] << 8) | ((uint32_t)p[18] << 16) | ((uint32_t)p[19] << 24));
        const int64_t head = 32 + (int64_t)l_qname + 4 * (int64_t)n_cigar;
        const int64_t packed = ((int64_t)l_seq + 1) >> 1;
        int strip = l_seq >= 0 && head + packed + l_seq <= bs;
        if (strip) {
            int64_t q = 0;
            const uint8_t* cig = p + 32 + l_qname;
            for (int32_t k = 0; k < n_cigar; k++) {
                const uint32_t cw = bamr_cigar_at(cig, k), op = cw & 15u;
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += cw >> 4;
            }
            strip = q <= l_seq;             /* a CIGAR that reaches past l_seq has the bytes behind the bases looked at (new_readaln) */
        }
        if (strip) {
            memcpy(dst, p, (size_t)(head + packed));
            memcpy(dst + head + packed, p + head + packed + l_seq, (size_t)(bs - head - packed - l_seq));
            dst[10] = 0xFF; dst[11] = 0xFF;
            out = bs - l_seq;
        } else {
            memcpy(dst, p, (size_t)bs);
            if (dst[10] == 0xFF && dst[11] == 0xFF) dst[10] = 0xFE;
        }
    } else {
        memcpy(dst, p, (size_t)bs);
        if (dst[10] == 0xFF && dst[11] == 0xFF) dst[10] = 0xFE;        /* not a bin of any record: it must not read as the marker */
    }
    r->upos += 4 + bs;
    return out;
}

int bam_region_next_raw(bam_region_iter* it, uint8_t* dst, int64_t cap, int32_t* len_out, bam_record* view)
{
    while (!it->done) {
        if (it->pending_size == 0) {
            const int32_t fast = record_from_block(it, dst, cap);
            if (fast == -2) return -2;
            if (fast > 0) {
                bam_record_view(dst, fast, view);
                if (view->tid != it->tid || view->pos >= it->end) { it->done = 1; return 0; }
                if (32 + (int64_t)view->l_qname + 4 * (int64_t)view->n_cigar > fast) { it->done = 1; return -1; }
                const uint32_t rbeg = (uint32_t)view->pos;
                if (it->by_start) { if (rbeg >= (uint32_t)it->beg && rbeg < (uint32_t)it->end) { *len_out = fast; return 1; } continue; }
                const uint32_t rend = view->n_cigar ? (uint32_t)bam_record_end(view) : (uint32_t)view->pos + 1u;
                if (rend > (uint32_t)it->beg && rbeg < (uint32_t)it->end) { *len_out = fast; return 1; }
                continue;
            }
        }
        if (it->pending_size == 0) {
            uint8_t c[4];
            const int64_t got = bgzf_read(it->r, c, 4);
            if (got == 0) { it->done = 1; return 0; }
            if (got != 4) { it->done = 1; return -1; }
            const int32_t bs = (int32_t)((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
            if (bs < 32) { it->done = 1; return -1; }
            it->pending_size = bs;
        }
        if ((int64_t)it->pending_size > cap) return -2;
        int32_t bs = it->pending_size;
        it->pending_size = 0;
        if (!it->drop_qual) {
            if (bgzf_read(it->r, dst, bs) != bs) { it->done = 1; return -1; }
            if (dst[10] == 0xFF && dst[11] == 0xFF) dst[10] = 0xFE;        /* not a bin of any record: it must not read as the marker */
        } else {
            /* core, name and CIGAR first: they say where the qualities lie and whether the CIGAR stays inside l_seq */
            if (bgzf_read(it->r, dst, 32) != 32) { it->done = 1; return -1; }
            const int32_t l_qname = dst[8], n_cigar = dst[12] | (dst[13] << 8);
            const int32_t l_seq = (int32_t)((uint32_t)dst[16] | ((uint32_t)dst[17] << 8) | ((uint32_t)dst[18] << 16) | ((uint32_t)dst[19] << 24));
            const int64_t head = 32 + (int64_t)l_qname + 4 * (int64_t)n_cigar;
            if (l_seq < 0 || head + (((int64_t)l_seq + 1) >> 1) + l_seq > bs) {
                /* a record that is not what it says: as it is (the triage calls it malformed) */
                if (bgzf_read(it->r, dst + 32, bs - 32) != bs - 32) { it->done = 1; return -1; }
                if (dst[10] == 0xFF && dst[11] == 0xFF) dst[10] = 0xFE;
            } else {
                if (bgzf_read(it->r, dst + 32, head - 32) != head - 32) { it->done = 1; return -1; }
                int64_t q = 0;
                for (int32_t k = 0; k < n_cigar; k++) {
                    const uint32_t cw = bamr_cigar_at(dst + 32 + l_qname, k), op = cw & 15u;
                    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += cw >> 4;
                }
                const int64_t packed = ((int64_t)l_seq + 1) >> 1;
                if (q > l_seq) {
                    if (bgzf_read(it->r, dst + head, bs - head) != bs - head) { it->done = 1; return -1; }
                    if (dst[10] == 0xFF && dst[11] == 0xFF) dst[10] = 0xFE;
                } else {
                    if (bgzf_read(it->r, dst + head, packed) != packed) { it->done = 1; return -1; }
                    uint8_t skip[512];
                    for (int64_t left = l_seq; left > 0;) { const int64_t n = left < (int64_t)sizeof skip ? left : (int64_t)sizeof skip; if (bgzf_read(it->r, skip, n) != n) { it->done = 1; return -1; } left -= n; }
                    const int64_t aux = bs - head - packed - l_seq;
                    if (bgzf_read(it->r, dst + head + packed, aux) != aux) { it->done = 1; return -1; }
                    dst[10] = 0xFF; dst[11] = 0xFF;
                    bs -= l_seq;
                }
            }
        }
        bam_record_view(dst, bs, view);
        if (view->tid != it->tid || view->pos >= it->end) { it->done = 1; return 0; }    /* bam_index.c:704-707 */
        if (32 + (int64_t)view->l_qname + 4 * (int64_t)view->n_cigar > bs) { it->done = 1; return -1; }
        const uint32_t rbeg = (uint32_t)view->pos;
        const uint32_t rend = view->n_cigar ? (uint32_t)bam_record_end(view) : (uint32_t)view->pos + 1u;
        if (it->by_start ? (rbeg >= (uint32_t)it->beg && rbeg < (uint32_t)it->end) : (rend > (uint32_t)it->beg && rbeg < (uint32_t)it->end)) { *len_out = bs; return 1; }
    }
    return 0;
}

int bam_parse_region_str(const bam_header* h, const char* str, int* tid, int* beg, int* end)
{
    /* bam_parse_region (bam_aux.c:107-161): spaces dropped, last ':' splits the name, commas
     * ignored, 1-based begin -> 0-based, missing end = 2^29 */
    *tid = *beg = *end = -1;
    size_t l = strlen(str);
    char* s = malloc(l + 1);
    size_t k = 0;
    for (size_t i = 0; i < l; i++) if (!isspace((unsigned char)str[i])) s[k++] = str[i];
    s[k] = 0; l = k;
    long name_end = (long)l;
    for (long i = (long)l - 1; i >= 0; i--) if (s[i] == ':') { name_end = i; break; }
    int found = -1;
    if (name_end < (long)l) {
        int n_hyphen = 0; long i;
        for (i = name_end + 1; i < (long)l; i++) {
            if (s[i] == '-') n_hyphen++;
            else if (!isdigit((unsigned char)s[i]) && s[i] != ',') break;
        }
        if (i < (long)l || n_hyphen > 1) name_end = (long)l;
        char save = s[name_end]; s[name_end] = 0;
        for (int32_t t = 0; t < h->n_targets; t++) if (strcmp(h->target_name[t], s) == 0) { found = t; break; }
        if (found < 0) {
            for (int32_t t = 0; t < h->n_targets; t++) if (strcmp(h->target_name[t], str) == 0) { found = t; break; }
            if (found < 0) { free(s); return -1; }
            s[name_end] = save; name_end = (long)l;
        }
    } else {
        for (int32_t t = 0; t < h->n_targets; t++) if (strcmp(h->target_name[t], str) == 0) { found = t; break; }
    }
    if (found < 0) { free(s); return -1; }
    *tid = found;
    if (name_end < (long)l) {
        long i, kk;
        for (i = kk = name_end + 1; i < (long)l; i++) if (s[i] != ',') s[kk++] = s[i];
        s[kk] = 0;
        *beg = atoi(s + name_end + 1);
        for (i = name_end + 1; i != kk; i++) if (s[i] == '-') break;
        *end = i < kk ? atoi(s + i + 1) : 1 << 29;
        if (*beg > 0) --*beg;
    } else { *beg = 0; *end = 1 << 29; }
    free(s);
    return *beg <= *end ? 0 : -1;
}

/* ----------------------------------------------------------------- FASTA -- */

static int fasta_keep(int ch)
{
    /* bases[] of src/sequences.c:6-20: A B C D G H K M N R S T V W Y in either case */
    switch (ch) {
    case 'A': case 'B': case 'C': case 'D': case 'G': case 'H': case 'K': case 'M': case 'N': case 'R':
    case 'S': case 'T': case 'V': case 'W': case 'Y':
    case 'a': case 'b': case 'c': case 'd': case 'g': case 'h': case 'k': case 'm': case 'n': case 'r':
    case 's': case 't': case 'v': case 'w': case 'y':
        return 1;
    default:
        return 0;
    }
}

/* the serial reader: block by block through stdio */
static int fasta_load_serial(const char* path, int32_t n_expected, char*** seqs_out, int64_t** lens_out, int only_index)
{
    /* Same grammar as the reference's reader (src/sequences.c:63-120): blanks, then '>' header lines, every byte
     * up to the next '>' is sequence data filtered by fasta_keep and upper-cased -- read in blocks instead of
     * one fgetc per base (a 3 Gb reference is 3 G calls). */
    FILE* fp = fopen(path, "rb");
    if (!fp) return -1;
    char** seqs = calloc((size_t)(n_expected > 0 ? n_expected : 1), sizeof(char*));
    int64_t* lens = calloc((size_t)(n_expected > 0 ? n_expected : 1), sizeof(int64_t));
    uint8_t tab[256];
    for (int c = 0; c < 256; c++) tab[c] = fasta_keep(c) ? (uint8_t)toupper(c) : 0;
    const size_t BLK = 1 << 20;
    uint8_t* blk = malloc(BLK);
    size_t bn = fread(blk, 1, BLK, fp), bi = 0;
#define NEXTCH() (bi < bn ? (int)blk[bi++] : ((bn = fread(blk, 1, BLK, fp)), (bi = 0), (bn == 0 ? EOF : (int)blk[bi++])))
    int32_t indx = 0;
    int ch = NEXTCH();
    while (ch == ' ' || ch == '\t') ch = NEXTCH();
    while (ch == '>') {
        do { ch = NEXTCH(); } while (ch != '\n' && ch != EOF);
        size_t cap = 1 << 16, len = 0;
        const int keep = (only_index < 0 || only_index == indx) && indx < n_expected;
        char* buf = keep ? malloc(cap) : NULL;
        ch = EOF;
        for (;;) {
            if (bi == bn) { bn = fread(blk, 1, BLK, fp); bi = 0; if (bn == 0) break; }
            if (keep && len + (bn - bi) + 2 > cap) { while (len + (bn - bi) + 2 > cap) cap *= 2; buf = realloc(buf, cap); }
            const uint8_t* p = blk + bi; const uint8_t* e = blk + bn;
            if (keep) { for (; p < e && *p != '>'; p++) { const uint8_t u = tab[*p]; buf[len] = (char)u; len += u != 0; } }
            else { const uint8_t* q = memchr(p, '>', (size_t)(e - p)); p = q ? q : e; }
            bi = (size_t)(p - blk);
            if (p < e) { bi++; ch = '>'; break; }
        }
        if (keep) { buf[len] = 0; seqs[indx] = buf; lens[indx] = (int64_t)len; }
        indx++;
    }
#undef NEXTCH
    free(blk);
    fclose(fp);
    *seqs_out = seqs; *lens_out = lens;
    return indx;        /* caller checks indx == n_targets (forceassert, src/shared.c:77) */
}

/* The same grammar over a memory map of the file, in parallel: a 3 Gb reference is the longest serial stretch of a run otherwise.
 * The contigs' data ranges are found first (every '>' that does not lie inside a header line begins a header that ends at the next
 * newline); then each range is cut into slices, every slice's kept bytes are counted and, once the counts give the offsets,
 * written in place by the same threads. */
typedef struct { const uint8_t* base; size_t lo, hi; const uint8_t* tab; size_t kept; char* dst; int write; } fa_slice;
static void* fa_slice_run(void* arg)
{
    fa_slice* s = arg;
    const uint8_t* p = s->base + s->lo; const uint8_t* e = s->base + s->hi;
    if (!s->write) { size_t n = 0; for (; p < e; p++) n += s->tab[*p] != 0; s->kept = n; }
    else { char* d = s->dst; for (; p < e; p++) { const uint8_t u = s->tab[*p]; *d = (char)u; d += u != 0; } }
    return NULL;
}
typedef struct { const uint8_t* base; size_t lo, hi; size_t* pos; size_t n; } fa_scan;
static void* fa_scan_run(void* arg)
{
    fa_scan* s = arg;
    size_t cap = 0;
    for (const uint8_t* p = s->base + s->lo; p < s->base + s->hi;) {
        const uint8_t* q = memchr(p, '>', (size_t)(s->base + s->hi - p));
        if (!q) break;
        if (s->n == cap) { cap = cap ? cap * 2 : 64; s->pos = realloc(s->pos, sizeof(size_t) * cap); }
        s->pos[s->n++] = (size_t)(q - s->base);
        p = q + 1;
    }
    return NULL;
}
typedef struct { fa_slice* v; int lo, hi; } fa_batch;
static void* fa_batch_run(void* arg) { fa_batch* b = arg; for (int i = b->lo; i < b->hi; i++) fa_slice_run(&b->v[i]); return NULL; }

int fasta_load(const char* path, int32_t n_expected, char*** seqs_out, int64_t** lens_out, int only_index)
{
    const char* e = getenv("INDELMINER_FASTA_THREADS");
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    {   /* the cores this process may use (a container's CPU quota) */
        FILE* cf = fopen("/sys/fs/cgroup/cpu.max", "r");
        long quota = 0, period = 0;
        if (cf) { if (fscanf(cf, "%ld %ld", &quota, &period) == 2 && quota > 0 && period > 0 && quota / period < ncpu) ncpu = quota / period; fclose(cf); }
    }
    int nt = e ? atoi(e) : (int)(ncpu < 2 ? 2 : ncpu > 16 ? 16 : ncpu);
    int fd = nt > 1 ? open(path, O_RDONLY) : -1;
    struct stat sb;
    const char* mn = getenv("INDELMINER_FASTA_PARALLEL_FROM");     /* bytes; smaller files take the serial reader (tests set 0) */
    if (fd < 0 || fstat(fd, &sb) != 0 || sb.st_size < (mn ? atoll(mn) : (8 << 20)) || sb.st_size == 0) { if (fd >= 0) close(fd); return fasta_load_serial(path, n_expected, seqs_out, lens_out, only_index); }
    const size_t size = (size_t)sb.st_size;
    const uint8_t* m = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return fasta_load_serial(path, n_expected, seqs_out, lens_out, only_index);
    if (nt > 32) nt = 32;
    uint8_t tab[256];
    for (int c = 0; c < 256; c++) tab[c] = fasta_keep(c) ? (uint8_t)toupper(c) : 0;
    /* data ranges, in file order.  Every '>' of the file is found first, the file cut into one stretch per thread (a scan of
     * gigabytes by one thread was a third of the load); the walk below then hops from header to header as before: a '>' inside
     * a header line is part of that line. */
    size_t* gt = NULL; size_t n_gt = 0;
    {
        fa_scan sc[32]; pthread_t th[32]; int started = 0;
        for (int t = 0; t < nt; t++) {
            sc[t].base = m; sc[t].lo = size * (size_t)t / (size_t)nt; sc[t].hi = size * (size_t)(t + 1) / (size_t)nt; sc[t].pos = NULL; sc[t].n = 0;
            if (pthread_create(&th[t], NULL, fa_scan_run, &sc[t]) != 0) { fa_scan_run(&sc[t]); th[t] = 0; } else started |= 1 << t;
        }
        for (int t = 0; t < nt; t++) if (started & (1 << t)) pthread_join(th[t], NULL);
        for (int t = 0; t < nt; t++) n_gt += sc[t].n;
        gt = malloc(sizeof(size_t) * (n_gt + 1));
        n_gt = 0;
        for (int t = 0; t < nt; t++) { if (sc[t].n) memcpy(gt + n_gt, sc[t].pos, sizeof(size_t) * sc[t].n); n_gt += sc[t].n; free(sc[t].pos); }
    }
    size_t cap_r = 64, n_r = 0;
    size_t (*rng)[2] = malloc(sizeof(size_t[2]) * cap_r);
    size_t at = 0, gi = 0;
    while (at < size && (m[at] == ' ' || m[at] == '\t')) at++;
    while (at < size && m[at] == '>') {
        const uint8_t* nl = memchr(m + at, '\n', size - at);
        const size_t d0 = nl ? (size_t)(nl - m) + 1 : size;
        while (gi < n_gt && gt[gi] < d0) gi++;                     /* the first '>' at or behind the data's start */
        const size_t d1 = gi < n_gt ? gt[gi] : size;
        if (n_r == cap_r) { cap_r *= 2; rng = realloc(rng, sizeof(size_t[2]) * cap_r); }
        rng[n_r][0] = d0; rng[n_r][1] = d1; n_r++;
        at = d1;
    }
    free(gt);
    char** seqs = calloc((size_t)(n_expected > 0 ? n_expected : 1), sizeof(char*));
    int64_t* lens = calloc((size_t)(n_expected > 0 ? n_expected : 1), sizeof(int64_t));
    /* slices of the ranges that are kept */
    const size_t SL = 4 << 20;
    size_t n_sl = 0;
    for (size_t r = 0; r < n_r; r++) if ((only_index < 0 || only_index == (int)r) && (int32_t)r < n_expected) n_sl += (rng[r][1] - rng[r][0]) / SL + 1;
    fa_slice* sl = calloc(n_sl ? n_sl : 1, sizeof(fa_slice));
    size_t* first_of = calloc(n_r + 1, sizeof(size_t));
    size_t k = 0;
    for (size_t r = 0; r < n_r; r++) {
        first_of[r] = k;
        if (!((only_index < 0 || only_index == (int)r) && (int32_t)r < n_expected)) continue;
        for (size_t lo = rng[r][0]; lo < rng[r][1] || lo == rng[r][0]; lo += SL) {
            sl[k].base = m; sl[k].lo = lo; sl[k].hi = lo + SL < rng[r][1] ? lo + SL : rng[r][1]; sl[k].tab = tab; k++;
            if (rng[r][1] == rng[r][0]) break;
        }
    }
    first_of[n_r] = k;
    n_sl = k;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            for (size_t r = 0; r < n_r; r++) {
                if (first_of[r] == first_of[r + 1]) continue;
                size_t tot = 0;
                for (size_t i = first_of[r]; i < first_of[r + 1]; i++) tot += sl[i].kept;
                char* buf = malloc(tot + 2);
                if (buf && tot >= ((size_t)4 << 20) && !(getenv("INDELMINER_THP") && getenv("INDELMINER_THP")[0] == '0')) {
                    /* a genome is gigabytes of first touches: 2 MB pages where the kernel hands them out on request */
                    const uintptr_t a = (uintptr_t)buf & ~(uintptr_t)4095, z = ((uintptr_t)buf + tot) & ~(uintptr_t)4095;
                    if (z > a) (void)madvise((void*)a, (size_t)(z - a), MADV_HUGEPAGE);
                }
                size_t off = 0;
                for (size_t i = first_of[r]; i < first_of[r + 1]; i++) { sl[i].dst = buf + off; off += sl[i].kept; sl[i].write = 1; }
                buf[tot] = 0;
                seqs[r] = buf; lens[r] = (int64_t)tot;
            }
        }
        pthread_t th[32]; fa_batch bt[32];
        int started = 0;
        for (int t = 0; t < nt; t++) {
            bt[t].v = sl; bt[t].lo = (int)(n_sl * (size_t)t / (size_t)nt); bt[t].hi = (int)(n_sl * (size_t)(t + 1) / (size_t)nt);
            if (pthread_create(&th[t], NULL, fa_batch_run, &bt[t]) != 0) { fa_batch_run(&bt[t]); th[t] = 0; } else started |= 1 << t;
        }
        for (int t = 0; t < nt; t++) if (started & (1 << t)) pthread_join(th[t], NULL);
    }
    free(sl); free(first_of); free(rng);
    munmap((void*)m, size);
    *seqs_out = seqs; *lens_out = lens;
    return (int)n_r;    /* caller checks it against n_targets (forceassert, src/shared.c:77) */
}

/* ------------------------------------------------------------- getline -- */

long im_getline(char** lineptr, size_t* cap, FILE* fp)
{
    int ch = EOF;
    size_t size = 0;
    while ((ch = fgetc(fp)) != EOF) {
        if (size + 2 > *cap) { *cap = size + (size >> 5) + 16; *lineptr = realloc(*lineptr, *cap); }
        (*lineptr)[size++] = (char)ch;
        if (ch == '\n') break;
    }
    if (size != 0) (*lineptr)[size] = 0;
    if (size == 0 || ch == EOF) return -1;       /* a final line without '\n' is dropped (src/files.c:50-52) */
    return (long)size;
}
