/*
 * im_shim.c -- TEST-ONLY stand-in for libindelminer_amd.so.
 *
 * Implements the handful of C-ABI entry points the host driver calls with the CPU oracle
 * (oracle/im_oracle.c), so that the HOST LOGIC of the driver (BAM/FASTA readers, fetch_func
 * dispatch, flush replay, merge, VCF writer) can be tested without a GPU (-m "not gpu").
 * It lives under tests/ and is linked only into tests/shim/indelminer_shim; the product
 * binary links the HIP library and has no CPU path.
 */
#define _POSIX_C_SOURCE 200809L
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "indelminer_amd.h"
#include "../../oracle/im_oracle.h"

struct im_ctx { int n; char** seqs; int32_t* lens; char err[256]; int32_t* depth; int64_t depth_len; };
static char g_err[256] = "";

int im_abi_version(void) { return IM_ABI_VERSION; }
const char* im_last_error(const im_ctx* c) { return c ? c->err : g_err; }
int im_ctx_create(int device, im_ctx** out) { (void)device; *out = calloc(1, sizeof(im_ctx)); return IM_OK; }
void im_ctx_destroy(im_ctx* c) { if (!c) return; for (int i = 0; i < c->n; i++) free(c->seqs[i]); free(c->seqs); free(c->lens); free(c); }

int im_set_reference(im_ctx* c, int32_t n, const char* const* seqs, const int64_t* lens)
{
    c->n = n; c->seqs = calloc((size_t)n, sizeof(char*)); c->lens = calloc((size_t)n, sizeof(int32_t));
    for (int i = 0; i < n; i++) { c->seqs[i] = malloc((size_t)lens[i] + 1); memcpy(c->seqs[i], seqs[i], (size_t)lens[i]); c->seqs[i][lens[i]] = 0; c->lens[i] = (int32_t)lens[i]; }
    return IM_OK;
}

int im_dev_compact_results(im_ctx* c, const im_read_result* res, int32_t n, const int32_t* n_dev,
                           int32_t* status, int32_t* slot, im_read_result* compact, int32_t* count, void* stream)
{
    (void)c; (void)stream;
    if (n_dev && *n_dev < n) n = *n_dev;
    int32_t k = 0;
    /* the device leaves the packed records in no particular order: here from the back, so that nothing grows to rely on one */
    for (int32_t i = n - 1; i >= 0; i--) {
        status[i] = res[i].status;
        if (res[i].status == IM_ST_EVIDENCE && res[i].n_ev > 0) { compact[k] = res[i]; slot[i] = k++; } else slot[i] = -1;
    }
    *count = k;
    return IM_OK;
}

int im_expect_read_length(im_ctx* c, int32_t max_len) { (void)c; (void)max_len; return IM_OK; }   /* the oracle has one path for every length */

int im_realign_batch(im_ctx* c, const im_params* p, const im_read_batch* b, im_read_result* out)
{
    imo_params P = { p->klength, p->numgaps, p->maxdelsize, p->ethreshold };
    imo_result* r = malloc(sizeof *r);
    int worst = IM_OK;
    for (int32_t i = 0; i < b->n; i++) {
        const int64_t len = b->base_off[i + 1] - b->base_off[i];
        char* read = malloc((size_t)len + 1);
        memcpy(read, b->bases + b->base_off[i], (size_t)len); read[len] = 0;
        const int t = b->tid[i];
        int st = imo_realign(&P, c->seqs[t], c->lens[t], b->anchor[i], b->range_max[i], read, (int32_t)len, r);
        free(read);
        im_read_result* o = &out[i];
        memset(o, 0, sizeof *o);
        o->status = st == IMO_OK ? IM_ST_EVIDENCE : st == IMO_NONE ? IM_ST_NONE : st == IMO_ABORT ? IM_ST_ABORT : IM_ST_OVERFLOW;
        if (st == IMO_ABORT) { worst = IM_E_ABORT; snprintf(c->err, sizeof c->err, "read %d: the reference would abort", i); }
        if (st != IMO_OK) continue;
        if (r->n_ops > IM_MAX_OPS || r->n_ev > IM_MAX_EV) { o->status = IM_ST_OVERFLOW; worst = IM_E_OVERFLOW; continue; }
        o->ref_start = r->ref_start; o->n_ops = r->n_ops; o->n_ev = r->n_ev; o->n_band = r->n_band;
        memcpy(o->ops, r->ops, sizeof(uint32_t) * (size_t)r->n_ops);
        for (int k = 0; k < r->n_ev; k++) {
            o->ev[k].cls = r->ev[k].cls; o->ev[k].b1 = r->ev[k].b1; o->ev[k].b2 = r->ev[k].b2; o->ev[k].seg = r->ev[k].seg;
            o->ev[k].read_off = r->ev[k].read_off; o->ev[k].lflank = r->ev[k].lflank; o->ev[k].rflank = r->ev[k].rflank;
            o->ev[k].nd_print = r->ev[k].nd_print; o->ev[k].nd_filter = r->ev[k].nd_filter;
        }
    }
    free(r);
    return worst;
}

int im_cluster_sr(im_ctx* c, int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                  int32_t marker, int32_t tie_desc, int32_t* order, int32_t* first, int32_t* count,
                  uint8_t* used, int32_t* n_clusters)
{
    (void)c;
    *n_clusters = imo_cluster_sr(n, cls, b1, b2, marker, tie_desc, order, first, count, used);
    return IM_OK;
}

int im_depth_build(im_ctx* c, int64_t contig_len, int32_t n_seg, const int32_t* seg_start, const int32_t* seg_len)
{
    free(c->depth);
    c->depth = calloc((size_t)contig_len + 2, sizeof(int32_t));
    c->depth_len = contig_len;
    for (int32_t i = 0; i < n_seg; i++) {
        int64_t a = seg_start[i], b = (int64_t)seg_start[i] + seg_len[i];
        if (a < 0) a = 0;
        if (b > contig_len) b = contig_len;
        if (a >= b) continue;
        c->depth[a] += 1; c->depth[b] -= 1;
    }
    int32_t run = 0;
    for (int64_t p = 0; p <= contig_len; p++) { run += c->depth[p]; c->depth[p] = run; }
    return IM_OK;
}

int im_depth_query(im_ctx* c, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out)
{
    for (int32_t q = 0; q < n; q++) {
        int64_t a = beg[q], b = end[q];
        if (a < 0) a = 0;
        if (b > c->depth_len) b = c->depth_len;
        uint32_t s = 0;
        for (int64_t p = a; p < b; p++) s += (uint32_t)c->depth[p];
        sum_out[q] = s;
    }
    return IM_OK;
}

int im_support_batch(im_ctx* c, int32_t n, const uint8_t* targets, const int64_t* t_off,
                     const uint8_t* queries, const int64_t* q_off, int32_t* out)
{
    (void)c;
    if (getenv("IM_SHIM_TRACE")) fprintf(stderr, "[shim] im_support_batch: %d tasks\n", n);
    for (int32_t i = 0; i < n; i++) {
        int32_t subs, indels, aligned;
        imo_sw_indel((const char*)targets + t_off[i], (int32_t)(t_off[i + 1] - t_off[i]),
                     (const char*)queries + q_off[i], (int32_t)(q_off[i + 1] - q_off[i]), &subs, &indels, &aligned);
        out[4 * i] = subs; out[4 * i + 1] = indels; out[4 * i + 2] = aligned; out[4 * i + 3] = IM_ST_EVIDENCE;
    }
    return IM_OK;
}

/* ---- the device-pipeline entry points, CPU edition (host pointers stand in for device pointers,
 * everything runs synchronously) ------------------------------------------------------------------ */

struct im_event { int dummy; };
static int g_n_rg; static char** g_rg_names; static int32_t* g_rg_range;
static int32_t** g_gdepth;      /* per contig, length + 1 */

int im_set_insert_ranges(im_ctx* c, int32_t n, const char* const* names, const int32_t* range_max)
{
    (void)c;
    g_n_rg = n; g_rg_names = calloc((size_t)n + 1, sizeof(char*)); g_rg_range = calloc((size_t)n + 1, sizeof(int32_t));
    for (int i = 0; i < n; i++) { g_rg_names[i] = strdup(names[i]); g_rg_range[i] = range_max[i]; }
    return IM_OK;
}
int im_depth_enable(im_ctx* c)
{
    if (g_gdepth) return IM_OK;
    g_gdepth = calloc((size_t)c->n, sizeof(int32_t*));
    for (int i = 0; i < c->n; i++) g_gdepth[i] = calloc((size_t)c->lens[i] + 2, sizeof(int32_t));
    return IM_OK;
}
int im_depth_scan(im_ctx* c, int32_t tid, void* stream)
{
    (void)stream;
    int32_t run = 0;
    for (int64_t p = 0; p <= c->lens[tid]; p++) { run += g_gdepth[tid][p]; g_gdepth[tid][p] = run; }
    return IM_OK;
}
int im_depth_reset(im_ctx* c, int32_t tid, void* stream)
{
    (void)stream;
    memset(g_gdepth[tid], 0, sizeof(int32_t) * ((size_t)c->lens[tid] + 1));
    return IM_OK;
}
int im_depth_query_max_tid(im_ctx* c, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out, uint32_t* max_out)
{
    for (int32_t q = 0; q < n; q++) {
        int64_t a = beg[q], b = end[q];
        if (a < 0) a = 0;
        if (b > c->lens[tid]) b = c->lens[tid];
        uint32_t s = 0, mx = 0;
        for (int64_t p = a > 0 ? a - 1 : a; p < (b < c->lens[tid] ? b + 1 : b); p++) {
            const uint32_t v = (uint32_t)g_gdepth[tid][p];
            if (p >= a && p < b) s += v;
            if (v > mx) mx = v;
        }
        sum_out[q] = s;
        if (max_out) max_out[q] = mx;
    }
    return IM_OK;
}
int im_depth_query_tid(im_ctx* c, int32_t tid, int32_t n, const int32_t* beg, const int32_t* end, uint32_t* sum_out)
{
    for (int32_t q = 0; q < n; q++) {
        int64_t a = beg[q], b = end[q];
        if (a < 0) a = 0;
        if (b > c->lens[tid]) b = c->lens[tid];
        uint32_t s = 0;
        for (int64_t p = a; p < b; p++) s += (uint32_t)g_gdepth[tid][p];
        sum_out[q] = s;
    }
    return IM_OK;
}
int im_stream_create(im_ctx* c, void** out) { (void)c; *out = NULL; return IM_OK; }
int im_stream_destroy(im_ctx* c, void* s) { (void)c; (void)s; return IM_OK; }
int im_stream_sync(im_ctx* c, void* s) { (void)c; (void)s; return IM_OK; }
int im_host_alloc(im_ctx* c, size_t bytes, void** out) { (void)c; *out = malloc(bytes ? bytes : 1); return IM_OK; }
int im_host_free(im_ctx* c, void* p) { (void)c; free(p); return IM_OK; }
/* like hipMalloc, the block is NOT zeroed: filled with a pattern so that a caller relying on zeros fails here, on the CPU */
int im_dev_alloc(im_ctx* c, size_t bytes, void** out)
{
    (void)c;
    *out = malloc(bytes ? bytes : 1);
    if (!*out) return IM_E_HIP;
    if (bytes <= ((size_t)1 << 26)) memset(*out, 0xA5, bytes ? bytes : 1);      /* the large arrays are written before they are read; skip their fill */
    return IM_OK;
}
int im_dev_free(im_ctx* c, void* p) { (void)c; free(p); return IM_OK; }
int im_dev_upload(im_ctx* c, void* dst, const void* src, size_t bytes) { (void)c; memcpy(dst, src, bytes); return IM_OK; }
int im_dev_download(im_ctx* c, void* dst, const void* src, size_t bytes) { (void)c; memcpy(dst, src, bytes); return IM_OK; }
int im_dev_upload_async(im_ctx* c, void* dst, const void* src, size_t bytes, void* s) { (void)c; (void)s; memcpy(dst, src, bytes); return IM_OK; }
int im_dev_download_async(im_ctx* c, void* dst, const void* src, size_t bytes, void* s) { (void)c; (void)s; memcpy(dst, src, bytes); return IM_OK; }
int im_dev_copy_async(im_ctx* c, void* dst, const void* src, size_t bytes, void* s) { (void)c; (void)s; memcpy(dst, src, bytes); return IM_OK; }
int im_dev_memset(im_ctx* c, void* dst, int byte, size_t bytes, void* s) { (void)c; (void)s; memset(dst, byte, bytes); return IM_OK; }
int im_event_create(im_ctx* c, im_event** out) { (void)c; *out = calloc(1, sizeof(im_event)); return IM_OK; }
void im_event_destroy(im_event* e) { free(e); }
int im_event_record(im_event* e, void* s) { (void)e; (void)s; return IM_OK; }
int im_event_sync(im_event* e) { (void)e; return IM_OK; }
size_t im_dev_triage_scratch_bytes(int32_t n) { (void)n; return 256; }
int im_dev_triage_scratch_init(im_ctx* c, int32_t n, void* s, size_t b, void* st) { (void)c; (void)n; (void)s; (void)b; (void)st; return IM_OK; }
int im_dev_groupby_scratch_init(im_ctx* c, int32_t n, void* s, size_t b, void* st) { (void)c; (void)n; (void)s; (void)b; (void)st; return IM_OK; }
size_t im_dev_groupby_scratch_bytes(int32_t n) { (void)n; return 256; }

int im_dev_triage(im_ctx* c, const im_triage_params* tp, const im_dev_records* recs, const im_dev_cands* out,
                  void* scratch, size_t scratch_bytes, void* stream)
{
    (void)scratch; (void)scratch_bytes; (void)stream;
    int32_t* cnt = out->counters;
    if (tp->restart) for (int k = 0; k < 5; k++) cnt[k] = 0;       /* a new batch: the running counters count as zero */
    imo_triage t;
    char* bases = malloc(1 << 20);
    uint8_t* whole = NULL; size_t whole_cap = 0;
    for (int32_t i = 0; i < recs->n; i++) {
        const uint8_t* rec = recs->raw + recs->rec_off[i];
        uint32_t len = recs->rec_off[i + 1] - recs->rec_off[i];
        if (len >= 32 && rec[10] == 0xFF && rec[11] == 0xFF) {
            /* delivered without its base qualities (bin = 0xFFFF): the oracle takes records as the file has them -- put l_seq bytes back */
            const uint32_t l_qname = rec[8], n_cigar = rec[12] | (rec[13] << 8);
            const uint32_t l_seq = (uint32_t)rec[16] | ((uint32_t)rec[17] << 8) | ((uint32_t)rec[18] << 16) | ((uint32_t)rec[19] << 24);
            const uint32_t head = 32 + l_qname + 4 * n_cigar + ((l_seq + 1) >> 1);
            if ((size_t)len + l_seq + 8 > whole_cap) { whole_cap = ((size_t)len + l_seq + 8) * 2; whole = realloc(whole, whole_cap); }
            memcpy(whole, rec, head);
            memset(whole + head, 0xFF, l_seq);
            memcpy(whole + head + l_seq, rec + head, len - head);
            whole[10] = 0; whole[11] = 0;
            rec = whole; len += l_seq;
        }
        imo_triage_record(rec, len, tp->defer_ranges ? -1 : g_n_rg, (const char* const*)g_rg_names, g_rg_range, tp->qthreshold, tp->ethreshold_vcfcheck,
                          tp->maxpedelsize, &t, bases);
        int cls = t.cls;
        if (cls == 3 && t.n_ev > IM_MAX_EV) cls = IM_REC_ERR_LIMIT;
        if (out->rec_class) out->rec_class[i] = (uint8_t)cls;
        if (cls != IM_REC_SKIP) cnt[2]++;
        if (cls >= IM_REC_ERR_RG) cnt[3]++;
        if (tp->want_depth && g_gdepth) {
            /* the pileup's read filter and match segments (src/shared.c:160-176, bam_pileup.c:171-172), difference form */
#define RD32(o) ((int32_t)((uint32_t)rec[o] | ((uint32_t)rec[(o) + 1] << 8) | ((uint32_t)rec[(o) + 2] << 16) | ((uint32_t)rec[(o) + 3] << 24)))
            const int32_t tid = RD32(0);
            const int flag = rec[14] | (rec[15] << 8), ncig = rec[12] | (rec[13] << 8), lq = rec[8];
            if (len >= 32 && tid >= 0 && tid < c->n && !(flag & (0x4 | 0x100 | 0x200 | 0x400)) && 32u + (uint32_t)lq + 4u * (uint32_t)ncig <= len) {
                int64_t x = RD32(4);
                for (int k = 0; k < ncig; k++) {
                    const uint32_t w = (uint32_t)RD32(32 + lq + 4 * k);
                    const int op = (int)(w & 15u); const int64_t l = w >> 4;
                    if (op == 0 || op == 7 || op == 8) {
                        int64_t a = x < 0 ? 0 : x, b = x + l > c->lens[tid] ? c->lens[tid] : x + l;
                        if (a < b) { g_gdepth[tid][a] += 1; g_gdepth[tid][b] -= 1; }
                        x += l;
                    } else if (op == 2 || op == 3) x += l;
                }
            }
#undef RD32
        }
        if (t.cls != 2 && t.cls != 3) continue;
        const int32_t ci = cnt[0];
        const int64_t bo = cnt[1];
        const int32_t padded = (t.l_seq + 3) & ~3;
        if (ci >= out->cap_cand || bo + padded + 16 > out->cap_bases) { cnt[4]++; cnt[0]++; cnt[1] += padded; continue; }
        memset((uint8_t*)out->batch.bases + bo, 0, (size_t)padded);
        memcpy((uint8_t*)out->batch.bases + bo, bases, (size_t)t.l_seq);
        ((int64_t*)out->batch.base_off)[ci] = bo; ((int32_t*)out->batch.read_len)[ci] = t.l_seq;
        ((int32_t*)out->batch.tid)[ci] = t.tid; ((int32_t*)out->batch.anchor)[ci] = t.anchor; ((int32_t*)out->batch.range_max)[ci] = t.range_max;
        out->cand_rec[ci] = recs->rec_base + i;
        for (int k = 0; k < IM_MAX_EV; k++) {
            const int live = k < t.n_ev && t.n_ev <= IM_MAX_EV;
            if (out->consumed) out->consumed[(size_t)ci * IM_MAX_EV + k] = 0;
            out->batch.ev_cls[(size_t)ci * IM_MAX_EV + k] = live ? t.ev_cls[k] : -1;
            out->batch.ev_b1[(size_t)ci * IM_MAX_EV + k] = live ? t.ev_b1[k] : 0;
            out->batch.ev_b2[(size_t)ci * IM_MAX_EV + k] = live ? t.ev_b2[k] : 0;
        }
        cnt[0]++; cnt[1] += padded;
    }
    free(bases); free(whole);
    return IM_OK;
}

int im_dev_realign_keep(im_ctx* c, const im_params* p, const im_dev_batch* b, void* stream)
{
    (void)stream;
    im_read_batch hb;
    int64_t* off = malloc(sizeof(int64_t) * ((size_t)b->n + 1));
    uint8_t* bases = malloc(1);
    int64_t tot = 0;
    for (int32_t i = 0; i < b->n; i++) tot += b->read_len[i];
    bases = realloc(bases, (size_t)tot + 16);
    tot = 0;
    for (int32_t i = 0; i < b->n; i++) { off[i] = tot; memcpy(bases + tot, b->bases + b->base_off[i], (size_t)b->read_len[i]); tot += b->read_len[i]; }
    off[b->n] = tot;
    hb.n = b->n; hb.bases = bases; hb.base_off = off; hb.tid = b->tid; hb.anchor = b->anchor; hb.range_max = b->range_max;
    (void)im_realign_batch(c, p, &hb, b->out);      /* per-read statuses are checked by the caller */
    for (int32_t i = 0; i < b->n; i++) {
        const im_read_result* r = &b->out[i];
        if (r->status != IM_ST_EVIDENCE || r->n_ev <= 0 || !b->ev_cls) continue;
        for (int k = 0; k < IM_MAX_EV; k++) {
            const int live = k < r->n_ev;
            b->ev_cls[(size_t)i * IM_MAX_EV + k] = live ? r->ev[k].cls : -1;
            b->ev_b1[(size_t)i * IM_MAX_EV + k] = live ? r->ev[k].b1 : 0;
            b->ev_b2[(size_t)i * IM_MAX_EV + k] = live ? r->ev[k].b2 : 0;
        }
    }
    free(off); free(bases);
    return IM_OK;
}

int im_dev_flush_cut(im_ctx* c, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                     int32_t a0, int32_t a1, int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id,
                     uint64_t* cut_word, void* stream)
{
    (void)c; (void)cut_word; (void)stream;
    const int32_t na = a1 > a0 ? a1 - a0 : 0, nb = b1_end > b0 ? b1_end - b0 : 0, n = na + nb;
    if (n == 0) return IM_OK;
    int32_t *tc = malloc(4 * (size_t)n), *t1 = malloc(4 * (size_t)n), *t2 = malloc(4 * (size_t)n), *tu = malloc(4 * (size_t)n);
    for (int32_t i = 0; i < n; i++) {
        const int32_t s = i < na ? a0 + i : b0 + (i - na);
        tc[i] = cls[s]; t1[i] = b1[s]; t2[i] = b2[s]; tu[i] = consumed[s];
    }
    imo_flush_cut(n, tc, t1, t2, tu, marker, flush_id);
    for (int32_t i = 0; i < n; i++) consumed[i < na ? a0 + i : b0 + (i - na)] = tu[i];
    free(tc); free(t1); free(t2); free(tu);
    return IM_OK;
}

int im_dev_flush_cut_rec(im_ctx* c, const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         int32_t rec0, int32_t rec1, const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap,
                         int32_t b0, int32_t b1_end, int32_t marker, int32_t flush_id, uint64_t* cut_word, void* stream)
{
    const int32_t nc = *n_cand_dev < cand_cap ? *n_cand_dev : cand_cap;
    int32_t lo = 0, hi;
    while (lo < nc && cand_rec[lo] < rec0) lo++;
    hi = lo;
    while (hi < nc && cand_rec[hi] < rec1) hi++;
    return im_dev_flush_cut(c, cls, b1, b2, consumed, lo * IM_MAX_EV, hi * IM_MAX_EV, b0, b1_end, marker, flush_id, cut_word, stream);
}

int im_dev_flush_cuts(im_ctx* c, const im_flush_desc* desc, int32_t n_flushes,
                      const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                      const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count, void* stream)
{
    for (int32_t i = 0; i < pe_count; i++) consumed[pe_base + i] = 0;
    const int32_t nc = *n_cand_dev < cand_cap ? *n_cand_dev : cand_cap;
    for (int32_t f = 0; f < n_flushes; f++) {
        int32_t lo = 0, hi = 0;
        while (lo < nc && cand_rec[lo] < desc[f].rec0) lo++;
        hi = lo;
        while (hi < nc && cand_rec[hi] < desc[f].rec1) hi++;
        uint64_t dummy = ~0ull;
        im_dev_flush_cut(c, cls, b1, b2, consumed, lo * IM_MAX_EV, hi * IM_MAX_EV, pe_base + desc[f].pe0, pe_base + desc[f].pe1,
                         desc[f].marker, desc[f].id, &dummy, stream);
    }
    return IM_OK;
}

typedef struct { int32_t f, c, b1, b2, slot; } gkey;
static int cmp_gkey(const void* x, const void* y)
{
    const gkey* a = x; const gkey* b = y;
    if (a->f != b->f) return a->f < b->f ? -1 : 1;
    if (a->b1 != b->b1) return a->b1 < b->b1 ? -1 : 1;
    if (a->b2 != b->b2) return a->b2 < b->b2 ? -1 : 1;
    if (a->c != b->c) return a->c < b->c ? -1 : 1;
    return a->slot < b->slot ? -1 : a->slot > b->slot;
}
int im_dev_cluster_groupby(im_ctx* c, int32_t n_slots, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                           const int32_t* consumed, int32_t tie_desc,
                           int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                           void* scratch, size_t scratch_bytes, void* stream)
{
    (void)c; (void)scratch; (void)scratch_bytes; (void)stream;
    gkey* k = malloc(sizeof(gkey) * (size_t)(n_slots > 0 ? n_slots : 1));
    int32_t m = 0;
    for (int32_t i = 0; i < n_slots; i++)
        if (cls[i] >= 0 && cls[i] < 2 && consumed[i] > 0) { k[m].f = consumed[i]; k[m].c = cls[i]; k[m].b1 = b1[i]; k[m].b2 = b2[i]; k[m].slot = i; m++; }
    qsort(k, (size_t)m, sizeof(gkey), cmp_gkey);
    int32_t ncl = 0;
    for (int32_t i = 0; i < m;) {
        int32_t j = i;
        while (j < m && k[j].f == k[i].f && k[j].c == k[i].c && k[j].b1 == k[i].b1 && k[j].b2 == k[i].b2) j++;
        cl_key[4 * ncl] = k[i].f; cl_key[4 * ncl + 1] = k[i].c; cl_key[4 * ncl + 2] = k[i].b1; cl_key[4 * ncl + 3] = k[i].b2;
        cl_first[ncl] = i; cl_count[ncl] = j - i;
        for (int32_t t = i; t < j; t++) order[tie_desc ? (j - 1 - (t - i)) : t] = k[t].slot;
        ncl++;
        i = j;
    }
    counts[0] = ncl; counts[1] = m;
    free(k);
    return IM_OK;
}

/* The chip-wide form: flush marks without history (imo_flush_nohistory restates what the kernels compute), then the group-by. */
size_t im_dev_flushgroup_scratch_bytes(int32_t n_slots_cap, int32_t n_flushes_cap) { (void)n_slots_cap; (void)n_flushes_cap; return 256; }
int im_dev_flushgroup_scratch_init(im_ctx* c, int32_t n_slots_cap, int32_t n_flushes_cap, void* scratch, size_t scratch_bytes, void* stream)
{
    (void)c; (void)n_slots_cap; (void)n_flushes_cap; (void)scratch; (void)scratch_bytes; (void)stream;
    return IM_OK;
}
int im_dev_flush_groupby(im_ctx* c, const im_flush_desc* desc, int32_t n_flushes,
                         const int32_t* cls, const int32_t* b1, const int32_t* b2, int32_t* consumed,
                         const int32_t* cand_rec, const int32_t* n_cand_dev, int32_t cand_cap, int32_t pe_base, int32_t pe_count, int32_t tie_desc,
                         int32_t* order, int32_t* cl_key, int32_t* cl_first, int32_t* cl_count, int32_t* counts,
                         void* scratch, size_t scratch_bytes, void* stream)
{
    const int32_t nc = *n_cand_dev < cand_cap ? *n_cand_dev : cand_cap;
    const int32_t ns = nc * IM_MAX_EV, n = ns + pe_count;
    int32_t *mk = malloc(4 * (size_t)(n_flushes + 1)), *id = malloc(4 * (size_t)(n_flushes + 1)), *last = malloc(4 * (size_t)(n_flushes + 1));
    for (int32_t f = 0; f < n_flushes; f++) {
        mk[f] = desc[f].marker; id[f] = desc[f].id;
        last[f] = desc[f].last < f ? f : (desc[f].last >= n_flushes ? n_flushes - 1 : desc[f].last);
    }
    int32_t *tc = malloc(4 * (size_t)(n + 1)), *t1 = malloc(4 * (size_t)(n + 1)), *t2 = malloc(4 * (size_t)(n + 1)), *ta = malloc(4 * (size_t)(n + 1)), *tu = malloc(4 * (size_t)(n + 1));
    int32_t f = 0;
    for (int32_t k = 0; k < nc; k++) {
        while (f < n_flushes && desc[f].rec1 <= cand_rec[k]) f++;       /* cand_rec ascends */
        for (int j = 0; j < IM_MAX_EV; j++) { const int32_t s = k * IM_MAX_EV + j; tc[s] = cls[s]; t1[s] = b1[s]; t2[s] = b2[s]; ta[s] = f; }
    }
    f = 0;
    for (int32_t i = 0; i < pe_count; i++) {
        while (f < n_flushes && desc[f].pe1 <= i) f++;
        const int32_t s = pe_base + i; tc[ns + i] = cls[s]; t1[ns + i] = b1[s]; t2[ns + i] = b2[s]; ta[ns + i] = f;
    }
    const int rc = imo_flush_nohistory(n_flushes, mk, id, last, n, tc, t1, t2, ta, tu);
    if (rc == 0) {
        for (int32_t s = 0; s < ns; s++) consumed[s] = tu[s];
        for (int32_t i = 0; i < pe_count; i++) consumed[pe_base + i] = tu[ns + i];
    }
    free(mk); free(id); free(last); free(tc); free(t1); free(t2); free(ta); free(tu);
    if (rc != 0) { snprintf(c->err, sizeof c->err, "im_dev_flush_groupby: the markers of a contig decrease"); return IM_E_ARG; }
    return im_dev_cluster_groupby(c, ns, cls, b1, b2, consumed, tie_desc, order, cl_key, cl_first, cl_count, counts, scratch, scratch_bytes, stream);
}

/* ---- the collective, CPU edition: an all-gather through files in a directory named by the unique id ---- */
#include <time.h>
#include <unistd.h>
struct im_comm { int rank, world, seq; char dir[200]; };
static char g_comm_err[1024] = "";
const char* im_comm_last_error(void) { return g_comm_err; }
void* im_ctx_stream(im_ctx* c) { (void)c; return NULL; }
int im_comm_unique_id(void* id_bytes)
{
    memset(id_bytes, 0, IM_COMM_ID_BYTES);
    snprintf((char*)id_bytes, IM_COMM_ID_BYTES, "/tmp/im_shim_comm_%ld_%d", (long)time(NULL), (int)getpid());
    return IM_OK;
}
int im_comm_init(im_ctx* c, const void* id_bytes, int rank, int world, im_comm** out)
{
    (void)c;
    im_comm* m = calloc(1, sizeof *m);
    m->rank = rank; m->world = world;
    snprintf(m->dir, sizeof m->dir, "%s", (const char*)id_bytes);
    char cmd[300];
    snprintf(cmd, sizeof cmd, "mkdir -p '%s'", m->dir);
    if (system(cmd) != 0) { snprintf(g_comm_err, sizeof g_comm_err, "mkdir failed"); return IM_E_HIP; }
    *out = m;
    return IM_OK;
}
int im_comm_allgather(im_comm* m, const void* send, void* recv, size_t bytes, void* stream)
{
    (void)stream;
    char path[300], tmp[310];
    snprintf(path, sizeof path, "%s/ag.%d.%d", m->dir, m->seq, m->rank);
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE* fp = fopen(tmp, "wb");
    if (!fp || fwrite(send, 1, bytes, fp) != bytes) { snprintf(g_comm_err, sizeof g_comm_err, "cannot write %s", tmp); return IM_E_HIP; }
    fclose(fp);
    rename(tmp, path);
    for (int r = 0; r < m->world; r++) {
        snprintf(path, sizeof path, "%s/ag.%d.%d", m->dir, m->seq, r);
        for (int tries = 0;; tries++) {
            fp = fopen(path, "rb");
            if (fp) {
                const size_t got = fread((char*)recv + (size_t)r * bytes, 1, bytes, fp);
                fclose(fp);
                if (got == bytes) break;
            }
            if (tries > 6000) { snprintf(g_comm_err, sizeof g_comm_err, "rank %d never wrote %s", r, path); return IM_E_HIP; }
            struct timespec ts = { 0, 10 * 1000 * 1000 };
            nanosleep(&ts, NULL);
        }
    }
    m->seq++;
    return IM_OK;
}
/* the point-to-point step over files: every send leaves x.<step>.<from>.<to>.<k-th operation between the two>, every receive waits for its file */
int im_comm_exchange(im_comm* m, int32_t n, const int32_t* dir, const int32_t* peer, void* const* dev, const size_t* bytes, void* stream)
{
    (void)stream;
    int* nth = calloc((size_t)m->world * 2, sizeof(int));
    char path[320], tmp[330];
    for (int pass = 0; pass < 2; pass++)                /* all sends first: nothing here can wait on itself */
        for (int32_t k = 0; k < n; k++) {
            if (dir[k] != pass) continue;
            const int ord = nth[peer[k] * 2 + pass]++;
            if (pass == 0) {
                snprintf(path, sizeof path, "%s/x.%d.%d.%d.%d", m->dir, m->seq, m->rank, peer[k], ord);
                snprintf(tmp, sizeof tmp, "%s.tmp", path);
                FILE* fp = fopen(tmp, "wb");
                if (!fp || (bytes[k] && fwrite(dev[k], 1, bytes[k], fp) != bytes[k])) { snprintf(g_comm_err, sizeof g_comm_err, "cannot write %s", tmp); free(nth); return IM_E_HIP; }
                fclose(fp);
                rename(tmp, path);
            } else {
                snprintf(path, sizeof path, "%s/x.%d.%d.%d.%d", m->dir, m->seq, peer[k], m->rank, ord);
                for (int tries = 0;; tries++) {
                    FILE* fp = fopen(path, "rb");
                    if (fp) {
                        const size_t got = bytes[k] ? fread(dev[k], 1, bytes[k], fp) : 0;
                        fclose(fp);
                        if (got == bytes[k]) { unlink(path); break; }
                    }
                    if (tries > 6000) { snprintf(g_comm_err, sizeof g_comm_err, "rank %d never wrote %s", peer[k], path); free(nth); return IM_E_HIP; }
                    struct timespec ts = { 0, 10 * 1000 * 1000 };
                    nanosleep(&ts, NULL);
                }
            }
        }
    free(nth);
    m->seq++;
    return IM_OK;
}
int im_comm_allreduce_sum_i32(im_comm* m, int32_t* buf, size_t count, void* stream)
{
    int32_t* all = malloc(4 * count * (size_t)m->world);
    const int rc = im_comm_allgather(m, buf, all, 4 * count, stream);
    if (rc == IM_OK)
        for (size_t i = 0; i < count; i++) { int32_t s = 0; for (int r = 0; r < m->world; r++) s += all[(size_t)r * count + i]; buf[i] = s; }
    free(all);
    return rc;
}
int im_depth_allreduce(im_ctx* c, im_comm* m)
{
    if (!g_gdepth) return IM_E_ARG;
    for (int i = 0; i < c->n; i++) {
        const int rc = im_comm_allreduce_sum_i32(m, g_gdepth[i], (size_t)c->lens[i] + 1, NULL);
        if (rc != IM_OK) return rc;
    }
    return IM_OK;
}

void im_comm_destroy(im_comm* m)
{
    if (!m) return;
    if (m->rank == 0) { char cmd[300]; sleep(1); snprintf(cmd, sizeof cmd, "rm -rf '%s'", m->dir); if (system(cmd) != 0) { } }
    free(m);
}
