/*
 * im_shim.c -- TEST-ONLY stand-in for libindelminer_amd.so.
 *
 * Implements the handful of C-ABI entry points the host driver calls with the CPU oracle
 * (oracle/im_oracle.c), so that the HOST LOGIC of the driver (BAM/FASTA readers, fetch_func
 * dispatch, flush replay, merge, VCF writer) can be tested without a GPU (-m "not gpu").
 * It lives under tests/ and is linked only into tests/shim/indelminer_shim; the product
 * binary links the HIP library and has no CPU path.
 */
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
