/*
 * simgen.c -- streaming generator of the synthetic inputs BASELINE.json's configs name, at any size (TEST / BENCH TOOLING,
 * not part of the product): reference FASTA, coordinate-sorted BAM, BAI, config file, list of planted events.
 *
 * The model is the one of indelminer_amd/synth.py, restated so that nothing is ever
 * held for more than one contig and one segment of records:
 *   reference     i.i.d. uniform A C G T (a counter-based generator: base i of contig t is a function of (seed, t, i))
 *   donor         the reference with an indel about every 2 kb (1-50 bases, half insertions; with --big-every k every
 *                 k-th event a 150-900 base deletion), position / size / inserted bases functions of (seed, t, event)
 *   pairs         2 x L bases, FR, insert ~ N(500, 50) clipped to [300, 700], fragment k of a contig starts at
 *                 k * len / pairs + a jitter, 0.5 % substitutions -- all functions of (seed, t, k)
 *   alignments    what a BWA-like mapper reports: an indel with >= 20 read bases on both sides -> CIGAR with I / D (D only up
 *                 to 50 bases), otherwise the shorter side soft-clipped, larger side < 30 bases -> unmapped at its mate;
 *                 MAPQ 60, MQ tag, no read group, proper pair iff both mapped and the template <= 700
 * Records are generated per SEGMENT of record positions (1 Mb), segments on a thread pool, written in order; a segment
 * regenerates the fragments that start up to 3 kb in front of it and keeps the records that fall inside.
 *
 *   simgen --prefix P [--seed S] [--lens a,b,c | --human TOTAL_BASES] [--coverage C] [--read-len L] [--big-every K]
 *          [--threads T] [--level Z] [--segment BASES]
 *   -> P.fa  P.bam  P.bam.bai  P.cfg  P.truth.tsv ; prints one line of JSON (records, pairs, bytes, seconds)
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <zlib.h>

static void die(const char* m) { fprintf(stderr, "simgen: %s\n", m); exit(1); }
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) die("out of memory"); return p; }
static void* xrealloc(void* p, size_t n) { p = realloc(p, n ? n : 1); if (!p) die("out of memory"); return p; }
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static inline uint64_t mix64(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static inline uint64_t key4(uint64_t seed, uint64_t a, uint64_t b, uint64_t c) { return mix64(mix64(mix64(mix64(seed) ^ a) ^ b) ^ c); }
typedef struct { uint64_t s; } rng_t;
static inline uint64_t rng_next(rng_t* r) { return mix64(r->s++ * 0x2545f4914f6cdd1dull + 0x1234567ull); }
static inline double rng_unit(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* ---- options ---- */
static uint64_t g_seed = 3;
static double g_cov = 30.0;
static int g_L = 100, g_big_every = 0, g_threads = 8, g_level = 1;
static int64_t g_segment = 1000000;
static const char* g_prefix = NULL;
static int g_nctg = 0;
static int64_t g_len[4096];
enum { ISZ_MEAN = 500, ISZ_SD = 50, ISZ_MIN = 300, ISZ_MAX = 700, SPACING = 2000, MARGIN = 1500, LOOKBACK = 3000 };

/* ---- the contig being generated ---- */
typedef struct { int64_t pos; int32_t size; int is_ins; } event_t;
static uint8_t* c_ref; static int64_t c_len; static int c_tid;
static event_t* c_ev; static int64_t c_nev;
static int64_t c_pairs, c_pair_base;

static void gen_reference_range(int64_t lo, int64_t hi)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    for (int64_t w = lo / 32; w * 32 < hi; w++) {
        uint64_t v = key4(g_seed, 0x5ef, (uint64_t)c_tid, (uint64_t)w);
        const int64_t b0 = w * 32;
        for (int j = 0; j < 32; j++) { const int64_t i = b0 + j; if (i >= lo && i < hi) c_ref[i] = (uint8_t)acgt[(v >> (2 * j)) & 3]; }
    }
}
typedef struct { int64_t lo, hi; } range_t;
static void* ref_thread(void* a) { range_t* r = a; gen_reference_range(r->lo, r->hi); return NULL; }

static void gen_events(void)
{
    c_nev = 0;
    const int64_t n_max = c_len / SPACING + 2;
    c_ev = xrealloc(c_ev, sizeof(event_t) * (size_t)n_max);
    int64_t j = 0;
    for (int64_t p0 = MARGIN; p0 < c_len - MARGIN; p0 += SPACING, j++) {
        const uint64_t h = key4(g_seed, 0xe7e, (uint64_t)c_tid, (uint64_t)j);
        event_t e;
        e.pos = p0 + (int64_t)(h % (SPACING / 2 + 1)) - SPACING / 4;
        e.size = 1 + (int32_t)((h >> 20) % 50);
        e.is_ins = (int)((h >> 40) & 1);
        if (g_big_every && (j % g_big_every) == g_big_every - 1) { e.size = 150 + (int32_t)((h >> 20) % 750); e.is_ins = 0; }
        c_ev[c_nev++] = e;
    }
}
static inline uint8_t ins_base(int64_t ev, int k)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    return (uint8_t)acgt[(key4(g_seed, 0x1a5, (uint64_t)c_tid, (uint64_t)ev * 1024 + (uint64_t)(k / 32)) >> (2 * (k % 32))) & 3];
}
static inline int64_t first_event_after(int64_t x)     /* first event with pos > x */
{
    int64_t j = (x - MARGIN + SPACING / 4) / SPACING - 1;
    if (j < 0) j = 0;
    if (j > c_nev) j = c_nev;
    while (j > 0 && c_ev[j - 1].pos > x) j--;
    while (j < c_nev && c_ev[j].pos <= x) j++;
    return j;
}

/* a position in the donor: reference position x, or k bases into the insertion that sits in front of x */
typedef struct { int64_t x; int32_t k; int64_t e; } dpos_t;       /* e = next event at or behind this point */

static dpos_t donor_at_ref(int64_t s)
{
    dpos_t d; d.k = -1;
    d.e = first_event_after(s);
    /* inside a deleted stretch: the donor goes on behind it; an insertion in front of s lies behind this point's past */
    if (d.e > 0 && !c_ev[d.e - 1].is_ins && s < c_ev[d.e - 1].pos + c_ev[d.e - 1].size && s >= c_ev[d.e - 1].pos) s = c_ev[d.e - 1].pos + c_ev[d.e - 1].size;
    d.x = s;
    return d;
}
/* n donor bases on from d; when out != NULL they are written there.  Reports the one event met strictly inside:
 * *a = bases in front of it, *ev = its index (-1: none), *n_ins = inserted bases taken */
static dpos_t donor_walk(dpos_t d, int n, uint8_t* out, int* a, int64_t* ev, int* n_ins)
{
    int got = 0;
    if (a) { *a = -1; *ev = -1; *n_ins = 0; }
    while (got < n) {
        if (d.k >= 0) {                                 /* inside the insertion in front of d.x (event d.e - 1) */
            const event_t* E = &c_ev[d.e - 1];
            const int take = E->size - d.k < n - got ? E->size - d.k : n - got;
            if (a && *ev < 0) { *a = got; *ev = d.e - 1; }
            if (a && *ev == d.e - 1) *n_ins += take;
            if (out) for (int i = 0; i < take; i++) out[got + i] = ins_base(d.e - 1, d.k + i);
            got += take; d.k += take;
            if (d.k == E->size) d.k = -1;
            continue;
        }
        const int64_t until = d.e < c_nev ? c_ev[d.e].pos : c_len + n;
        int64_t run = until - d.x;
        if (run > n - got) run = n - got;
        if (run > 0) {
            if (out) for (int64_t i = 0; i < run; i++) out[got + i] = d.x + i < c_len ? c_ref[d.x + i] : (uint8_t)'A';
            got += (int)run; d.x += run;
            if (got == n) break;
        }
        /* at an event */
        const event_t* E = &c_ev[d.e];
        if (E->is_ins) { d.k = 0; d.e++; }
        else {
            if (a && *ev < 0 && got > 0) { *a = got; *ev = d.e; }
            d.x += E->size; d.e++;
        }
    }
    return d;
}

/* ---- records ---- */
typedef struct { int32_t pos; uint32_t at, len; uint64_t key; } rref_t;
typedef struct {
    int tid; int64_t lo, hi;            /* record positions [lo, hi) */
    uint8_t* raw; size_t raw_n, raw_cap;
    rref_t* rr; size_t n, cap;
    /* after compression */
    uint8_t* comp; size_t comp_n, comp_cap;
    /* index material, relative to the segment's first byte */
    struct { int32_t bin; uint64_t beg, end; }* runs; size_t n_runs, cap_runs;
    uint64_t* lin; int64_t win0, n_lin;  /* first record touching window win0 + i (all ones: none) */
    int64_t n_pairs_seen;
    volatile int done;
} seg_t;

static inline int reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return ((1 << 15) - 1) / 7 + (int)(beg >> 14);
    if (beg >> 17 == end >> 17) return ((1 << 12) - 1) / 7 + (int)(beg >> 17);
    if (beg >> 20 == end >> 20) return ((1 << 9) - 1) / 7 + (int)(beg >> 20);
    if (beg >> 23 == end >> 23) return ((1 << 6) - 1) / 7 + (int)(beg >> 23);
    if (beg >> 26 == end >> 26) return ((1 << 3) - 1) / 7 + (int)(beg >> 26);
    return 0;
}

typedef struct { int unmapped; int64_t pos; int ncig; uint32_t cig[3]; int64_t span; uint8_t seq[1024]; } aln_t;

/* the alignment a BWA-like mapper reports for L donor bases from d (synth.py simulate, the per-read part) */
static void make_read(dpos_t d, aln_t* A)
{
    const int L = g_L;
    int a; int64_t ev; int n_ins;
    /* a deletion exactly where the read starts lies in front of the read */
    while (d.k < 0 && d.e < c_nev && c_ev[d.e].pos == d.x && !c_ev[d.e].is_ins) { d.x += c_ev[d.e].size; d.e++; }
    const int started_in_ins = d.k >= 0;
    const int64_t x0 = d.x;
    donor_walk(d, L, A->seq, &a, &ev, &n_ins);
    A->unmapped = 0; A->ncig = 1; A->cig[0] = ((uint32_t)L << 4) | 0; A->pos = x0; A->span = L;
    if (ev < 0) return;
    const event_t* E = &c_ev[ev];
    if (E->is_ins) {
        const int aa = started_in_ins ? 0 : a, b = L - aa - n_ins;
        if (n_ins <= 0) return;
        if (aa >= 20 && b >= 20) { A->ncig = 3; A->cig[0] = ((uint32_t)aa << 4) | 0; A->cig[1] = ((uint32_t)n_ins << 4) | 1; A->cig[2] = ((uint32_t)b << 4) | 0; A->span = aa + b; }
        else if ((aa > b ? aa : b) < 30) { A->unmapped = 1; A->ncig = 0; A->span = 0; }
        else if (aa >= b) { A->ncig = 2; A->cig[0] = ((uint32_t)aa << 4) | 0; A->cig[1] = ((uint32_t)(L - aa) << 4) | 4; A->span = aa; }
        else { A->ncig = 2; A->cig[0] = ((uint32_t)(L - b) << 4) | 4; A->cig[1] = ((uint32_t)b << 4) | 0; A->span = b; A->pos = E->pos; }
    } else {
        const int b = L - a;
        if (a <= 0 || b <= 0) return;
        if (a >= 20 && b >= 20 && E->size <= 50) { A->ncig = 3; A->cig[0] = ((uint32_t)a << 4) | 0; A->cig[1] = ((uint32_t)E->size << 4) | 2; A->cig[2] = ((uint32_t)b << 4) | 0; A->span = L + E->size; }
        else if (a >= b) { A->ncig = 2; A->cig[0] = ((uint32_t)a << 4) | 0; A->cig[1] = ((uint32_t)b << 4) | 4; A->span = a; }
        else { A->ncig = 2; A->cig[0] = ((uint32_t)a << 4) | 4; A->cig[1] = ((uint32_t)b << 4) | 0; A->span = b; A->pos = x0 + a + E->size; }
    }
}

static void substitute(uint8_t* seq, rng_t* r)
{
    static const char acgt[4] = { 'A', 'C', 'G', 'T' };
    /* 0.5 % per base: the number of substituted bases first (binomial by inversion), then where */
    const double u = rng_unit(r);
    double p = pow(0.995, g_L), c = p;
    int n = 0;
    while (u > c && n < g_L) { p *= (double)(g_L - n) / (double)(n + 1) * (0.005 / 0.995); c += p; n++; }
    for (int i = 0; i < n; i++) { const uint64_t v = rng_next(r); seq[v % (uint64_t)g_L] = (uint8_t)acgt[(v >> 32) & 3]; }
}

static const uint8_t kCode[256] = { ['A'] = 1, ['C'] = 2, ['G'] = 4, ['T'] = 8, ['N'] = 15 };
static const uint8_t kComp[256] = { ['A'] = 'T', ['C'] = 'G', ['G'] = 'C', ['T'] = 'A', ['N'] = 'N' };

static void seg_push(seg_t* S, int64_t pos, uint64_t key, int flag, int mapq, int mq, int64_t mpos, int64_t isize, const aln_t* A, int64_t pair_id)
{
    char name[32];
    const int nl = snprintf(name, sizeof name, "r%lld", (long long)pair_id) + 1;
    const int L = g_L, ncig = A->unmapped ? 0 : A->ncig;
    const int64_t end = A->span > 0 && !A->unmapped ? pos + A->span : pos + 1;
    const uint32_t body = 32u + (uint32_t)nl + 4u * (uint32_t)ncig + (uint32_t)((L + 1) / 2) + (uint32_t)L + 4u;
    if (S->raw_n + body + 4 > S->raw_cap) { S->raw_cap = (S->raw_cap + body) * 2 + (1 << 20); S->raw = xrealloc(S->raw, S->raw_cap); }
    if (S->n == S->cap) { S->cap = S->cap * 2 + 4096; S->rr = xrealloc(S->rr, sizeof(rref_t) * S->cap); }
    uint8_t* p = S->raw + S->raw_n;
    rref_t* R = &S->rr[S->n++];
    R->pos = (int32_t)pos; R->at = (uint32_t)S->raw_n; R->len = body + 4; R->key = key;
    S->raw_n += body + 4;
    int32_t w[9];
    w[0] = (int32_t)body; w[1] = S->tid; w[2] = (int32_t)pos;
    w[3] = (int32_t)((uint32_t)nl | ((uint32_t)mapq << 8) | ((uint32_t)reg2bin(pos, end) << 16));
    w[4] = (int32_t)((uint32_t)ncig | ((uint32_t)flag << 16));
    w[5] = L; w[6] = S->tid; w[7] = (int32_t)mpos; w[8] = (int32_t)isize;
    memcpy(p, w, 36); p += 36;
    memcpy(p, name, (size_t)nl); p += nl;
    for (int i = 0; i < ncig; i++) { memcpy(p, &A->cig[i], 4); p += 4; }
    uint8_t sq[1024];
    const uint8_t* seq = A->seq;
    if (A->unmapped && !(flag & 0x20)) { for (int i = 0; i < L; i++) sq[i] = kComp[A->seq[L - 1 - i]]; seq = sq; }     /* stored as sequenced */
    for (int i = 0; i < L; i += 2) *p++ = (uint8_t)((kCode[seq[i]] << 4) | (i + 1 < L ? kCode[seq[i + 1]] : 0));
    memset(p, 0x28, (size_t)L); p += L;
    p[0] = 'M'; p[1] = 'Q'; p[2] = 'C'; p[3] = (uint8_t)mq;
}

static int cmp_rref(const void* x, const void* y)
{
    const rref_t* a = x; const rref_t* b = y;
    if (a->pos != b->pos) return a->pos < b->pos ? -1 : 1;
    return a->key < b->key ? -1 : a->key > b->key;
}

static inline int64_t frag_start(int64_t k)
{
    const int64_t usable = c_len - LOOKBACK;                     /* fragments start where a whole template still fits */
    const int64_t base = (int64_t)((__int128)k * usable / c_pairs), next = (int64_t)((__int128)(k + 1) * usable / c_pairs);
    const int64_t g = next - base > 0 ? next - base : 1;
    return base + (int64_t)(key4(g_seed, 0xf7a, (uint64_t)c_tid, (uint64_t)k) % (uint64_t)g);
}

static void generate_segment(seg_t* S)
{
    const int L = g_L;
    const int64_t usable = c_len - LOOKBACK;
    if (usable <= 0 || c_pairs <= 0) return;
    int64_t k_lo = (int64_t)((__int128)(S->lo - LOOKBACK > 0 ? S->lo - LOOKBACK : 0) * c_pairs / usable) - 2;
    int64_t k_hi = (int64_t)((__int128)(S->hi < usable ? S->hi : usable) * c_pairs / usable) + 2;
    if (k_lo < 0) k_lo = 0;
    if (k_hi > c_pairs) k_hi = c_pairs;
    for (int64_t k = k_lo; k < k_hi; k++) {
        const int64_t s = frag_start(k);
        if (s >= S->hi) break;
        rng_t r = { key4(g_seed, 0x9a1, (uint64_t)c_tid, (uint64_t)k) };
        /* insert length: N(500, 50) by Box-Muller, rounded and clipped */
        const double u1 = rng_unit(&r), u2 = rng_unit(&r);
        double z = sqrt(-2.0 * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
        int64_t isz = (int64_t)llrint(ISZ_MEAN + ISZ_SD * z);
        if (isz < ISZ_MIN) isz = ISZ_MIN;
        if (isz > ISZ_MAX) isz = ISZ_MAX;
        aln_t A1, A2;
        const dpos_t d1 = donor_at_ref(s);
        make_read(d1, &A1);
        const dpos_t d2 = donor_walk(d1, (int)(isz - L), NULL, NULL, NULL, NULL);
        make_read(d2, &A2);
        substitute(A1.seq, &r);
        substitute(A2.seq, &r);
        if (A1.unmapped && A2.unmapped) continue;
        const int64_t p1 = A1.unmapped ? A2.pos : A1.pos, p2 = A2.unmapped ? A1.pos : A2.pos;
        if ((p1 < S->lo || p1 >= S->hi) && (p2 < S->lo || p2 >= S->hi)) continue;
        const int64_t e1 = p1 + (A1.unmapped ? 0 : A1.span), e2 = p2 + (A2.unmapped ? 0 : A2.span);
        const int64_t lo = p1 < p2 ? p1 : p2, hi = e1 > e2 ? e1 : e2, tl = hi - lo;
        const int any_unm = A1.unmapped || A2.unmapped;
        const int proper = !any_unm && tl <= ISZ_MAX;
        const int64_t pid = c_pair_base + k;
        if (p1 >= S->lo && p1 < S->hi) {
            const int flag = 0x1 | 0x40 | 0x20 | (A1.unmapped ? 0x4 : 0) | (A2.unmapped ? 0x8 : 0) | (proper ? 0x2 : 0);
            seg_push(S, p1, (uint64_t)k * 2, flag, A1.unmapped ? 0 : 60, A2.unmapped ? 0 : 60, p2, any_unm ? 0 : (p1 <= p2 ? tl : -tl), &A1, pid);
        }
        if (p2 >= S->lo && p2 < S->hi) {
            const int flag = 0x1 | 0x80 | 0x10 | (A2.unmapped ? 0x4 : 0) | (A1.unmapped ? 0x8 : 0) | (proper ? 0x2 : 0);
            seg_push(S, p2, (uint64_t)k * 2 + 1, flag, A2.unmapped ? 0 : 60, A1.unmapped ? 0 : 60, p1, any_unm ? 0 : (p2 <= p1 ? tl : -tl), &A2, pid);
        }
        S->n_pairs_seen++;
    }
    qsort(S->rr, S->n, sizeof(rref_t), cmp_rref);
}

/* one BGZF block (SAM/BAM specification 4.1) of data[0..n) appended to S->comp; returns its compressed size */
static size_t bgzf_block(seg_t* S, z_stream* zs, const uint8_t* data, size_t n)
{
    const size_t bound = n + n / 8 + 1024;
    if (S->comp_n + bound > S->comp_cap) { S->comp_cap = (S->comp_cap + bound) * 2; S->comp = xrealloc(S->comp, S->comp_cap); }
    uint8_t* o = S->comp + S->comp_n;
    static const uint8_t head[12] = { 31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0 };
    memcpy(o, head, 12); o[12] = 'B'; o[13] = 'C'; o[14] = 2; o[15] = 0;
    deflateReset(zs);
    zs->next_in = (Bytef*)data; zs->avail_in = (uInt)n;
    zs->next_out = o + 18; zs->avail_out = (uInt)(bound - 26);
    if (deflate(zs, Z_FINISH) != Z_STREAM_END) die("deflate failed");
    const size_t clen = bound - 26 - zs->avail_out, total = clen + 26;
    if (total > 65536) die("a BGZF block grew past 64 KiB");
    const uint16_t bs = (uint16_t)(total - 1);
    memcpy(o + 16, &bs, 2);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isz = (uint32_t)n;
    memcpy(o + 18 + clen, &crc, 4); memcpy(o + 22 + clen, &isz, 4);
    S->comp_n += total;
    return total;
}

static void compress_segment(seg_t* S)
{
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2 failed");
    uint8_t* blk = xmalloc(0x10000);
    size_t fill = 0;
    size_t first_in_block = 0;
    /* records never span blocks: whole records up to 0xFF00 payload bytes per block.  A record's virtual offset (relative to the
     * segment) is known when its block is compressed, so the index material is made block by block. */
    uint64_t* rel = xmalloc(sizeof(uint64_t) * (S->n + 1));
    S->win0 = (S->lo >> 14) - 1; S->n_lin = ((S->hi + 4096) >> 14) - S->win0 + 2;
    S->lin = xmalloc(sizeof(uint64_t) * (size_t)S->n_lin);
    for (int64_t q = 0; q < S->n_lin; q++) S->lin[q] = ~0ull;
    for (size_t i = 0; i <= S->n; i++) {
        const int flush = i == S->n || fill + S->rr[i].len > 0xFF00;
        if (flush && fill > 0) {
            const uint64_t coff = S->comp_n;
            bgzf_block(S, &zs, blk, fill);
            size_t off = 0;
            for (size_t j = first_in_block; j < i; j++) { rel[j] = (coff << 16) | off; off += S->rr[j].len; }
            fill = 0; first_in_block = i;
        }
        if (i < S->n) { memcpy(blk + fill, S->raw + S->rr[i].at, S->rr[i].len); fill += S->rr[i].len; }
    }
    rel[S->n] = (uint64_t)S->comp_n << 16;
    for (size_t i = 0; i < S->n; i++) {
        const uint8_t* rec = S->raw + S->rr[i].at;
        int32_t w[5]; memcpy(w, rec, 20);
        const int bin = (int)((uint32_t)w[3] >> 16);
        if (S->n_runs == 0 || S->runs[S->n_runs - 1].bin != bin) {
            if (S->n_runs == S->cap_runs) { S->cap_runs = S->cap_runs * 2 + 256; S->runs = xrealloc(S->runs, sizeof(*S->runs) * S->cap_runs); }
            S->runs[S->n_runs].bin = bin; S->runs[S->n_runs].beg = rel[i]; S->n_runs++;
        }
        S->runs[S->n_runs - 1].end = rel[i + 1];
        /* linear index: the first record that touches a 16 kb window; the end comes from the CIGAR's reference span */
        const int ncig = (int)((uint32_t)w[4] & 0xffff), nl = (int)((uint32_t)w[3] & 0xff);
        int64_t span = 0;
        for (int c = 0; c < ncig; c++) { uint32_t cw; memcpy(&cw, rec + 36 + nl + 4 * c, 4); const int op = (int)(cw & 15); if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += cw >> 4; }
        const int64_t pos = w[2], end = span > 0 ? pos + span : pos + 1;
        for (int64_t win = pos >> 14; win <= (end - 1) >> 14; win++) {
            const int64_t q = win - S->win0;
            if (q >= 0 && q < S->n_lin && rel[i] < S->lin[q]) S->lin[q] = rel[i];
        }
    }
    free(rel); free(blk);
    deflateEnd(&zs);
    free(S->raw); S->raw = NULL; free(S->rr); S->rr = NULL;
}

/* ---- thread pool over the segments of the contig ---- */
static seg_t* g_segs; static int64_t g_nseg; static volatile int64_t g_next_seg;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER; static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER;
static int64_t g_written;            /* segments the writer has taken: workers stay at most a window ahead of it */
static void* seg_thread(void* arg)
{
    (void)arg;
    for (;;) {
        pthread_mutex_lock(&g_mu);
        while (g_next_seg < g_nseg && g_next_seg >= g_written + 4 * g_threads) pthread_cond_wait(&g_cv, &g_mu);
        const int64_t i = g_next_seg < g_nseg ? g_next_seg++ : -1;
        pthread_mutex_unlock(&g_mu);
        if (i < 0) break;
        generate_segment(&g_segs[i]);
        compress_segment(&g_segs[i]);
        pthread_mutex_lock(&g_mu); g_segs[i].done = 1; pthread_cond_broadcast(&g_cv); pthread_mutex_unlock(&g_mu);
    }
    return NULL;
}

/* ---- BAI accumulation for one contig ---- */
typedef struct { uint64_t beg, end; } chunk_t;
typedef struct { chunk_t* c; int32_t n, cap; } binv_t;
enum { N_BINS = 37450 };

int main(int argc, char** argv)
{
    int64_t human_total = 0;
    for (int i = 1; i < argc; i++) {
        const char* a = argv[i];
        const char* v = i + 1 < argc ? argv[i + 1] : NULL;
        if (!strcmp(a, "--prefix") && v) { g_prefix = v; i++; }
        else if (!strcmp(a, "--seed") && v) { g_seed = strtoull(v, NULL, 10); i++; }
        else if (!strcmp(a, "--coverage") && v) { g_cov = atof(v); i++; }
        else if (!strcmp(a, "--read-len") && v) { g_L = atoi(v); i++; }
        else if (!strcmp(a, "--big-every") && v) { g_big_every = atoi(v); i++; }
        else if (!strcmp(a, "--threads") && v) { g_threads = atoi(v); i++; }
        else if (!strcmp(a, "--level") && v) { g_level = atoi(v); i++; }
        else if (!strcmp(a, "--segment") && v) { g_segment = atoll(v); i++; }
        else if (!strcmp(a, "--human") && v) { human_total = atoll(v); i++; }
        else if (!strcmp(a, "--lens") && v) {
            char* s = strdup(v);
            for (char* t = strtok(s, ","); t; t = strtok(NULL, ",")) { if (g_nctg == 4096) die("too many contigs"); g_len[g_nctg++] = atoll(t); }
            free(s); i++;
        } else die("usage: simgen --prefix P [--seed S] [--lens a,b,c | --human TOTAL] [--coverage C] [--read-len L] [--big-every K] [--threads T] [--level Z]");
    }
    if (!g_prefix) die("--prefix is required");
    if (human_total > 0) {
        /* 24 contigs with the length spread of the human assembly (GRCh38 chr1-22, X, Y), scaled to the total asked for */
        static const double h[24] = { 248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
                                      135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
                                      46709983, 50818468, 156040895, 57227415 };
        double tot = 0; for (int i = 0; i < 24; i++) tot += h[i];
        g_nctg = 24;
        for (int i = 0; i < 24; i++) g_len[i] = (int64_t)(h[i] / tot * (double)human_total);
    }
    if (g_nctg == 0) { g_nctg = 1; g_len[0] = 1000000; }
    if (g_L < 30 || g_L > 1000 || g_threads < 1 || g_threads > 256 || g_segment < 20000) die("bad option value");
    for (int i = 0; i < g_nctg; i++) if (g_len[i] < 3 * LOOKBACK || g_len[i] >= (1ll << 29)) die("contig lengths must lie in [7500, 2^29)");
    const double t_start = now_s();
    char path[1024];
    snprintf(path, sizeof path, "%s.fa", g_prefix); FILE* ffa = fopen(path, "wb");
    snprintf(path, sizeof path, "%s.bam", g_prefix); FILE* fbam = fopen(path, "wb");
    snprintf(path, sizeof path, "%s.truth.tsv", g_prefix); FILE* ftr = fopen(path, "w");
    if (!ffa || !fbam || !ftr) die("cannot open the output files");
    setvbuf(ffa, NULL, _IOFBF, 1 << 22); setvbuf(fbam, NULL, _IOFBF, 1 << 22);
    /* BAM header */
    uint64_t file_off = 0;
    {
        size_t cap = 1 << 16, n = 0;
        char* text = xmalloc(cap);
        n += (size_t)snprintf(text + n, cap - n, "@HD\tVN:1.0\tSO:coordinate\n");
        for (int i = 0; i < g_nctg; i++) { if (n + 128 > cap) { cap *= 2; text = xrealloc(text, cap); } n += (size_t)snprintf(text + n, cap - n, "@SQ\tSN:ctg%d\tLN:%lld\n", i, (long long)g_len[i]); }
        size_t hcap = n + 64 + 32 * (size_t)g_nctg, hn = 0;
        uint8_t* hdr = xmalloc(hcap);
        memcpy(hdr, "BAM\1", 4); hn = 4;
        int32_t v = (int32_t)n; memcpy(hdr + hn, &v, 4); hn += 4; memcpy(hdr + hn, text, n); hn += n;
        v = g_nctg; memcpy(hdr + hn, &v, 4); hn += 4;
        for (int i = 0; i < g_nctg; i++) {
            char nm[32]; const int l = snprintf(nm, sizeof nm, "ctg%d", i) + 1;
            v = l; memcpy(hdr + hn, &v, 4); hn += 4; memcpy(hdr + hn, nm, (size_t)l); hn += (size_t)l;
            v = (int32_t)g_len[i]; memcpy(hdr + hn, &v, 4); hn += 4;
        }
        seg_t H; memset(&H, 0, sizeof H);
        z_stream zs; memset(&zs, 0, sizeof zs);
        deflateInit2(&zs, g_level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        for (size_t at = 0; at < hn; at += 0xFF00) bgzf_block(&H, &zs, hdr + at, hn - at < 0xFF00 ? hn - at : 0xFF00);
        deflateEnd(&zs);
        fwrite(H.comp, 1, H.comp_n, fbam); file_off = H.comp_n;
        free(H.comp); free(hdr); free(text);
    }
    /* the BAI is kept in memory and written at the end */
    size_t bai_cap = 1 << 20, bai_n = 0;
    uint8_t* bai = xmalloc(bai_cap);
#define BAI_PUT(ptr, bytes) do { if (bai_n + (bytes) > bai_cap) { bai_cap = (bai_cap + (bytes)) * 2; bai = xrealloc(bai, bai_cap); } memcpy(bai + bai_n, (ptr), (bytes)); bai_n += (bytes); } while (0)
    BAI_PUT("BAI\1", 4);
    { int32_t v = g_nctg; BAI_PUT(&v, 4); }
    binv_t* bins = calloc(N_BINS, sizeof(binv_t));
    int64_t total_records = 0, total_pairs = 0, total_events = 0;
    for (c_tid = 0; c_tid < g_nctg; c_tid++) {
        c_len = g_len[c_tid];
        c_ref = xrealloc(c_ref, (size_t)c_len + 64);
        {
            pthread_t th[256]; range_t rg[256];
            for (int t = 0; t < g_threads; t++) {
                rg[t].lo = (c_len / g_threads * t) / 32 * 32; rg[t].hi = t == g_threads - 1 ? c_len : (c_len / g_threads * (t + 1)) / 32 * 32;
                pthread_create(&th[t], NULL, ref_thread, &rg[t]);
            }
            for (int t = 0; t < g_threads; t++) pthread_join(th[t], NULL);
        }
        fprintf(ffa, ">ctg%d\n", c_tid);
        for (int64_t at = 0; at < c_len; at += 60) { const size_t n = (size_t)(c_len - at < 60 ? c_len - at : 60); fwrite(c_ref + at, 1, n, ffa); fputc('\n', ffa); }
        gen_events();
        for (int64_t j = 0; j < c_nev; j++) fprintf(ftr, "%d\t%lld\t%d\t%s\n", c_tid, (long long)c_ev[j].pos, c_ev[j].size, c_ev[j].is_ins ? "INS" : "DEL");
        total_events += c_nev;
        c_pairs = (int64_t)llrint(g_cov * (double)c_len / (2.0 * g_L));
        c_pair_base = total_pairs;
        total_pairs += c_pairs;
        g_nseg = (c_len + g_segment - 1) / g_segment;
        g_segs = calloc((size_t)g_nseg, sizeof(seg_t));
        for (int64_t i = 0; i < g_nseg; i++) { g_segs[i].tid = c_tid; g_segs[i].lo = i * g_segment; g_segs[i].hi = (i + 1) * g_segment < c_len ? (i + 1) * g_segment : c_len + LOOKBACK; }
        g_next_seg = 0; g_written = 0;
        pthread_t th[256];
        for (int t = 0; t < g_threads; t++) pthread_create(&th[t], NULL, seg_thread, NULL);
        const int64_t n_win = ((c_len + LOOKBACK) >> 14) + 2;
        uint64_t* lin = xmalloc(sizeof(uint64_t) * (size_t)n_win);
        for (int64_t w = 0; w < n_win; w++) lin[w] = ~0ull;
        int64_t max_win = -1;
        for (int64_t i = 0; i < g_nseg; i++) {
            seg_t* S = &g_segs[i];
            pthread_mutex_lock(&g_mu);
            while (!S->done) pthread_cond_wait(&g_cv, &g_mu);
            pthread_mutex_unlock(&g_mu);
            fwrite(S->comp, 1, S->comp_n, fbam);
            const uint64_t base = file_off << 16;
            for (size_t r = 0; r < S->n_runs; r++) {
                binv_t* B = &bins[S->runs[r].bin];
                const uint64_t beg = S->runs[r].beg + base, end = S->runs[r].end + base;
                if (B->n > 0 && B->c[B->n - 1].end == beg) { B->c[B->n - 1].end = end; continue; }
                if (B->n == B->cap) { B->cap = B->cap * 2 + 4; B->c = xrealloc(B->c, sizeof(chunk_t) * (size_t)B->cap); }
                B->c[B->n].beg = beg; B->c[B->n].end = end; B->n++;
            }
            for (int64_t r = 0; r < S->n_lin; r++) {
                const int64_t w = S->win0 + r;
                if (w < 0 || w >= n_win || S->lin[r] == ~0ull) continue;
                const uint64_t vo = S->lin[r] + base;
                if (vo < lin[w]) lin[w] = vo;
                if (w > max_win) max_win = w;
            }
            /* every record is one record: count them through the runs' sizes is not possible, so the workers count */
            file_off += S->comp_n;
            total_records += (int64_t)S->n;
            free(S->comp); S->comp = NULL; free(S->runs); S->runs = NULL; free(S->lin); S->lin = NULL;
            pthread_mutex_lock(&g_mu); g_written = i + 1; pthread_cond_broadcast(&g_cv); pthread_mutex_unlock(&g_mu);
        }
        for (int t = 0; t < g_threads; t++) pthread_join(th[t], NULL);
        /* this contig's part of the BAI: bins with their chunks, then the linear index (a window nobody touched repeats its predecessor) */
        int32_t n_bin = 0;
        for (int b = 0; b < N_BINS; b++) if (bins[b].n > 0) n_bin++;
        BAI_PUT(&n_bin, 4);
        for (int b = 0; b < N_BINS; b++) {
            if (bins[b].n == 0) continue;
            const uint32_t ub = (uint32_t)b; BAI_PUT(&ub, 4); BAI_PUT(&bins[b].n, 4);
            BAI_PUT(bins[b].c, sizeof(chunk_t) * (size_t)bins[b].n);
            bins[b].n = 0;
        }
        const int32_t n_intv = (int32_t)(max_win + 1);
        BAI_PUT(&n_intv, 4);
        uint64_t prev = 0;
        for (int64_t w = 0; w < n_intv; w++) { if (lin[w] != ~0ull) prev = lin[w]; BAI_PUT(&prev, 8); }
        free(lin); free(g_segs);
    }
    static const uint8_t eof_block[28] = { 0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    fwrite(eof_block, 1, 28, fbam);
    fclose(fbam); fclose(ffa); fclose(ftr);
    snprintf(path, sizeof path, "%s.bam.bai", g_prefix);
    FILE* fb = fopen(path, "wb");
    if (!fb || fwrite(bai, 1, bai_n, fb) != bai_n) die("cannot write the index");
    fclose(fb);
    snprintf(path, sizeof path, "%s.cfg", g_prefix);
    FILE* fc = fopen(path, "w");
    if (!fc) die("cannot write the config file");
    fprintf(fc, "IL generic %d %d\n", ISZ_MIN, ISZ_MAX);
    fclose(fc);
    printf("{\"lens\": [");
    for (int i = 0; i < g_nctg; i++) printf("%s%lld", i ? ", " : "", (long long)g_len[i]);
    printf("], ");
    printf("\"contigs\": %d, \"reference_bases\": %lld, \"pairs\": %lld, \"records\": %lld, \"events\": %lld, \"bam_bytes\": %llu, \"seconds\": %.2f}\n",
           g_nctg, (long long)({ int64_t s = 0; for (int i = 0; i < g_nctg; i++) s += g_len[i]; s; }), (long long)total_pairs, (long long)total_records,
           (long long)total_events, (unsigned long long)file_off + 28, now_s() - t_start);
    return 0;
}
