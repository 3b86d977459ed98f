/*
 * im_oracle.c -- plain-C CPU restatement of indelMINER's split-read hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see im_oracle.h).  Re-entrant, no globals, no
 * linked lists: the algorithm of the reference restated over flat arrays.
 * Every function cites the reference lines it follows (relative to
 * /root/reference/).  Tie-breaking rules are kept literally; allocator churn,
 * debug printing and the O(contig) strlen (src/alignment.c:771) are not.
 */
#include "im_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define NEGINF   (-9999999)     /* MININT, src/localalign.c:3 */
#define S_MATCH     1           /* src/localalign.c:10-13 */
#define S_MISMATCH (-10)
#define S_GAPOPEN   10
#define S_GAPEXT    10

#define CIG(len, op) (((uint32_t)(len) << 4) | (uint32_t)(op))
#define CIG_OP(c)  ((int)((c) & 15u))
#define CIG_LEN(c) ((int)((c) >> 4))

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

/* substitution weight: W[i][j] = MATCH iff i == j on raw bytes (src/localalign.c:61-67) */
static inline int wsub(char a, char b) { return a == b ? S_MATCH : S_MISMATCH; }

/* 2-bit base code, everything that is not C/G/T maps to 0 (src/alignment.c:11-24) */
static inline uint32_t base_code(char c)
{
    switch (c) {
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 0;
    }
}

/* ------------------------------------------------------------------ K1 -- */

typedef struct { uint32_t code; uint32_t pos; } kmer_at;

static int cmp_kmer(const void* a, const void* b)
{
    const kmer_at* x = a; const kmer_at* y = b;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/*
 * find_best_band (src/alignment.c:393-447) with its helpers read_seeds (29-68),
 * bin_diagonals (70-128), bin_bands (130-140), select_band (142-181).
 *
 * The reference indexes both sequences in 4^k tables with collision chains; the
 * observable result only depends on (a) which read k-mers occur exactly once in
 * the read piece (bin_diagonals:97-98) and (b) every window position carrying
 * the same k-mer (chain walk 101-112).  Restated: sort the read k-mers, look
 * every window k-mer up by binary search.
 */
int imo_find_best_band(const imo_params* P,
                       const char* ref, uint32_t zstart1, uint32_t end1, uint32_t anchor,
                       const char* read, uint32_t zstart2, uint32_t end2,
                       int* plow, int* pup, int* pindex, int* pcount)
{
    const uint32_t k = P->klength, g = P->numgaps;
    const uint32_t W = end1 - zstart1, L = end2 - zstart2;
    /* unsigned arithmetic as in the reference (403-404) */
    const uint32_t numdiag = (W - (k - 1)) + (L - (k - 1));
    if (pindex) *pindex = -1;
    if (pcount) *pcount = 0;
    if (!(numdiag > g)) return IMO_ABORT;           /* forceassert, 405 */
    if (end2 < zstart2) return IMO_ABORT;           /* forceassert, 407 */
    if (L < k) {                                    /* 408-412 */
        *plow = (int)(numdiag - 1);
        *pup  = (int)(numdiag - 1);
        return 0;
    }
    if ((int32_t)numdiag <= 0 || numdiag > (1u << 28)) return IMO_ABORT; /* reference would run off its arrays */

    const uint32_t mask = (k >= 16) ? 0xffffffffu : ((1u << (2 * k)) - 1u);
    const uint32_t nread = L - k + 1;
    kmer_at* rk = malloc(sizeof(kmer_at) * nread);
    int* diag = calloc(numdiag, sizeof(int));
    if (!rk || !diag) { free(rk); free(diag); return IMO_OVERFLOW; }

    const char* rs = read + zstart2;
    uint32_t code = 0;
    for (uint32_t i = 0; i < L; i++) {
        code = ((code << 2) | base_code(rs[i])) & mask;
        if (i + 1 >= k) { rk[i + 1 - k].code = code; rk[i + 1 - k].pos = i + 1 - k; }
    }
    qsort(rk, nread, sizeof(kmer_at), cmp_kmer);

    if (W >= k) {
        const char* ws = ref + zstart1;
        code = 0;
        for (uint32_t i = 0; i < W; i++) {
            code = ((code << 2) | base_code(ws[i])) & mask;
            if (i + 1 < k) continue;
            const uint32_t p = i + 1 - k;            /* 0-based window k-mer start */
            /* binary search for code among the read k-mers */
            uint32_t lo = 0, hi = nread;
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if (rk[mid].code < code) lo = mid + 1; else hi = mid;
            }
            if (lo == nread || rk[lo].code != code) continue;
            if (lo + 1 < nread && rk[lo + 1].code == code) continue;   /* not unique in read (97-98) */
            const uint32_t q = rk[lo].pos;           /* 0-based read k-mer start */
            /* indx = j-1-(i-k+1)+readlen-k+1 with j=p+1 (102-105) */
            const uint32_t indx = p - q + L - k + 1;
            if (indx < numdiag) diag[indx] += 1;     /* 106-111 */
        }
    }

    /* bin_bands + select_band, fused: bands[i] = sum diag[i..i+g] for i < numdiag-g, else 0 */
    const int anchor_rel = (int)(anchor - zstart1);  /* passed as uint, received as int (431,146) */
    int best = 0, dist = INT_MAX;
    uint32_t indx = 0;
    for (uint32_t i = 0; i < numdiag; i++) {
        int b = 0;
        if (i < numdiag - g)
            for (uint32_t j = i; j < i + g + 1; j++) b += diag[j];
        const int d = abs((int)((uint32_t)anchor_rel - i));
        if (b > best) { best = b; indx = i; dist = d; }
        else if (b == best && d < dist) { indx = i; dist = d; }
    }
    if (pindex) *pindex = (int)indx;
    if (pcount) *pcount = best;
    *plow = (int)(indx - (L - k + 1));               /* 438-439 */
    *pup  = (int)(indx + g - (L - k + 1));
    free(rk); free(diag);
    return 0;
}

/* --------------------------------------------------------------- K2, K3 -- */

/* edit-script writer of the global aligner (DEL/INS/REP macros, src/globalalign.c:36-59) */
typedef struct {
    int* sapp;      /* append pointer */
    int  last;      /* last op appended */
    int  g, h, m;   /* gap open, extend, open+extend */
    /* work arrays shared by every level of the recursion (src/globalalign.c:19-26) */
    int *CC, *DD, *CP, *DP;
    int *MP[3]; char *MT[3];
    int *FP; char *FT;
    int IP;
} galign;

static inline void s_del(galign* G, int k)
{ if (G->last < 0) G->last = G->sapp[-1] -= k; else G->last = *G->sapp++ = -k; }
static inline void s_ins(galign* G, int k)
{ if (G->last > 0) G->last = G->sapp[-1] += k; else G->last = *G->sapp++ = k; }
static inline void s_rep(galign* G)
{ G->last = *G->sapp++ = 0; }

/*
 * align() -- src/globalalign.c:66-307: optimal conversion of A[1..M] into
 * B[1..N] inside diagonals [low,up], linear space, by locating where the
 * optimal path crosses the middle diagonal and recursing on the pieces.
 * tb/te: 1/2 = no gap-open charge if the path begins/ends with a delete/insert.
 * Every comparison below keeps the reference's strictness.
 */
static int galign_rec(galign* G, const char* A, const char* B, int M, int N,
                      int low, int up, char tb, char te)
{
    const int g = G->g, h = G->h, m = G->m;
    int *CC = G->CC, *DD = G->DD, *CP = G->CP, *DP = G->DP;
    int rmid, k, l, r, v, kt;
    int t1, t2, t3;

    if (N <= 0) { if (M > 0) s_del(G, M); return -1; }      /* 82-85 */
    if (M <= 0) { s_ins(G, N); return -1; }                  /* 86-89 */
    int band = up - low + 1;
    if (band <= 1) { for (int i = 1; i <= M; i++) s_rep(G); return -1; }   /* 90-93 */

    {
        int midd = band / 2 + 1;
        rmid = low + midd - 1;
        int leftd = 1 - low;
        int rightd = up - low + 1;
        int fr, j, i, c = 0, d = 0, e = 0, t, ib, curd;

        if (leftd < midd) {                                  /* 102-111 */
            fr = -1;
            for (j = 0; j < midd; j++) CP[j] = DP[j] = -1;
            for (j = midd; j <= rightd; j++) CP[j] = DP[j] = 0;
            G->MP[0][0] = G->MP[1][0] = G->MP[2][0] = -1;
        } else if (leftd > midd) {                           /* 112-121 */
            fr = leftd - midd;
            for (j = 0; j <= midd; j++) CP[j] = DP[j] = fr;
            for (j = midd + 1; j <= rightd; j++) CP[j] = DP[j] = -1;
            G->MP[0][fr] = G->MP[1][fr] = G->MP[2][fr] = -1;
        } else {                                             /* 122-133 */
            fr = 0;
            for (j = 0; j <= rightd; j++) CP[j] = DP[j] = 0;
            G->MP[0][0] = G->MP[1][0] = G->MP[2][0] = -1;
        }
        (void)fr;

        CC[leftd] = 0;                                       /* 135-146 */
        t = (tb == 2) ? 0 : -g;
        for (j = leftd + 1; j <= rightd; j++) { CC[j] = t = t - h; DD[j] = t - g; }
        CC[rightd + 1] = NEGINF;
        DD[rightd + 1] = NEGINF;
        DD[leftd] = (tb == 1) ? 0 : -g;
        CC[leftd - 1] = NEGINF;

        for (i = 1; i <= M; i++) {                           /* 147-234 */
            if (i > N - up) rightd--;
            if (leftd > 1) leftd--;
            const char ai = A[i];
            if ((c = CC[leftd + 1] - m) > (d = DD[leftd + 1] - h)) { d = c; DP[leftd] = CP[leftd + 1]; }
            else DP[leftd] = DP[leftd + 1];
            if ((ib = leftd + low - 1 + i) > 0) c = CC[leftd] + wsub(ai, B[ib]);
            if (d > c || ib <= 0) { c = d; CP[leftd] = DP[leftd]; }
            e = c - g;
            DD[leftd] = d;
            CC[leftd] = c;
            G->IP = CP[leftd];
            if (leftd == midd) CP[leftd] = DP[leftd] = G->IP = i;
            for (curd = leftd + 1; curd <= rightd; curd++) {
                if (curd != midd) {                          /* 166-188 */
                    if ((c = c - m) > (e = e - h)) { e = c; G->IP = CP[curd - 1]; }
                    if ((c = CC[curd + 1] - m) > (d = DD[curd + 1] - h)) { d = c; DP[curd] = CP[curd + 1]; }
                    else DP[curd] = DP[curd + 1];
                    c = CC[curd] + wsub(ai, B[curd + low - 1 + i]);
                    if (c < d || c < e) {
                        if (e > d) { c = e; CP[curd] = G->IP; }
                        else       { c = d; CP[curd] = DP[curd]; }
                    }
                    CC[curd] = c;
                    DD[curd] = d;
                } else {                                     /* 189-232: on the middle diagonal */
                    if ((c = c - m) > (e = e - h)) { e = c; G->MP[1][i] = CP[curd - 1]; G->MT[1][i] = 2; }
                    else { G->MP[1][i] = G->IP; G->MT[1][i] = 2; }
                    if ((c = CC[curd + 1] - m) > (d = DD[curd + 1] - h)) { d = c; G->MP[2][i] = CP[curd + 1]; G->MT[2][i] = 1; }
                    else { G->MP[2][i] = DP[curd + 1]; G->MT[2][i] = 1; }
                    c = CC[curd] + wsub(ai, B[curd + low - 1 + i]);
                    if (c < d || c < e) {
                        if (e > d) { c = e; G->MP[0][i] = G->MP[1][i]; G->MT[0][i] = 2; }
                        else       { c = d; G->MP[0][i] = G->MP[2][i]; G->MT[0][i] = 1; }
                    } else { G->MP[0][i] = i - 1; G->MT[0][i] = 0; }
                    if (c - g > e) { G->MP[1][i] = G->MP[0][i]; G->MT[1][i] = G->MT[0][i]; }
                    if (c - g > d) { G->MP[2][i] = G->MP[0][i]; G->MT[2][i] = G->MT[0][i]; }
                    CP[curd] = DP[curd] = G->IP = i;
                    CC[curd] = c;
                    DD[curd] = d;
                }
            }
        }

        /* which end state to trace back from (236-249) */
        if (te == 1 && d + g > c)      { k = DP[rightd]; l = 2; }
        else if (te == 2 && e + g > c) { k = G->IP;      l = 1; }
        else                           { k = CP[rightd]; l = 0; }
        if (rmid > N - M) l = 2;
        else if (rmid < N - M) l = 1;
        v = c;
    }

    /* chain of crossing points, reversed into FP/FT (253-258) */
    r = -1;
    for (; k > -1; r = k, k = G->MP[l][r], l = G->MT[l][r]) { G->FP[k] = r; G->FT[k] = (char)l; }

    if (r == -1) {                                           /* never crossed the middle (260-262) */
        if (rmid < 0) galign_rec(G, A, B, M, N, rmid + 1, up, tb, te);
        else          galign_rec(G, A, B, M, N, low, rmid - 1, tb, te);
    } else {
        k = r; l = G->FP[k]; kt = G->FT[k];
        if (rmid < 0) {                                      /* first block (269-275) */
            galign_rec(G, A, B, r - 1, r + rmid, rmid + 1, imin(up, r + rmid), tb, 1);
            s_del(G, 1);
        } else if (rmid > 0) {
            galign_rec(G, A, B, r, r + rmid - 1, imax(-r, low), rmid - 1, tb, 2);
            s_ins(G, 1);
        }
        t2 = up - rmid - 1;                                  /* intermediate blocks (278-293) */
        t3 = low - rmid + 1;
        for (; l > -1; k = l, l = G->FP[k], kt = G->FT[k]) {
            if (kt == 0) s_rep(G);
            else if (kt == 1) {
                s_ins(G, 1);
                t1 = l - k - 1;
                galign_rec(G, A + k, B + k + rmid + 1, t1, t1, 0, imin(t1, t2), 2, 1);
                s_del(G, 1);
            } else {
                s_del(G, 1);
                t1 = l - k - 1;
                galign_rec(G, A + k + 1, B + k + rmid, t1, t1, imax(-t1, t3), 0, 1, 2);
                s_ins(G, 1);
            }
        }
        if (N - M > rmid) {                                  /* last block (296-304) */
            s_ins(G, 1);
            t1 = k + rmid + 1;
            galign_rec(G, A + k, B + t1, M - k, N - t1, 0, imin(N - t1, t2), 2, te);
        } else if (N - M < rmid) {
            s_del(G, 1);
            t1 = M - (k + 1);
            galign_rec(G, A + k + 1, B + k + rmid, t1, N - (k + rmid), imax(-t1, t3), 0, 1, te);
        }
    }
    return v;
}

/* ALIGN -- src/globalalign.c:333-401.  A,B are 1-based (A[1..M]).  Returns the
 * score; *check receives CHECK_SCORE (311-330) of the script. */
static int galign_top(const char* A, const char* B, int M, int N, int low, int up,
                      int gopen, int gext, int* S, int* check)
{
    galign G;
    memset(&G, 0, sizeof G);
    G.g = gopen; G.h = gext; G.m = gopen + gext;
    G.sapp = S; G.last = 0;
    low = imin(imax(-M, low), imin(N - M, 0));               /* 347-348 */
    up  = imax(imin(N, up), imax(N - M, 0));
    int c;
    if (N <= 0) { if (M > 0) s_del(&G, M); *check = 0; return -(M <= 0 ? 0 : gopen + gext * M); }
    if (M <= 0) { s_ins(&G, N); *check = 0; return -(N <= 0 ? 0 : gopen + gext * N); }
    int band = up - low + 1;
    if (band <= 1) {                                         /* 358-365 */
        c = 0;
        for (int i = 1; i <= M; i++) { s_rep(&G); c += wsub(A[i], B[i]); }
        *check = c;
        return c;
    }
    size_t jb = (size_t)(band + 2);
    size_t jm = (size_t)(M + 1);
    G.CC = malloc(jb * sizeof(int)); G.DD = malloc(jb * sizeof(int));
    G.CP = malloc(jb * sizeof(int)); G.DP = malloc(jb * sizeof(int));
    for (int t = 0; t < 3; t++) { G.MT[t] = malloc(jm); G.MP[t] = malloc(jm * sizeof(int)); }
    G.FT = malloc(jm); G.FP = malloc(jm * sizeof(int));

    c = galign_rec(&G, A, B, M, N, low, up, 0, 0);

    /* CHECK_SCORE, 311-330 */
    {
        int i = 0, j = 0, score = 0; const int* s = S;
        while (i < M || j < N) {
            int op = *s++;
            if (op == 0) { ++i; ++j; score += wsub(A[i], B[j]); }
            else if (op > 0) { score -= gopen + op * gext; j += op; }
            else { score -= gopen - op * gext; i -= op; }
        }
        *check = score;
    }
    free(G.CC); free(G.DD); free(G.CP); free(G.DP);
    for (int t = 0; t < 3; t++) { free(G.MT[t]); free(G.MP[t]); }
    free(G.FT); free(G.FP);
    return c;
}

/*
 * local_align -- src/localalign.c:15-196.  seq1 = read piece (M), seq2 = window
 * (N), band [low,up].  Forward banded Gotoh pass finds the first strictly-best
 * end cell, reverse pass stops at the first cell that reaches the best score,
 * then ALIGN produces the script for the located sub-strings.
 */
static int local_align_restated(const char* seq1, int M, const char* seq2, int N,
                                int low, int up,
                                int* psi, int* psj, int* pei, int* pej, int* S, int* check_bad)
{
    const char* A = seq1 - 1;       /* 1-based views (42-43) */
    const char* B = seq2 - 1;
    const int G = S_GAPOPEN, H = S_GAPEXT, m = G + H;
    int i, j, si, ei, c, d, e = 0, t, leftd, rightd, curd, ib;
    int best = 0, starti = 0, startj = 0, endi, endj;
    int flag = 0;

    low = imax(-M, low);            /* 70-71 */
    up  = imin(N, up);
    const int band = up - low + 1;
    if (band < 1) return INT_MIN;   /* reference prints and exit(1)s (74-77) */

    int* CC = malloc((size_t)(band + 3) * sizeof(int));
    int* DD = malloc((size_t)(band + 3) * sizeof(int));

    if (low > 0) leftd = 1;         /* 82-99 */
    else if (up < 0) leftd = band;
    else leftd = 1 - low;
    rightd = band;
    si = imax(0, -up);
    ei = imin(M, N - low);
    CC[leftd] = 0;
    for (j = leftd + 1; j <= rightd; j++) { CC[j] = 0; DD[j] = -G; }
    CC[rightd + 1] = NEGINF;
    DD[rightd + 1] = NEGINF;
    endi = si;
    endj = si + low;
    CC[leftd - 1] = NEGINF;
    DD[leftd] = -G;

    for (i = si + 1; i <= ei; i++) {                         /* 100-131 */
        if (i > N - up) rightd--;
        if (leftd > 1) leftd--;
        const char ai = A[i];
        if ((c = CC[leftd + 1] - m) > (d = DD[leftd + 1] - H)) d = c;
        if ((ib = leftd + low - 1 + i) > 0) c = CC[leftd] + wsub(ai, B[ib]);
        if (d > c) c = d;
        if (c < 0) c = 0;
        e = c - G;
        DD[leftd] = d;
        CC[leftd] = c;
        if (c > best) { best = c; endi = i; endj = ib; }
        for (curd = leftd + 1; curd <= rightd; curd++) {
            if ((c = c - m) > (e = e - H)) e = c;
            if ((c = CC[curd + 1] - m) > (d = DD[curd + 1] - H)) d = c;
            c = CC[curd] + wsub(ai, B[curd + low - 1 + i]);
            if (e > c) c = e;
            if (d > c) c = d;
            if (c < 0) c = 0;
            CC[curd] = c;
            DD[curd] = d;
            if (c > best) { best = c; endi = i; endj = curd + low - 1 + i; }
        }
    }

    leftd = imax(1, -endi - low + 1);                        /* 132-143 */
    rightd = band - (up - (endj - endi));
    CC[rightd] = 0;
    t = -G;
    for (j = rightd - 1; j >= leftd; j--) { CC[j] = t = t - H; DD[j] = t - G; }
    for (j = rightd + 1; j <= band; ++j) CC[j] = NEGINF;
    CC[leftd - 1] = DD[leftd - 1] = NEGINF;
    DD[rightd] = -G;

    for (i = endi; i >= 1; i--) {                            /* 144-176 */
        if (i + low <= 0) leftd++;
        if (rightd < band) rightd++;
        const char ai = A[i];
        if ((c = CC[rightd - 1] - m) > (d = DD[rightd - 1] - H)) d = c;
        if ((ib = rightd + low - 1 + i) <= N) c = CC[rightd] + wsub(ai, B[ib]);
        if (d > c) c = d;
        e = c - G;
        DD[rightd] = d;
        CC[rightd] = c;
        if (c == best) { starti = i; startj = ib; flag = 1; break; }
        for (curd = rightd - 1; curd >= leftd; curd--) {
            if ((c = c - m) > (e = e - H)) e = c;
            if ((c = CC[curd - 1] - m) > (d = DD[curd - 1] - H)) d = c;
            c = CC[curd] + wsub(ai, B[curd + low - 1 + i]);
            if (e > c) c = e;
            if (d > c) c = d;
            CC[curd] = c;
            DD[curd] = d;
            if (c == best) { starti = i; startj = curd + low - 1 + i; flag = 1; break; }
        }
        if (flag == 1) break;
    }
    free(CC); free(DD);

    if (starti < 0 || starti > M || startj < 0 || startj > N) return 0;   /* 180-185 */
    *psi = starti; *psj = startj; *pei = endi; *pej = endj;
    if ((endi - starti) == 0 || (endj - startj) == 0) return 0;           /* 191-193 */

    int check = 0;
    int score = galign_top(A + starti - 1, B + startj - 1,
                           endi - starti + 1, endj - startj + 1,
                           low - (startj - starti), up - (startj - starti),
                           G, H, S, &check);                              /* 195 */
    if (check != score && check_bad) *check_bad = 1;                      /* src/globalalign.c:384-385 */
    return score;
}

/* cigar run appender of fetch_cigar (add_operation, src/globalalign.c:465-505) */
static int push_op(uint32_t* ops, int* n, int op, int len)
{
    if (*n >= IMO_MAX_OPS) return -1;
    ops[(*n)++] = CIG(len, op);
    return 0;
}

/*
 * fetch_cigar -- src/globalalign.c:507-604.  A,B 1-based views of the aligned
 * sub-strings, S the edit script, AP the 1-based read start.  Note the
 * reference adds deletion run lengths to the consumed-read total as well
 * (numtotal += numrun for every run, 541-595); kept.
 */
static int script_to_cigar(const char* A, const char* B, int M, int N, const int* S,
                           int AP, int readlength, uint32_t* ops, int* pn, int* pmm)
{
    enum { R_EQ = 0, R_DEL = 1, R_INS = 2, R_X = 4 };
    static const int run2op[5] = { IMO_OP_EQ, IMO_OP_D, IMO_OP_I, -1, IMO_OP_X };
    int i = 0, j = 0, op = 0, mm = 0, n = 0;
    AP--;
    if (AP > 0 && push_op(ops, &n, IMO_OP_S, AP)) return IMO_OVERFLOW;
    int run = -1, numrun = 0, numtotal = AP;
    while (i < M || j < N) {
        int kind;
        if (op == 0 && *S == 0) { op = *S++; i++; j++; if (A[i] == B[j]) kind = R_EQ; else { kind = R_X; mm++; } }
        else {
            if (op == 0) op = *S++;
            if (op > 0) { op--; j++; kind = R_DEL; }
            else        { op++; i++; kind = R_INS; }
        }
        if (run != -1 && run != kind) {
            if (push_op(ops, &n, run2op[run], numrun)) return IMO_OVERFLOW;
            numtotal += numrun;
            run = kind; numrun = 1;
        } else { run = kind; numrun += 1; }
    }
    if (run != -1 && numrun > 0) {
        if (push_op(ops, &n, run2op[run], numrun)) return IMO_OVERFLOW;
        numtotal += numrun;
    }
    if (numtotal < readlength && push_op(ops, &n, IMO_OP_S, readlength - numtotal)) return IMO_OVERFLOW;
    *pn = n; *pmm = mm;
    return 0;
}

/* attempt_band_alignment -- src/alignment.c:343-391 */
int imo_band_alignment(const imo_params* P,
                       const char* ref, uint32_t zstart1, uint32_t end1,
                       const char* read, uint32_t zstart2, uint32_t end2,
                       int low, int up, imo_band_aln* out)
{
    (void)P;
    memset(out, 0, sizeof *out);
    out->low = low; out->up = up;
    if (low > up) return IMO_ABORT;                          /* forceassert, 359 */
    const int N = (int)(end1 - zstart1), M = (int)(end2 - zstart2);
    if (M <= 0 || N <= 0) return IMO_ABORT;                  /* forceassert(strlen > 0), src/localalign.c:31-32 */
    int* S = calloc((size_t)(M + N) + 2, sizeof(int));
    int si = 0, sj = 0, ei = 0, ej = 0, bad = 0;
    int score = local_align_restated(read + zstart2, M, ref + zstart1, N, low, up,
                                     &si, &sj, &ei, &ej, S, &bad);
    if (score == INT_MIN) { free(S); return IMO_ABORT; }
    if (bad) out->mismatches = -1;      /* CHECK_SCORE disagreed: the reference prints "Check_score=" to stdout and carries on */
    if (score <= 0) { free(S); return 0; }                   /* all zeros (365-372) */
    int rc = script_to_cigar(read + zstart2 + si - 2, ref + zstart1 + sj - 2,
                             ei - si + 1, ej - sj + 1, S, si, M,
                             out->ops, &out->n_ops, &out->mismatches);    /* 374-376 */
    free(S);
    if (rc) return rc;
    out->r1 = sj + (int)zstart1 - 1;                         /* 385-388 */
    out->r2 = ej + (int)zstart1;
    out->q1 = si + (int)zstart2 - 1;
    out->q2 = ei + (int)zstart2;
    return 0;
}

/* ------------------------------------------------------------------ K4 -- */

/* count_matches -- src/alignment.c:219-303 (q1 == 0 and q3 == q2 asserted there) */
static int split_score(const uint32_t* c1, int n1, int q2,
                       const uint32_t* c2, int n2, int q4, int* pmm)
{
    const int q3 = q2;
    int i, j, matches = 0, mm = 0;
    for (i = 0, j = 0; i < n1; i++) {
        int len = CIG_LEN(c1[i]), op = CIG_OP(c1[i]);
        if (op != IMO_OP_D) j += len;
        if (j < q2) { if (op == IMO_OP_EQ) matches += len; else if (op == IMO_OP_X) mm += len; }
        if (j >= q2) {
            if (op == IMO_OP_EQ) matches += q2 - (j - len); else if (op == IMO_OP_X) mm += q2 - (j - len);
            break;
        }
    }
    for (i = 0, j = 0; i < n2; i++) {
        int len = CIG_LEN(c2[i]), op = CIG_OP(c2[i]);
        if (op != IMO_OP_D) j += len;
        if (j >= q3) {
            if (op == IMO_OP_EQ) matches += j - q3; else if (op == IMO_OP_X) mm += j - q3;
            i += 1;
            break;
        }
    }
    for (; i < n2; i++) {
        int len = CIG_LEN(c2[i]), op = CIG_OP(c2[i]);
        if (op != IMO_OP_D) j += len;
        if (j < q4) { if (op == IMO_OP_EQ) matches += len; else if (op == IMO_OP_X) mm += len; }
        if (j >= q4) {
            if (op == IMO_OP_EQ) matches += q4 - (j - len); else if (op == IMO_OP_X) mm += q4 - (j - len);
            break;
        }
    }
    *pmm = mm;
    return matches;
}

/* find_best_del_candidate -- src/alignment.c:306-339 */
static int best_split(int q1, int q2, const uint32_t* c1, int n1,
                      int q3, int q4, const uint32_t* c2, int n2, int readlength, int* pindex)
{
    if (q1 != 0 || q3 > q2) return IMO_ABORT;
    int bestm = 0, bestmm = INT_MAX, index = -1;
    for (int i = q3; i <= q2; i++) {
        int mm, matches = split_score(c1, n1, i, c2, n2, q4, &mm);
        if (matches > readlength) return IMO_ABORT;
        if (matches > bestm || (matches == bestm && mm < bestmm)) { bestm = matches; bestmm = mm; index = i; }
        if (matches == readlength && mm == 0) break;
    }
    if (index == -1) return IMO_ABORT;
    *pindex = index;
    return 0;
}

/* ----------------------------------------------------------------- a10 -- */

/*
 * update_readsegs -- src/readaln.c:348-458.  Builds the final segment list as
 * packed ops in out->ops (every new_readseg call appends one op; start/end
 * follow from ref_start and the ops, src/readaln.c:24-99).
 */
static int build_segments(int r1, const uint32_t* c1, int n1, int index,
                          int q2, int r2, const uint32_t* c2, int n2, imo_result* out)
{
    int i, j, n = 0;
    int refindx = r1;
    uint32_t* ops = out->ops;
    out->ref_start = r1;
#define EMIT(len, op) do { if (push_op(ops, &n, (op), (len))) return IMO_OVERFLOW; \
                           if ((op) == IMO_OP_EQ || (op) == IMO_OP_X || (op) == IMO_OP_D || (op) == IMO_OP_M) refindx += (len); } while (0)
    for (i = 0, j = 0; i < n1; i++) {                        /* 362-385 */
        int op = CIG_OP(c1[i]), len = CIG_LEN(c1[i]);
        if (len <= 0) return IMO_ABORT;
        if (op != IMO_OP_D) j += len;
        if (j <= index) EMIT(len, op);
        if (j > index) {
            int part = index - (j - len);
            if (part > 0) EMIT(part, op);
            break;
        }
    }
    int rindex = r2, nextindex = index;
    if (c2 == NULL) n2 = 0;
    if (index >= q2) {                                       /* 389-412 */
        int offset = 0;
        for (i = 0, j = 0; i < n2; i++) {
            int op = CIG_OP(c2[i]), len = CIG_LEN(c2[i]);
            if (op != IMO_OP_D) j += len;
            if (j <= q2) { }
            else if (j > q2 && j <= index) {
                if (op != IMO_OP_I) { offset += len; if ((j - len) <= q2) offset -= q2 - (j - len); }
            } else if (j > index) {
                if (op != IMO_OP_I) { if ((j - len) <= index) offset += index - (j - len); }
            }
        }
        rindex = r2 + offset;
    } else {                                                 /* 413-421 */
        EMIT(q2 - index, IMO_OP_I);
        nextindex += q2 - index;
    }
    if (refindx < rindex) EMIT(rindex - refindx, IMO_OP_D);  /* 424-430 */
    for (i = 0, j = 0; i < n2; i++) {                        /* 432-446 */
        int op = CIG_OP(c2[i]), len = CIG_LEN(c2[i]);
        if (op != IMO_OP_D) j += len;
        if (j > nextindex) { EMIT(j - nextindex, op); i++; break; }
    }
    for (; i < n2; i++) EMIT(CIG_LEN(c2[i]), CIG_OP(c2[i])); /* 448-453 */
#undef EMIT
    out->n_ops = n;
    return 0;
}

/*
 * add_evidence_from_segment + new_evidence (src/alignment.c:449-476,
 * src/evidence.c:4-34): one record per D / I segment, with the per-evidence
 * reductions print_variants / print_vcf_output later take over aln1 and aln3
 * (src/variant.c:217-290, 704-775).
 */
static int collect_evidence(imo_result* out)
{
    int n = 0;
    int refpos = out->ref_start, readpos = 0;
    for (int s = 0; s < out->n_ops; s++) {
        int op = CIG_OP(out->ops[s]), len = CIG_LEN(out->ops[s]);
        if (op == IMO_OP_D || op == IMO_OP_I) {
            if (n >= IMO_MAX_EV) return IMO_OVERFLOW;
            imo_evidence* e = &out->ev[n++];
            memset(e, 0, sizeof *e);
            e->cls = (op == IMO_OP_D) ? 1 : 0;
            e->b1 = refpos;
            e->b2 = (op == IMO_OP_D) ? refpos + len : refpos;
            e->seg = s;
            e->read_off = readpos;
            for (int t = 0; t < out->n_ops; t++) {
                if (t == s) continue;
                int o = CIG_OP(out->ops[t]), l = CIG_LEN(out->ops[t]);
                int* flank = (t < s) ? &e->lflank : &e->rflank;
                switch (o) {
                case IMO_OP_EQ: *flank += l; break;
                case IMO_OP_X:  *flank += l; e->nd_print += l; e->nd_filter += l; break;
                case IMO_OP_I:  *flank += l; e->nd_print += l; e->nd_filter += l; break;
                case IMO_OP_D:  e->nd_print += l; e->nd_filter += l; break;
                case IMO_OP_S:  e->nd_filter += l; break;
                default: return IMO_ABORT;
                }
            }
        }
        if (op == IMO_OP_EQ || op == IMO_OP_X || op == IMO_OP_D) refpos += len;
        if (op != IMO_OP_D) readpos += len;
    }
    out->n_ev = n;
    return 0;
}

/* add_prefix_soft_clip / add_suffix_soft_clip -- src/alignment.c:478-532 */
static int clip_prefix(int clip, imo_band_aln* a)
{
    if (clip == 0) return 0;
    if (CIG_OP(a->ops[0]) == IMO_OP_S) { a->ops[0] = CIG(CIG_LEN(a->ops[0]) + clip, IMO_OP_S); return 0; }
    if (a->n_ops >= IMO_MAX_OPS) return IMO_OVERFLOW;
    memmove(a->ops + 1, a->ops, (size_t)a->n_ops * sizeof(uint32_t));
    a->ops[0] = CIG(clip, IMO_OP_S);
    a->n_ops++;
    return 0;
}
static int clip_suffix(int clip, imo_band_aln* a)
{
    if (clip == 0) return 0;
    if (a->n_ops <= 0) return IMO_ABORT;
    if (CIG_OP(a->ops[a->n_ops - 1]) == IMO_OP_S) {
        a->ops[a->n_ops - 1] = CIG(CIG_LEN(a->ops[a->n_ops - 1]) + clip, IMO_OP_S); return 0; }
    if (a->n_ops >= IMO_MAX_OPS) return IMO_OVERFLOW;
    a->ops[a->n_ops++] = CIG(clip, IMO_OP_S);
    return 0;
}

/* ------------------------------------------------------------ a2, a8 ---- */

/* attempt_diagonal_alignments -- src/alignment.c:539-759 (case table: SURVEY.md A.13) */
static int two_piece(const imo_params* P, const char* ref,
                     int32_t left1, int32_t right1, int32_t left2, int32_t right2,
                     int32_t anchor, const char* read, int readlength, imo_result* out)
{
    if (!(anchor >= left1 && anchor >= left2 && anchor <= right1 && anchor <= right2 &&
          left2 >= 0 && right2 > 0)) return IMO_ABORT;       /* 548-553 */
    const uint32_t L = (uint32_t)readlength, eth = P->ethreshold;
    int rc, low, up;
    imo_band_aln* a1 = &out->piece[0];
    imo_band_aln* a2 = &out->piece[1];

    rc = imo_find_best_band(P, ref, (uint32_t)left1, (uint32_t)right1, (uint32_t)anchor,
                            read, 0, L, &low, &up, NULL, NULL);          /* 557 */
    if (rc) return rc;
    out->n_band = 1;
    out->win_bytes[0] = right1 - left1; out->piece_bytes[0] = readlength;
    rc = imo_band_alignment(P, ref, (uint32_t)left1, (uint32_t)right1, read, 0, L, low, up, a1);
    if (rc) return rc;
    const int r1 = a1->r1, r2 = a1->r2, q1 = a1->q1, q2 = a1->q2;
    if (q1 == q2) return IMO_NONE;                           /* 568-572 */

    if (q1 == 0 && q2 == readlength) {                       /* 575-582 */
        rc = build_segments(r1, a1->ops, a1->n_ops, readlength, 0, -1, NULL, 0, out);
        if (rc) return rc;
        rc = collect_evidence(out);
        if (rc) return rc;
        return out->n_ev > 0 ? IMO_OK : IMO_NONE;            /* empty list == NULL */
    }

    int i, j;                                                /* 585-599 */
    for (i = 0, j = 0; i < a1->n_ops; i++) {
        int op = CIG_OP(a1->ops[i]);
        if (i == 0 && op == IMO_OP_S) continue;
        if (op != IMO_OP_EQ) break;
        j += CIG_LEN(a1->ops[i]);
    }
    const uint32_t f = (uint32_t)j;
    for (i = a1->n_ops - 1, j = 0; i >= 0; i--) {
        int op = CIG_OP(a1->ops[i]);
        if (i == a1->n_ops - 1 && op == IMO_OP_S) continue;
        if (op != IMO_OP_EQ) break;
        j += CIG_LEN(a1->ops[i]);
    }
    const uint32_t l = (uint32_t)j;

    /* piece 2: window, anchor, read piece per the four geometric cases (605-717).
     * The guards are evaluated in unsigned arithmetic exactly as written there. */
    uint32_t w0, w1, anc, p0, p1; int want_tail;
    if (r1 > anchor) {
        if (q1 == 0) {
            if (!(L > f)) return IMO_ABORT;
            if ((L - f) < eth || ((uint32_t)right2 - (uint32_t)r1 - f) < eth) return IMO_NONE;
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)right2; anc = (uint32_t)r1; p0 = f; p1 = L; want_tail = 1;
        } else if (q2 == readlength) {
            if (!(L > l)) return IMO_ABORT;
            if ((L - l) < eth || ((uint32_t)r2 - l - (uint32_t)anchor) < eth) return IMO_NONE;
            w0 = (uint32_t)anchor; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = L - l; want_tail = 0;
        } else return IMO_NONE;
    } else if (r1 < anchor) {
        if (r2 >= anchor) return IMO_NONE;
        if (q1 == 0) {
            if (!(L > f)) return IMO_ABORT;
            if ((L - f) < eth || ((uint32_t)anchor - (uint32_t)r1 - f) < eth) return IMO_NONE;
            w0 = (uint32_t)r1 + f; w1 = (uint32_t)anchor; anc = (uint32_t)r1; p0 = f; p1 = L; want_tail = 1;
        } else if (q2 == readlength) {
            if (!(L > l)) return IMO_ABORT;
            if ((L - l) < eth || ((uint32_t)r2 - l - (uint32_t)left2) < eth) return IMO_NONE;
            w0 = (uint32_t)left2; w1 = (uint32_t)r2 - l; anc = (uint32_t)r2; p0 = 0; p1 = L - l; want_tail = 0;
        } else return IMO_NONE;
    } else return IMO_NONE;                                  /* r1 == anchor (712-717) */

    if ((int32_t)(w1 - w0) <= 0) return IMO_ABORT;           /* reference would read out of bounds */
    rc = imo_find_best_band(P, ref, w0, w1, anc, read, p0, p1, &low, &up, NULL, NULL);
    if (rc) return rc;
    out->n_band = 2;
    out->win_bytes[1] = (int32_t)(w1 - w0); out->piece_bytes[1] = (int32_t)(p1 - p0);
    rc = imo_band_alignment(P, ref, w0, w1, read, p0, p1, low, up, a2);
    if (rc) return rc;
    const int r3 = a2->r1, r4 = a2->r2, q3 = a2->q1, q4 = a2->q2;
    if (want_tail) {
        if (q4 != readlength || q3 == q4) return IMO_NONE;
        rc = clip_prefix((int)f, a2);
    } else {
        if (q3 != 0 || q3 == q4) return IMO_NONE;
        rc = clip_suffix((int)l, a2);
    }
    if (rc) return rc;

    if (!(q1 < q2 && q3 < q4)) return IMO_ABORT;             /* 720-721 */
    int index = -1;
    if (q1 > q3 && q1 <= q4) {                               /* 724-731 */
        rc = best_split(q3, q4, a2->ops, a2->n_ops, q1, q2, a1->ops, a1->n_ops, readlength, &index);
        if (rc) return rc;
        rc = build_segments(r3, a2->ops, a2->n_ops, index, q1, r1, a1->ops, a1->n_ops, out);
    } else if (q3 > q1 && q3 <= q2) {                        /* 732-739 */
        rc = best_split(q1, q2, a1->ops, a1->n_ops, q3, q4, a2->ops, a2->n_ops, readlength, &index);
        if (rc) return rc;
        rc = build_segments(r1, a1->ops, a1->n_ops, index, q3, r3, a2->ops, a2->n_ops, out);
    } else if (q1 > q4 && r1 == r4) {                        /* 740-744 */
        rc = build_segments(r3, a2->ops, a2->n_ops, q4, q1, r1, a1->ops, a1->n_ops, out);
    } else if (q3 > q2 && r2 == r3) {                        /* 745-749 */
        rc = build_segments(r1, a1->ops, a1->n_ops, q2, q3, r3, a2->ops, a2->n_ops, out);
    } else return IMO_NONE;
    if (rc) return rc;
    rc = collect_evidence(out);
    if (rc) return rc;
    return out->n_ev > 0 ? IMO_OK : IMO_NONE;
}

/* attempt_pe_alignment -- src/alignment.c:764-799 (window geometry 774-783) */
int imo_realign(const imo_params* P, const char* contig, int32_t contig_len,
                int32_t anchor, int32_t range_max,
                const char* read, int32_t readlen, imo_result* out)
{
    memset(out, 0, sizeof *out);
    int32_t distance = range_max;
    int32_t left1  = anchor >= distance ? anchor - distance : 0;
    int32_t right1 = contig_len < (anchor + distance) ? contig_len : anchor + distance;
    distance = range_max + (int32_t)P->maxdelsize;
    int32_t left2  = anchor >= distance ? anchor - distance : 0;
    int32_t right2 = contig_len < (anchor + distance) ? contig_len : anchor + distance;
    int st = two_piece(P, contig, left1, right1, left2, right2, anchor, read, readlen, out);
    out->status = st;
    if (st != IMO_OK) { out->n_ev = 0; }
    return st;
}

int imo_realign_batch(const imo_params* P,
                      int32_t n_contigs, const char* const* contigs, const int32_t* contig_len,
                      int32_t n, const uint8_t* bases, const int64_t* off,
                      const int32_t* tid, const int32_t* anchor, const int32_t* range_max,
                      imo_result* out)
{
    char* tmp = NULL; size_t cap = 0;
    for (int32_t i = 0; i < n; i++) {
        if (tid[i] < 0 || tid[i] >= n_contigs) { free(tmp); return IMO_ABORT; }
        size_t len = (size_t)(off[i + 1] - off[i]);
        if (len + 1 > cap) { cap = len + 64; tmp = realloc(tmp, cap); }
        memcpy(tmp, bases + off[i], len); tmp[len] = 0;
        imo_realign(P, contigs[tid[i]], contig_len[tid[i]], anchor[i], range_max[i],
                    tmp, (int32_t)len, &out[i]);
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ K5 -- */

typedef struct { int32_t b1, b2, arrival; } ev_key;

static int cmp_ev(const void* a, const void* b)
{
    const ev_key* x = a; const ev_key* y = b;
    if (x->b1 != y->b1) return x->b1 < y->b1 ? -1 : 1;     /* sort_evidence, src/evidence.c:50-58 */
    if (x->b2 != y->b2) return x->b2 < y->b2 ? -1 : 1;
    /* allevidence is a prepend list and glibc qsort is a stable merge sort:
     * ties come out newest first */
    return x->arrival > y->arrival ? -1 : (x->arrival < y->arrival);
}

int32_t imo_cluster_sr(int32_t n, const int32_t* cls, const int32_t* b1, const int32_t* b2,
                       int32_t marker, int32_t tie_desc,
                       int32_t* order, int32_t* cl_first, int32_t* cl_count, uint8_t* used)
{
    ev_key* keys = malloc(sizeof(ev_key) * (size_t)(n > 0 ? n : 1));
    for (int32_t i = 0; i < n; i++) { keys[i].b1 = b1[i]; keys[i].b2 = b2[i]; keys[i].arrival = i; used[i] = 0; }
    qsort(keys, (size_t)n, sizeof(ev_key), cmp_ev);
    /* nodes are made for the sorted prefix up to the first b2 >= marker (src/indelminer.c:137-146) */
    int32_t m = 0;
    while (m < n && keys[m].b2 < marker) m++;
    int32_t ncl = 0, pos = 0;
    int32_t i = 0;
    while (i < m) {
        /* a run of identical (b1,b2); split by class (src/graph.c:122-127) */
        int32_t j = i;
        while (j < m && keys[j].b1 == keys[i].b1 && keys[j].b2 == keys[i].b2) j++;
        for (int c = 0; c < 2; c++) {
            int32_t first = pos, cnt = 0;
            if (!tie_desc) { for (int32_t t = j - 1; t >= i; t--) if (cls[keys[t].arrival] == c) { order[pos++] = keys[t].arrival; cnt++; } }
            else           { for (int32_t t = i; t < j; t++)      if (cls[keys[t].arrival] == c) { order[pos++] = keys[t].arrival; cnt++; } }
            if (cnt) { cl_first[ncl] = first; cl_count[ncl] = cnt; ncl++; }
        }
        for (int32_t t = i; t < j; t++) used[keys[t].arrival] = 1;
        i = j;
    }
    free(keys);
    return ncl;
}

/* ------------------------------------------------------------------ K7 -- */

#include <ctype.h>

void imo_sw_indel(const char* t1, int32_t len1, const char* t2, int32_t len2,
                  int32_t* subs_out, int32_t* indels_out, int32_t* aligned_out)
{
    const int match = 2, mismatch = 1, gopen = 4, gextend = 1;      /* src/variant.c:1290-1293 */
    enum { D_SUB = 0, D_INS = 1, D_DEL = 2 };
    const size_t W = (size_t)len1 + 1;
    int* V = calloc(((size_t)len2 + 1) * W, sizeof(int));
    char* I = calloc(((size_t)len2 + 1) * W, 1);
    int* F = calloc(W, sizeof(int));
    for (int i = 0; i <= len2; i++) V[(size_t)i * W] = -gopen - i * gextend;
    for (int j = 0; j <= len1; j++) V[j] = -gopen - j * gextend;
    int max_score = 0, max_i = -1, max_j = -1;
    for (int i = 1; i <= len2; i++) {
        int E = 0;
        for (int j = 1; j <= len1; j++) {
            int ifsub = V[(size_t)(i - 1) * W + j - 1];
            ifsub = toupper((unsigned char)t1[j - 1]) == toupper((unsigned char)t2[i - 1]) ? ifsub + match : ifsub - mismatch;
            const int ifins = imax(F[j], V[(size_t)(i - 1) * W + j] - gopen) - gextend;
            F[j] = ifins;
            const int ifdel = imax(E, V[(size_t)i * W + j - 1] - gopen) - gextend;
            E = ifdel;
            const int ifindel = imax(ifins, ifdel);
            int v = ifsub; char dir = D_SUB;
            if (v < ifindel) { dir = (ifins >= ifdel) ? D_INS : D_DEL; v = ifindel; }
            V[(size_t)i * W + j] = v; I[(size_t)i * W + j] = dir;
            if (v > max_score) { max_score = v; max_i = i; max_j = j; }
        }
    }
    int subs = 0, ins = 0, dels = 0, aligned = 0;
    int score = max_score, i = max_i, j = max_j;
    while (score > 0) {
        const char dir = I[(size_t)i * W + j];
        if (dir == D_SUB) { if (t1[j - 1] != t2[i - 1]) subs++; aligned++; i--; j--; }
        else if (dir == D_INS) { ins++; aligned++; i--; }
        else { dels++; j--; }
        score = V[(size_t)i * W + j];
    }
    aligned++;      /* the counting loop starts at k = strlen(nt1): NUL == NUL counts as aligned (1392-1404) */
    free(V); free(I); free(F);
    *subs_out = subs; *indels_out = ins + dels; *aligned_out = aligned;
}
