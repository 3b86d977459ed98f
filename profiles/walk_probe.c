/* walk_probe.c -- how fast can the host deliver records?  The walkers' record loop alone (hostio.c: BGZF inflate, record
 * framing, base qualities dropped, copy into a chunk), T threads over pieces of equal compressed size, nothing else.
 *
 *   gcc -O2 -std=gnu11 -pthread -Iindelminer_amd/host -o /tmp/walk_probe profiles/walk_probe.c indelminer_amd/host/hostio.c \
 *       indelminer_amd/host/iminflate.c -lz -lm
 *   /tmp/walk_probe aln.bam 16 [piece_bytes]
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "hostio.h"

typedef struct { int32_t tid, beg, end; } piece;
static piece* g_pieces; static int g_n, g_next; static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static const char* g_bam; static bai_index* g_idx; static long g_records; static long g_bytes;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void* worker(void* arg)
{
    (void)arg;
    bgzf_reader* r = bgzf_open(g_bam);
    bgzf_set_workers(r, 0);
    bam_header* h = bam_header_load(r);
    uint8_t* chunk = malloc(1 << 25);
    bam_record b; memset(&b, 0, sizeof b);
    long n = 0, bytes = 0;
    for (;;) {
        pthread_mutex_lock(&g_mu); const int i = g_next < g_n ? g_next++ : -1; pthread_mutex_unlock(&g_mu);
        if (i < 0) break;
        bam_region_iter it;
        if (bam_piece_begin(&it, r, g_idx, g_pieces[i].tid, g_pieces[i].beg, g_pieces[i].end) != 0) continue;
        it.drop_qual = 1;
        int64_t at = 0;
        for (;;) {
            int32_t len = 0;
            const int rc = bam_region_next_raw(&it, chunk + at, (1 << 25) - at, &len, &b);
            if (rc == -2) { at = 0; continue; }
            if (rc <= 0) break;
            n++; bytes += len; at += (len + 3) & ~3;
            if (at > (1 << 25) - 70000) at = 0;
        }
    }
    pthread_mutex_lock(&g_mu); g_records += n; g_bytes += bytes; pthread_mutex_unlock(&g_mu);
    free(chunk); bam_header_free(h); bgzf_close(r);
    return NULL;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: walk_probe aln.bam threads [piece_bytes]\n"); return 2; }
    g_bam = argv[1];
    const int nt = atoi(argv[2]);
    const int64_t piece_bytes = argc > 3 ? atoll(argv[3]) : 64 << 20;
    bgzf_reader* r = bgzf_open(g_bam);
    bam_header* h = bam_header_load(r);
    g_idx = bai_load(g_bam);
    int cap = 1 << 16;
    g_pieces = malloc(sizeof(piece) * (size_t)cap);
    for (int t = 0; t < h->n_targets; t++) {
        int32_t cuts[4096];
        const int nc = bai_split_points(g_idx, t, h->target_len[t], piece_bytes, cuts, 4096);
        int32_t beg = 0;
        for (int k = 0; k <= nc && g_n < cap; k++) { const int32_t end = k < nc ? cuts[k] : h->target_len[t]; g_pieces[g_n].tid = t; g_pieces[g_n].beg = beg; g_pieces[g_n].end = end; g_n++; beg = end; }
    }
    pthread_t th[256];
    const double t0 = now();
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, worker, NULL);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    const double t = now() - t0;
    printf("{\"threads\": %d, \"pieces\": %d, \"records\": %ld, \"delivered_mb\": %.1f, \"seconds\": %.3f, \"records_per_s\": %.4g, \"ns_per_record_per_thread\": %.1f}\n",
           nt, g_n, g_records, g_bytes / 1e6, t, g_records / t, t * 1e9 * nt / (double)g_records);
    return 0;
}
