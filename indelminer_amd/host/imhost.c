/*
 * imhost.c -- the indelminer host driver: same CLI, config file and VCF output as the
 * reference (src/indelminer.c), with the split-read realignment and the split-read
 * clustering done on the GPU through the C ABI of include/indelminer_amd.h.
 *
 * Host side = what BASELINE.json's north_star keeps on the host: BAM decode, the
 * fetch_func dispatch rules, insert-length estimation, paired-read evidence, variant
 * merge / filter / emit.  There is no CPU implementation of the two GPU seams in this
 * program: without the device it stops with the library's error.
 *
 * Two passes per contig:
 *   pass A  walk the BAM records, apply fetch_func's rules (src/indelminer.c:339-615),
 *           collect candidate reads, CIGAR-derived and paired-read evidence in arrival
 *           order, and record every READCHUNK flush point with its marker (617-623)
 *   GPU     one im_realign_batch over the contig's candidates
 *   pass B  replay: evidence enters the pending list in arrival order; at every flush
 *           point process_evidence -> sort -> merge -> print exactly as the reference
 */
#define _GNU_SOURCE            /* fopencookie */
#define _POSIX_C_SOURCE 200809L
#include "imhost.h"

#include <ctype.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <dirent.h>
#include <sys/wait.h>
#include <setjmp.h>
#include <spawn.h>

#define INDELMINER_VERSION 0.2          /* src/indelminer.c:26 */

#define OP_M 0
#define OP_I 1
#define OP_D 2
#define OP_N 3
#define OP_S 4
#define OP_H 5
#define OP_P 6
#define OP_EQ 7
#define OP_X 8
#define CIG_OP(c)  ((int)((c) & 15u))
#define CIG_LEN(c) ((int)((c) >> 4))

static im_options O;
static time_t t0;

static void out_flush_on_exit(void);
static void walker_bails_out(void);
static void mg_rank_failed(void);
static volatile int g_spec_active;      /* the walk is being staged on a provisional insert-length table (run_pipeline) */
/* ONE thread ends the process -- with a message, or by handing the run to a child and waiting for it; a second thread that gets
 * there (two walkers meeting the same unknown read group while the first one's fall-back child is already running) waits
 * for ever, i.e. until the first one's exit.  Found by pipeline_soak.py 94036: a walker's message and exit(1) under a running child. */
static pthread_mutex_t g_end_mu = PTHREAD_MUTEX_INITIALIZER;
static __thread int t_holds_end;
static void end_lock(void) { if (!t_holds_end) { pthread_mutex_lock(&g_end_mu); t_holds_end = 1; } }
static void spec_fallback(const char* why);
/* The record-at-a-time path reads a contig's records first (pass A: dispatch, candidates, flush points) and clusters / prints
 * behind that (pass B).  The reference prints every READCHUNK flush as it goes, so a record it dies on has the flushes in front of
 * it on stdout: a death in pass A waits -- the message is kept, the pass ends at that record, pass B prints the flushes that lie
 * in front of it, then the run ends with the message and the status (run_contig). */
static jmp_buf g_passA_jmp;
static int g_passA_armed = 0;
static char g_passA_msg[1024];
static __thread int t_is_main_thread_of_passA = 0;
static void passA_defer(const char* fmt, va_list ap)
{
    vsnprintf(g_passA_msg, sizeof g_passA_msg, fmt, ap);
    g_passA_armed = 0;
    longjmp(g_passA_jmp, 1);
}
static void fatalf(const char* fmt, ...)
{
    /* src/errors.c:15-27: message on stderr, exit(1) */
    va_list ap;
    if (g_passA_armed && t_is_main_thread_of_passA) { va_start(ap, fmt); passA_defer(fmt, ap); }
    if (g_spec_active) spec_fallback(fmt);      /* nothing is out yet: the run without the speculation finds out what is wrong, if anything is */
    walker_bails_out();         /* a walker thread of the pipeline does not come back from this (see handoff_to_host_child) */
    mg_rank_failed();           /* multi-GPU: rank 0 stops waiting for this rank's output */
    end_lock();
    va_start(ap, fmt);
    out_flush_on_exit();
    fflush(stdout);
    fprintf(stderr, "indelminer: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    exit(EXIT_FAILURE);
}
/* Everything the driver prints as output goes through printf.  A replay worker (run_pipeline) points its own t_out at a
 * memory buffer, so that several groups can be replayed at once and still come out in contig order; every other thread
 * prints to stdout. */
static __thread FILE* t_out;
#define OUT (t_out ? t_out : stdout)
#define printf(...) fprintf(OUT, __VA_ARGS__)
static pthread_mutex_t g_query_mu = PTHREAD_MUTEX_INITIALIZER;     /* depth queries share the context's workspace and stream */

/* Runs the reference aborts.  The device pipeline finds the record the reference would die on during the walk (or in the device
 * stage), when only the groups in front of it have been printed; the reference has by then also printed the flushes of that
 * group in front of the record.  Rather than unpick a half-walked group, the pipeline hands the run over: it lets the groups in
 * front go out, counts the bytes it has printed (the main thread prints through a counting stream), and starts this program
 * again as a child in its record-at-a-time mode, told to drop that many bytes of its output.  The child prints the rest exactly
 * as the reference does and dies at the record with the reference's message and status; the parent exits with its status. */
static int64_t g_out_bytes = 0;             /* parent: bytes the main thread has written to stdout */
static int64_t g_out_skip = 0;              /* child: bytes of its output still to drop */
static int g_real_stderr = -1;              /* child: stderr is silent until it has something new to say */
static char** g_argv = NULL;
static __thread int t_is_main = 0;
static __thread jmp_buf* t_abort_jmp = NULL;    /* a walker thread's way out of a walk that met such a record */
static ssize_t out_cookie_write(void* c, const char* buf, size_t n)
{
    (void)c;
    size_t at = 0;
    if (g_out_skip > 0) { at = (size_t)g_out_skip < n ? (size_t)g_out_skip : n; g_out_skip -= (int64_t)at; }
    while (at < n) { const ssize_t w = write(STDOUT_FILENO, buf + at, n - at); if (w <= 0) return 0; at += (size_t)w; g_out_bytes += w; }
    return (ssize_t)n;
}
static FILE* out_cookie_open(void)
{
    cookie_io_functions_t io = { NULL, out_cookie_write, NULL, NULL };
    FILE* f = fopencookie(NULL, "w", io);
    if (f) setvbuf(f, NULL, _IOFBF, 1 << 16);
    return f;
}
/* Whatever ends the run inside a walker's walk -- the triage's error classes at harvest, the host's own checks on discordant
 * mates and read groups as the records go by -- may not be the FIRST thing the reference dies of (errors are found chunk by
 * chunk, contigs in parallel): the walker leaves the walk, and the record-at-a-time child finds the first in record order. */
static int g_main_in_walk = 0;              /* annotate mode walks on the main thread: the same, without the jump */
static void pipeline_handoff(void);
static void walker_bails_out(void)
{
    if (t_abort_jmp) { jmp_buf* j = t_abort_jmp; t_abort_jmp = NULL; longjmp(*j, 1); }
    if (t_is_main && g_main_in_walk) { g_main_in_walk = 0; pipeline_handoff(); }
}
static void out_flush_on_exit(void)
{
    if (t_out) fflush(t_out);
    if (g_real_stderr >= 0) { fflush(stderr); dup2(g_real_stderr, STDERR_FILENO); g_real_stderr = -1; }
}
extern char** environ;
static void spawn_self_and_exit(const char* mode);
static void handoff_to_host_child(void) { spawn_self_and_exit("INDELMINER_PIPELINE=host"); }
/* The one-pass run staged its groups on an insert-length table made from the first pieces, and the whole file says otherwise (or
 * something else went wrong on the way): nothing is out, the header waits for the table (g_header_held); the same program takes the run again with the pre-pass. */
static void spec_fallback(const char* why)
{
    static pthread_mutex_t once = PTHREAD_MUTEX_INITIALIZER;
    pthread_mutex_lock(&once);                  /* the first caller takes the process with it; a second one waits for that */
    g_spec_active = 0;
    if (getenv("INDELMINER_TIMING") || getenv("INDELMINER_DEBUG_HANDOFF")) fprintf(stderr, "[one pass] provisional insert lengths did not hold (%s): the run is taken again with the pre-pass\n", why);
    spawn_self_and_exit("INDELMINER_ONEPASS=0");
}
static void spawn_self_and_exit(const char* mode)
{
    const int host_mode = strncmp(mode, "INDELMINER_PIPELINE=", 20) == 0;
    end_lock();
    if (t_out) fflush(t_out);
    fflush(stdout);
    /* the child's environment is a private copy: other threads (walkers, replay workers) may be inside getenv, and setenv
     * moves the block they read */
    char skip[64], from[64];
    snprintf(skip, sizeof skip, "INDELMINER_SKIP_STDOUT=%lld", (long long)g_out_bytes);
    snprintf(from, sizeof from, "INDELMINER_HANDOFF_PARENT=%ld", (long)getpid());
    size_t n_env = 0;
    while (environ[n_env]) n_env++;
    char** envp = malloc(sizeof(char*) * (n_env + 5));
    if (!envp) _exit(EXIT_FAILURE);
    size_t k = 0;
    for (size_t i = 0; i < n_env; i++)
        if (strncmp(environ[i], host_mode ? "INDELMINER_PIPELINE=" : "INDELMINER_ONEPASS=", host_mode ? 20 : 19) != 0 && strncmp(environ[i], "INDELMINER_SKIP_STDOUT=", 23) != 0 &&
            strncmp(environ[i], "INDELMINER_HANDOFF_PARENT=", 26) != 0 && strncmp(environ[i], "INDELMINER_HANDOFF_QUIET=", 25) != 0) envp[k++] = environ[i];
    envp[k++] = (char*)mode;
    envp[k++] = skip;
    envp[k++] = from;
    if (host_mode) envp[k++] = (char*)"INDELMINER_HANDOFF_QUIET=1";       /* what that child has to say on stderr has been said (until it dies of something) */
    envp[k] = NULL;
    pid_t pid;
    if (getenv("INDELMINER_DEBUG_HANDOFF")) fprintf(stderr, "[handoff] %lld bytes printed, starting the child\n", (long long)g_out_bytes);
    if (posix_spawn(&pid, "/proc/self/exe", NULL, NULL, g_argv, envp) != 0) { fprintf(stderr, "indelminer: cannot start the record-at-a-time run\n"); _exit(EXIT_FAILURE); }
    int status = 0;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) { }
    if (getenv("INDELMINER_DEBUG_HANDOFF")) fprintf(stderr, "[handoff] child status 0x%x\n", status);
    _exit(WIFEXITED(status) ? WEXITSTATUS(status) : EXIT_FAILURE);
}

static void passA_assert(const char* fmt, ...) { va_list ap; va_start(ap, fmt); passA_defer(fmt, ap); }
#define forceassert(e) do { if (!(e)) { if (g_passA_armed && t_is_main_thread_of_passA) passA_assert("\x01" "Assertion failed: %s file %s line %d", #e, __FILE__, __LINE__); \
                                        walker_bails_out(); end_lock(); out_flush_on_exit(); fprintf(stderr, "Assertion failed: %s file %s line %d\n", #e, __FILE__, __LINE__); exit(EXIT_FAILURE); } } while (0)

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int g_timing = 0;
static double g_t_last = 0;
static void phase_time(const char* what)
{
    if (!g_timing) return;
    const double t = now_ms();
    fprintf(stderr, "[timing] %-34s %9.2f ms\n", what, t - g_t_last);
    g_t_last = t;
}

/* INDELMINER_TIMING: processor seconds by kind of thread (on a box with as many threads as cores the run is as long as their sum) */
static int64_t g_cpu_walk_ns, g_cpu_walk_dev_ns, g_cpu_replay_ns, g_wall_walk_throttled_ns, g_wall_walk_ns, g_wall_walk_dev_ns;
static __thread int64_t t_cpu_dev_ns, t_wall_dev_ns;      /* a walker's processor time inside device calls (uploads, launches, waits) */
#define DEV_TIMED(call) do { if (g_timing) { const int64_t t0_ = thread_cpu_ns(), w0_ = wall_ns(); call; t_cpu_dev_ns += thread_cpu_ns() - t0_; t_wall_dev_ns += wall_ns() - w0_; } else { call; } } while (0)
static int64_t thread_cpu_ns(void) { struct timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec; }
static int64_t wall_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec; }
static int64_t g_sw_ns, g_sw_calls, g_sw_tasks, g_sw_cells;      /* annotate mode: im_support_batch calls, their tasks and DP cells */
static void cpu_report(void)
{
    if (!g_timing) return;
    if (g_sw_calls) fprintf(stderr, "[timing] annotate mode: %ld im_support_batch calls, %ld tasks, %ld cells, %.3f s in the calls (launch + copies + kernel)\n",
                            (long)g_sw_calls, (long)g_sw_tasks, (long)g_sw_cells, g_sw_ns / 1e9);
    struct timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    {   /* what the kernel will have to take apart when the process ends */
        FILE* fp = fopen("/proc/self/status", "r");
        char line[256]; long rss = 0, hwm = 0, anon = 0, file = 0, shm = 0;
        while (fp && fgets(line, sizeof line, fp)) {
            if (sscanf(line, "VmRSS: %ld", &rss) == 1 || sscanf(line, "VmHWM: %ld", &hwm) == 1 || sscanf(line, "RssAnon: %ld", &anon) == 1 ||
                sscanf(line, "RssFile: %ld", &file) == 1 || sscanf(line, "RssShmem: %ld", &shm) == 1) continue;
        }
        if (fp) fclose(fp);
        struct rusage ru; memset(&ru, 0, sizeof ru); getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[timing] resident memory at the end: %.2f GB (anonymous %.2f, file %.2f, shared %.2f), peak %.2f GB; %ld page faults, %.2f s of the processor time in the kernel\n",
                rss / 1048576.0, anon / 1048576.0, file / 1048576.0, shm / 1048576.0, hwm / 1048576.0, ru.ru_minflt, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec / 1e6);
    }
    if (getenv("INDELMINER_TIMING_MAPS")) {
        /* the anonymous mappings by size: how many, how much of them resident, how much of that in huge pages */
        FILE* fp = fopen("/proc/self/smaps", "r");
        char line[512]; long rss = 0, ahp = 0, size = 0; int anon = 0;
        struct { long size_mb, n, rss, ahp; } cls[64]; int nc = 0;
        memset(cls, 0, sizeof cls);
        while (fp && fgets(line, sizeof line, fp)) {
            long v; unsigned long lo, hi; char perms[8]; unsigned long off, ino; int maj, min_, nn = 0;
            if (sscanf(line, "%lx-%lx %7s %lx %x:%x %lu %n", &lo, &hi, perms, &off, &maj, &min_, &ino, &nn) >= 7) { anon = ino == 0 && (line[nn] == 0 || line[nn] == '\n' || line[nn] == '['); continue; }
            if (sscanf(line, "Size: %ld", &v) == 1) size = v;
            else if (sscanf(line, "Rss: %ld", &v) == 1) rss = v;
            else if (sscanf(line, "AnonHugePages: %ld", &v) == 1) ahp = v;
            else if (strncmp(line, "VmFlags:", 8) == 0 && anon && rss >= 16 * 1024) {
                const long mb = size / 1024;
                int k = 0; while (k < nc && cls[k].size_mb != mb) k++;
                if (k == nc && nc < 64) { cls[nc].size_mb = mb; nc++; }
                if (k < 64) { cls[k].n++; cls[k].rss += rss; cls[k].ahp += ahp; }
            }
        }
        if (fp) fclose(fp);
        for (int i = 0; i < nc; i++) fprintf(stderr, "[maps] %3ld anonymous mappings of %6ld MB: resident %8.1f MB, in huge pages %8.1f MB\n", cls[i].n, cls[i].size_mb, cls[i].rss / 1024.0, cls[i].ahp / 1024.0);
    }
    fprintf(stderr, "[timing] walkers: %.2f s of wall time between their start and their last piece, %.2f of them inside device calls\n",
            __atomic_load_n(&g_wall_walk_ns, __ATOMIC_RELAXED) / 1e9, __atomic_load_n(&g_wall_walk_dev_ns, __ATOMIC_RELAXED) / 1e9);
    fprintf(stderr, "[timing] processor seconds: walkers %.2f (%.2f of them in device calls; + %.2f s held back behind the main thread), replay workers %.3f, main thread %.2f, whole process %.2f\n",
            __atomic_load_n(&g_cpu_walk_ns, __ATOMIC_RELAXED) / 1e9, __atomic_load_n(&g_cpu_walk_dev_ns, __ATOMIC_RELAXED) / 1e9, __atomic_load_n(&g_wall_walk_throttled_ns, __ATOMIC_RELAXED) / 1e9,
            __atomic_load_n(&g_cpu_replay_ns, __ATOMIC_RELAXED) / 1e9, thread_cpu_ns() / 1e9, ts.tv_sec + ts.tv_nsec / 1e9);
}

static void timestamp(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, " : %ld sec. elapsed\n", (long)(time(0) - t0));
    va_end(ap);
}

/* Large blocks ask for transparent huge pages (the boxes run THP in `madvise` mode): a whole-genome run takes tens of gigabytes of
 * fresh pages through its groups' logs and candidate stores, and a first touch costs 130 ns per 4 KB page against 80 us per 2 MB
 * one (8 GB: 1.06 s against 0.31 s, and 0.58 s against 0.30 s to give them back; profiles/r03_thp_exit.log). */
static int g_thp = -1;
static inline void* want_huge_pages(void* p, size_t n)
{
    if (n < ((size_t)4 << 20) || !p) return p;
    if (g_thp < 0) { const char* e = getenv("INDELMINER_THP"); g_thp = !(e && e[0] == '0'); }
    if (g_thp) {
        const uintptr_t a = (uintptr_t)p & ~(uintptr_t)4095, z = ((uintptr_t)p + n) & ~(uintptr_t)4095;
        if (z > a) (void)madvise((void*)a, (size_t)(z - a), MADV_HUGEPAGE);
    }
    return p;
}
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return want_huge_pages(p, n); }
static void* xcalloc(size_t n, size_t s) { void* p = calloc(n ? n : 1, s ? s : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return p; }
static void* xrealloc(void* p, size_t n) { p = realloc(p, n ? n : 1); if (!p) { fprintf(stderr, "out of memory\n"); exit(2); } return want_huge_pages(p, n); }
static char* xstrdup(const char* s) { size_t l = strlen(s); char* d = xmalloc(l + 1); memcpy(d, s, l + 1); return d; }

/* The parts, in dependency order (each one only uses what stands above it): */
#include "host_logic.c"         /* the reference's host-side logic, restated */
#include "host_setup.c"         /* GPU start-up, config / estimates, coverage table, record-at-a-time path */
#include "host_pipeline.c"      /* device pipeline: chunks, groups, stage, replay of a group */
#include "host_multirank.c"     /* one process per GPU */
#include "host_walk.c"          /* pieces, walker pool, run_pipeline */
#include "host_main.c"          /* CLI */
