/* inflate_fuzz.c -- indelminer_amd/host/iminflate.c against zlib (test harness; tests/test_inflate.py builds it with
 * -fsanitize=address,undefined and runs it).  Every buffer is allocated at exactly the size the decoder's contract allows it
 * to touch (IM_INFLATE_SLACK bytes behind input and output), so a byte too far is a sanitizer report.
 *   valid streams   : random / low-entropy / repetitive / BAM-like data, zlib levels 0-9, strategies default / fixed codes /
 *                     Huffman only / RLE, stored blocks, empty input -> identical bytes; output room of exactly the size -> ok;
 *                     one byte less -> -1
 *   truncated       : every prefix class of a valid stream -> -1
 *   corrupted       : random bit flips -> -1 or some output, never a byte outside the buffers */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "iminflate.h"

static uint64_t rng_state;
static uint32_t rnd(void) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(rng_state >> 33); }

static size_t make_data(uint8_t* d, size_t cap, int mode)
{
    size_t n = mode == 0 ? 0 : (rnd() % 8 == 0 ? rnd() % 40 : rnd() % cap);
    switch (mode % 6) {
    case 1: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)rnd(); break;                                   /* incompressible */
    case 2: for (size_t i = 0; i < n; i++) d[i] = (uint8_t)("ACGT"[rnd() & 3]); break;                     /* four symbols */
    case 3: { size_t i = 0; while (i < n) { const uint8_t c = (uint8_t)rnd(); size_t r = 1 + rnd() % 600; while (r-- && i < n) d[i++] = c; } } break;   /* runs */
    case 4: { uint8_t motif[97]; for (int k = 0; k < 97; k++) motif[k] = (uint8_t)rnd(); for (size_t i = 0; i < n; i++) d[i] = (rnd() % 50 == 0) ? (uint8_t)rnd() : motif[i % (1 + rnd() % 3 == 0 ? 97 : 31)]; } break;
    default: {                                                                                               /* BAM-like records */
        size_t i = 0; uint32_t pos = 1000;
        while (i + 220 < n) {
            pos += rnd() % 7; memcpy(d + i, &pos, 4); memset(d + i + 4, 0, 28); d[i + 8] = 12; i += 32;
            i += (size_t)snprintf((char*)d + i, 16, "r%u", rnd() % 1000000) + 1;
            for (int k = 0; k < 50; k++) d[i++] = (uint8_t)(((1u << (rnd() & 3)) << 4) | (1u << (rnd() & 3)));
            memset(d + i, 'I' - 33, 100); i += 100;
        }
        for (; i < n; i++) d[i] = 0;
    } break;
    }
    return n;
}

static size_t zdeflate(const uint8_t* in, size_t n, uint8_t* out, size_t cap, int level, int strategy)
{
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, strategy) != Z_OK) exit(3);
    z.next_in = (Bytef*)in; z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)cap;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) exit(4);
    const size_t m = z.total_out;
    deflateEnd(&z);
    return m;
}

/* run the decoder with buffers of exactly the allowed size */
static int64_t run(const uint8_t* comp, size_t clen, size_t out_cap, uint8_t** pout)
{
    uint8_t* in = malloc(clen + IM_INFLATE_SLACK);
    memcpy(in, comp, clen); memset(in + clen, 0xA5, IM_INFLATE_SLACK);
    uint8_t* out = malloc(out_cap + IM_INFLATE_SLACK);
    const int64_t r = im_inflate(in, clen, out, out_cap);
    free(in);
    if (pout) *pout = out; else free(out);
    return r;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    rng_state = argc > 2 ? (uint64_t)atoll(argv[2]) : 1;
    const size_t cap = 65536;
    uint8_t* data = malloc(cap + 16); uint8_t* comp = malloc(cap * 2 + 1024);
    long n_valid = 0, n_trunc = 0, n_corrupt = 0, n_corrupt_rejected = 0;
    const int strategies[4] = { Z_DEFAULT_STRATEGY, Z_FIXED, Z_HUFFMAN_ONLY, Z_RLE };
    for (int it = 0; it < rounds; it++) {
        const size_t n = make_data(data, cap, it % 7);
        const int level = (int)(rnd() % 10), strategy = strategies[rnd() % 4];
        const size_t clen = zdeflate(data, n, comp, cap * 2 + 1024, level, strategy);
        uint8_t* out = NULL;
        int64_t r = run(comp, clen, n + (rnd() % 3 == 0 ? 0 : rnd() % 500), &out);
        if (r != (int64_t)n || memcmp(out, data, n) != 0) { fprintf(stderr, "MISMATCH it %d n %zu clen %zu level %d strategy %d -> %lld\n", it, n, clen, level, strategy, (long long)r); return 1; }
        free(out); n_valid++;
        if (n > 0 && run(comp, clen, n - 1, NULL) != -1) { fprintf(stderr, "output one byte short accepted (it %d)\n", it); return 1; }
        for (int k = 0; k < 6 && clen > 0; k++) {
            const size_t cut = k == 0 ? clen - 1 : k == 1 ? 0 : rnd() % clen;
            if (run(comp, cut, n + 64, NULL) != -1) { fprintf(stderr, "truncated stream accepted (it %d, %zu of %zu)\n", it, cut, clen); return 1; }
            n_trunc++;
        }
        for (int k = 0; k < 12 && clen > 0; k++) {
            uint8_t* bad = malloc(clen);
            memcpy(bad, comp, clen);
            const int flips = 1 + (int)(rnd() % 4);
            for (int f = 0; f < flips; f++) bad[rnd() % clen] ^= (uint8_t)(1u << (rnd() & 7));
            const int64_t rr = run(bad, clen, n + (rnd() & 1 ? 0 : 300), NULL);
            if (rr < -1 || rr > (int64_t)n + 300) { fprintf(stderr, "corrupted stream: result %lld out of range\n", (long long)rr); return 1; }
            n_corrupt++; n_corrupt_rejected += rr == -1;
            free(bad);
        }
    }
    printf("{\"valid\": %ld, \"truncated\": %ld, \"corrupted\": %ld, \"corrupted_rejected\": %ld}\n", n_valid, n_trunc, n_corrupt, n_corrupt_rejected);
    return 0;
}
