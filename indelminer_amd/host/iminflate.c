/*
 * iminflate.c -- raw DEFLATE (RFC 1951) decoder for BGZF blocks, written for this driver.
 *
 * Once realignment and clustering run on the GPU the host's share of a run is the BGZF inflate (north_star keeps BAM decode
 * on the host): the walkers spend their time in it.  A BGZF block is a complete deflate stream of at most 64 KiB whose
 * input and output buffers are both in memory, which allows what a streaming inflate cannot do: a 64-bit bit buffer refilled
 * eight bytes at a time without bounds checks (the block's trailer and spare bytes follow the payload), decode tables of
 * 2^11 litlen / 2^8 distance entries that resolve most symbols in one look-up with their extra bits taken from the same
 * refill, word-wise match copies into an output buffer with slack, and no state to allocate or reset per block.
 *
 *   im_inflate(in, in_len, out, out_cap) -> decoded bytes, or -1 (corrupt stream / does not fit)
 *   REQUIRES: IM_INFLATE_SLACK readable bytes behind in[in_len) and as many writable bytes behind out[out_cap).
 */
#include "iminflate.h"

#include <stdlib.h>
#include <string.h>

#define LL_BITS 11
#define OF_BITS 8
#define LL_SIZE 4096            /* main table + sub-tables (2^11 + at most ~1400) */
#define OF_SIZE 1024

/* table entries
 *   litlen:  bits 0-7 ALL the bits the symbol takes, codeword + extra bits | 8-11 the codeword's share of them | 12-15 extra bits
 *            | 16-28 literal / length base | 31 literal | 29 end of block
 *            sub-table pointer (30): bits 0-7 the main table's bits | 8-15 sub-table bits | 16-28 sub-table start
 *   dist:    the same fields | 16-30 distance base / sub-table start | 31 sub-table
 *            an error entry (codes 30, 31; the unused half of a one-code table) has bit 31 and the start 0x7fff
 * One shift per symbol: the extra bits are read out of a copy of the bit buffer -- (saved & low `all` bits) >> codeword bits, a
 * BZHI and a SHRX where the processor has them (im_inflate picks that instance at run time) -- off the chain of dependent
 * operations that runs from one symbol's table entry through the bit buffer to the next look-up.  The loop is bound by the number
 * of instructions per match, not by that chain: entries are laid out so that a match needs no masking of flags (they are zero
 * in a length / distance entry) and no validity checks (every slot of an accepted code's tables is filled; what is invalid
 * carries a flag of the rare path). */
#define E_LIT  0x80000000u
#define E_SUB  0x40000000u
#define E_EOB  0x20000000u
#define D_SUB  0x80000000u

static const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
static const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
static const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
static const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
static const uint8_t kPreOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };

typedef struct {
    uint32_t ll[LL_SIZE];
    uint32_t of[OF_SIZE];
    uint32_t pre[128];
    uint8_t  lens[288 + 32 + 140];
    uint16_t sorted[288];
} tables_t;

static inline uint32_t ll_entry(int sym)
{
    if (sym < 256) return E_LIT | ((uint32_t)sym << 16);
    if (sym == 256) return E_EOB;
    if (sym > 285) return E_EOB | (1u << 16);          /* 286, 287: not valid in a stream; decoding one is an error */
    return ((uint32_t)kLenBase[sym - 257] << 16) | ((uint32_t)kLenExtra[sym - 257] << 12) | kLenExtra[sym - 257];
}
static uint32_t kLitlenEntry[288];
static void litlen_entries_init(void) { for (int s = 0; s < 288; s++) kLitlenEntry[s] = ll_entry(s); }
static inline uint32_t of_entry(int sym)
{
    if (sym > 29) return D_SUB | (0x7fffu << 16);       /* 30, 31: an error when met (found on the sub-table path) */
    return ((uint32_t)kDistBase[sym] << 16) | ((uint32_t)kDistExtra[sym] << 12) | kDistExtra[sym];
}

/* canonical Huffman code -> look-up table with sub-tables; kind: 0 precode, 1 litlen, 2 distance.  Returns 0 / -1. */
static int build(uint32_t* table, int table_size, int table_bits, const uint8_t* lens, int nsyms, int kind, uint16_t* sorted)
{
    int count[16] = { 0 }, offs[16];
    for (int s = 0; s < nsyms; s++) count[lens[s]]++;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= 15; l++) { left = (left << 1) - count[l]; if (left < 0) return -1; if (count[l]) maxlen = l; }
    const int nused = nsyms - count[0];
    if (nused == 0) {
        /* no codes at all: legal for the distance code of a block of literals only */
        for (int i = 0; i < (1 << table_bits); i++) table[i] = kind == 2 ? (D_SUB | (0x7fffu << 16) | 1u) : (E_EOB | (1u << 16) | 1u);
        return 0;
    }
    if (left > 0 && (kind == 0 || !(nused == 1 && count[1] == 1))) return -1;     /* incomplete: only a litlen / distance code of ONE 1-bit codeword is allowed */
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < nsyms; s++) if (lens[s]) sorted[offs[lens[s]]++] = (uint16_t)s;
    /* the unused half of a one-code table decodes to an error entry */
    if (left > 0) for (int i = 0; i < (1 << table_bits); i++) table[i] = kind == 2 ? (D_SUB | (0x7fffu << 16) | 1u) : (E_EOB | (1u << 16) | 1u);
    uint32_t code = 0;              /* bit-reversed codeword */
    int at = 0, sub_next = 1 << table_bits, sub_prefix = -1, sub_start = 0, sub_bits = 0;
    /* The main table grows by doubling: while the codes of length len are entered it is 2^len entries long (a codeword IS its
     * index there), and before the next length it is copied behind itself -- every entry reaches its 2^(bits - len) places
     * by block copies instead of by strided stores. */
    int tb = 0;                     /* the main table is 2^tb entries long so far; 0 = nothing entered yet */
    for (int len = 1; len <= maxlen; len++) {
        if (count[len] && len <= table_bits) {
            if (tb == 0) tb = len;
            for (; tb < len; tb++) memcpy(table + (1u << tb), table, sizeof(uint32_t) << tb);
        } else if (count[len] && tb < table_bits) {
            if (tb == 0) tb = table_bits;       /* every code is longer than the main table: its entries are all sub-table pointers */
            for (; tb < table_bits; tb++) memcpy(table + (1u << tb), table, sizeof(uint32_t) << tb);
        }
        for (int c = 0; c < count[len]; c++, at++) {
            const int sym = sorted[at];
            const uint32_t e = kind == 0 ? ((uint32_t)sym << 16) : kind == 1 ? kLitlenEntry[sym] : of_entry(sym);
            if (len <= table_bits) {
                table[code] = e + (uint32_t)len * 257u;
            } else {
                const int prefix = (int)(code & ((1u << table_bits) - 1u));
                if (prefix != sub_prefix) {
                    sub_prefix = prefix; sub_start = sub_next;
                    sub_bits = len - table_bits;
                    int used = count[len] - c;
                    while (used < (1 << sub_bits)) { sub_bits++; if (table_bits + sub_bits > 15) break; used = (used << 1) + count[table_bits + sub_bits]; }
                    if (sub_start + (1 << sub_bits) > table_size) return -1;
                    sub_next = sub_start + (1 << sub_bits);
                    table[prefix] = (kind == 2 ? D_SUB : E_SUB) | ((uint32_t)sub_start << 16) | ((uint32_t)sub_bits << 8) | (uint32_t)table_bits;
                }
                const int rest = len - table_bits;
                for (uint32_t i = code >> table_bits; i < (1u << sub_bits); i += 1u << rest) table[sub_start + (int)i] = e + (uint32_t)rest * 257u;
            }
            /* next codeword, in reversed bits */
            uint32_t bit = 1u << (len - 1);
            while (code & bit) bit >>= 1;
            code = bit ? (code & (bit - 1u)) | bit : 0;
        }
        /* the codes get one bit longer: in reversed form the new bit is the most significant one, and it is 0 already */
    }
    if (tb == 0) tb = table_bits;
    for (; tb < table_bits; tb++) memcpy(table + (1u << tb), table, sizeof(uint32_t) << tb);
    return 0;
}

static inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* the extra bits of the symbol whose entry is ent, out of the bit buffer as it stood in front of the symbol */
#define EXTRA_BITS(saved, ent) (((uint32_t)(saved) & ((1u << ((ent) & 31u)) - 1u)) >> (((ent) >> 8) & 15u))     /* a symbol takes at most 15 + 13 bits */

static inline __attribute__((always_inline)) int64_t inflate_stream(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap)
{
    tables_t T;
    const uint8_t* ip = in;
    const uint8_t* const in_end = in + in_len;
    uint8_t* op = out;
    uint8_t* const out_end = out + out_cap;
    uint64_t bb = 0;            /* bit buffer: the next bits of the stream in its low bits */
    unsigned bc = 0;            /* valid bits in bb */
    /* refill to at least 56 bits: reads 8 bytes at ip, which a valid stream never takes more than 8 bytes behind in_end (checked
     * before every refill of the loops below), i.e. at most IM_INFLATE_SLACK bytes behind the input are touched */
#define REFILL() do { bb |= load64(ip) << bc; ip += (63u - bc) >> 3; bc |= 56u; } while (0)
#define TAKE(n) do { bb >>= (n); bc -= (n); } while (0)
    int last;
    do {
        if (ip > in_end + 8) return -1;
        REFILL();
        last = (int)(bb & 1u);
        const unsigned type = (unsigned)(bb >> 1) & 3u;
        TAKE(3);
        if (type == 0) {
            /* stored: to the byte boundary, LEN, NLEN, bytes */
            TAKE(bc & 7u);
            /* give whole unread bytes back */
            ip -= bc >> 3; bb = 0; bc = 0;
            if (in_end - ip < 4) return -1;
            const unsigned len = ip[0] | ((unsigned)ip[1] << 8), nlen = ip[2] | ((unsigned)ip[3] << 8);
            ip += 4;
            if ((len ^ nlen) != 0xffffu || (size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) return -1;
            memcpy(op, ip, len); op += len; ip += len;
            continue;
        }
        if (type == 3) return -1;
        if (type == 1) {
            for (int i = 0; i < 144; i++) T.lens[i] = 8;
            for (int i = 144; i < 256; i++) T.lens[i] = 9;
            for (int i = 256; i < 280; i++) T.lens[i] = 7;
            for (int i = 280; i < 288; i++) T.lens[i] = 8;
            for (int i = 0; i < 32; i++) T.lens[288 + i] = 5;
            if (build(T.ll, LL_SIZE, LL_BITS, T.lens, 288, 1, T.sorted) || build(T.of, OF_SIZE, OF_BITS, T.lens + 288, 32, 2, T.sorted)) return -1;
        } else {
            const unsigned hlit = (unsigned)(bb & 31u) + 257u, hdist = (unsigned)((bb >> 5) & 31u) + 1u, hclen = (unsigned)((bb >> 10) & 15u) + 4u;
            TAKE(14);
            if (hlit > 286u || hdist > 30u) return -1;
            uint8_t pl[19] = { 0 };
            for (unsigned i = 0; i < hclen; i++) { if (bc < 3) { if (ip > in_end + 8) return -1; REFILL(); } pl[kPreOrder[i]] = (uint8_t)(bb & 7u); TAKE(3); }
            if (build(T.pre, 128, 7, pl, 19, 0, T.sorted)) return -1;
            unsigned n = 0;
            while (n < hlit + hdist) {
                if (ip > in_end + 8) return -1;
                REFILL();
                const uint32_t e = T.pre[bb & 127u];
                const unsigned l = e & 255u, sym = e >> 16;
                if (l == 0) return -1;
                TAKE(l);
                if (sym < 16) { T.lens[n++] = (uint8_t)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) { if (n == 0) return -1; val = T.lens[n - 1]; rep = 3u + (unsigned)(bb & 3u); TAKE(2); }
                else if (sym == 17) { rep = 3u + (unsigned)(bb & 7u); TAKE(3); }
                else { rep = 11u + (unsigned)(bb & 127u); TAKE(7); }
                if (n + rep > hlit + hdist) return -1;
                memset(T.lens + n, (int)val, rep); n += rep;
            }
            if (T.lens[256] == 0) return -1;
            /* the two codes apart: distance lengths behind 288 litlen slots */
            uint8_t dl[32];
            memcpy(dl, T.lens + hlit, hdist);
            memset(dl + hdist, 0, 32 - hdist);
            memset(T.lens + hlit, 0, 288 - hlit);
            if (build(T.ll, LL_SIZE, LL_BITS, T.lens, 288, 1, T.sorted) || build(T.of, OF_SIZE, OF_BITS, dl, 32, 2, T.sorted)) return -1;
        }
        /* ---- the block's symbols ---- */
        /* fast loop: while at least 3 literals + a longest match + its word-wise overrun fit the output and the input has not
         * run out, nothing is checked per symbol; the careful loop below finishes the block */
        uint8_t* const out_fast = out_cap > 320 ? out_end - 320 : out;
        int done = 0;
        /* The look-up of the NEXT symbol is issued before the current one's bytes are written (its index only needs the bit
         * buffer): the table load's latency hides behind the copy instead of heading the next iteration's chain. */
        if (op < out_fast && ip < in_end) {
            REFILL();
            uint32_t e = T.ll[bb & ((1u << LL_BITS) - 1u)];
            for (;;) {
                /* here: bc >= 56 - 33 when coming from literals, >= 56 otherwise; e is the entry of the bits at hand */
                if (e & E_LIT) {
                    /* literals come in runs: up to three from one refill (11 bits each at most straight from the main table) */
                    *op++ = (uint8_t)(e >> 16); TAKE(e & 255u);
                    e = T.ll[bb & ((1u << LL_BITS) - 1u)];
                    if (e & E_LIT) {
                        *op++ = (uint8_t)(e >> 16); TAKE(e & 255u);
                        e = T.ll[bb & ((1u << LL_BITS) - 1u)];
                        if (e & E_LIT) { *op++ = (uint8_t)(e >> 16); TAKE(e & 255u); e = T.ll[bb & ((1u << LL_BITS) - 1u)]; }
                    }
                    /* <= 33 bits are gone, >= 23 are left: the look-up above saw its 11 bits; what follows may need 48 */
                    if (!(op < out_fast && ip < in_end)) break;
                    REFILL();
                    continue;
                }
                if (e & (E_SUB | E_EOB)) {
                    if (e & E_EOB) { if ((e >> 16) & 1u) return -1; TAKE(e & 255u); done = 1; break; }
                    TAKE(LL_BITS);
                    e = T.ll[((e >> 16) & 0x1fffu) + (uint32_t)(bb & ((1u << ((e >> 8) & 255u)) - 1u))];
                    if (e & E_LIT) {
                        *op++ = (uint8_t)(e >> 16); TAKE(e & 255u);
                        if (!(op < out_fast && ip < in_end)) break;
                        REFILL();
                        e = T.ll[bb & ((1u << LL_BITS) - 1u)];
                        continue;
                    }
                    if (e & E_EOB) { if ((e >> 16) & 1u) return -1; TAKE(e & 255u); done = 1; break; }
                }
                /* a length: no flags in e */
                const uint64_t bl = bb;
                TAKE(e & 255u);
                const unsigned length = (e >> 16) + (unsigned)EXTRA_BITS(bl, e);
                uint32_t d = T.of[bb & ((1u << OF_BITS) - 1u)];
                if (d & D_SUB) {
                    if (((d >> 16) & 0x7fffu) == 0x7fffu) return -1;
                    TAKE(OF_BITS);
                    d = T.of[((d >> 16) & 0x7fffu) + (uint32_t)(bb & ((1u << ((d >> 8) & 255u)) - 1u))];
                    if (d & D_SUB) return -1;
                }
                const uint64_t bd = bb;
                TAKE(d & 255u);
                const unsigned dist = (d >> 16) + (unsigned)EXTRA_BITS(bd, d);
                if (dist > (size_t)(op - out)) return -1;
                const uint8_t* src = op - dist;
                uint8_t* dst = op;
                op += length;
                /* the next symbol's entry: on its way while the match is copied */
                const int more = op < out_fast && ip < in_end;
                if (more) {
                    /* with 11 bits still at hand the look-up does not wait for the refill's load (the usual case: a match takes ~20) */
                    if (bc >= LL_BITS) { e = T.ll[bb & ((1u << LL_BITS) - 1u)]; REFILL(); }
                    else { REFILL(); e = T.ll[bb & ((1u << LL_BITS) - 1u)]; }
                }
                if (dist >= 16) {
                    /* 32 bytes whatever the length (most matches are shorter: no branch on it); the second piece may read what the
                     * first has just written */
                    memcpy(dst, src, 16); memcpy(dst + 16, src + 16, 16);
                    if (length > 32) { dst += 32; src += 32; do { memcpy(dst, src, 16); dst += 16; src += 16; } while (dst < op); }
                } else if (dist >= 8) {
                    memcpy(dst, src, 8); memcpy(dst + 8, src + 8, 8);
                    if (length > 16) { dst += 16; src += 16; do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < op); }
                } else if (dist == 1) {
                    memset(dst, *src, length);
                } else {
                    do { *dst++ = *src++; } while (dst < op);
                }
                if (!more) break;
            }
        }
        while (!done) {
            if (ip > in_end + 8) return -1;
            REFILL();
            uint32_t e = T.ll[bb & ((1u << LL_BITS) - 1u)];
            if (e & E_SUB) { TAKE(LL_BITS); e = T.ll[((e >> 16) & 0x1fffu) + (uint32_t)(bb & ((1u << ((e >> 8) & 255u)) - 1u))]; }
            if (e & E_LIT) {
                if (op >= out_end) return -1;
                *op++ = (uint8_t)(e >> 16);
                TAKE(e & 255u);
                continue;
            }
            if (e & E_EOB) { if ((e >> 16) & 1u) return -1; TAKE(e & 255u); break; }
            const uint64_t bl = bb;
            TAKE(e & 255u);
            const unsigned length = (e >> 16) + (unsigned)EXTRA_BITS(bl, e);
            /* <= 15 + 5 bits gone of >= 56: the distance code and its extra bits (15 + 13) fit what is left */
            uint32_t d = T.of[bb & ((1u << OF_BITS) - 1u)];
            if (d & D_SUB) {
                if (((d >> 16) & 0x7fffu) == 0x7fffu) return -1;
                TAKE(OF_BITS);
                d = T.of[((d >> 16) & 0x7fffu) + (uint32_t)(bb & ((1u << ((d >> 8) & 255u)) - 1u))];
                if (d & D_SUB) return -1;
            }
            const uint64_t bd = bb;
            TAKE(d & 255u);
            const unsigned dist = (d >> 16) + (unsigned)EXTRA_BITS(bd, d);
            if (dist > (size_t)(op - out) || (size_t)(out_end - op) < length) return -1;
            const uint8_t* src = op - dist;
            uint8_t* dst = op;
            op += length;
            if (dist >= 8) {
                /* word-wise; runs up to 7 bytes past the match into the slack (or bytes that are written next anyway) */
                do { memcpy(dst, src, 8); dst += 8; src += 8; } while (dst < op);
            } else if (dist == 1) {
                memset(dst, *src, length);
            } else {
                do { *dst++ = *src++; } while (dst < op);
            }
        }
    } while (!last);
    /* the bytes the last refill took beyond the stream's end are not the stream's: the stream may not have run past its input */
    if (ip - (bc >> 3) > in_end) return -1;
    return (int64_t)(op - out);
#undef REFILL
#undef TAKE
}

static int64_t inflate_plain(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) { return inflate_stream(in, in_len, out, out_cap); }
__attribute__((target("bmi,bmi2"))) static int64_t inflate_bmi2(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap) { return inflate_stream(in, in_len, out, out_cap); }

int64_t im_inflate(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap)
{
    static int64_t (*fn)(const uint8_t*, size_t, uint8_t*, size_t);
    int64_t (*f)(const uint8_t*, size_t, uint8_t*, size_t) = __atomic_load_n(&fn, __ATOMIC_ACQUIRE);
    if (!f) {
        litlen_entries_init();          /* the same values from whichever thread comes first */
        __builtin_cpu_init();
        const char* plain = getenv("INDELMINER_INFLATE_PLAIN");       /* the instance for processors without BMI2, for the tests */
        f = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi") && !(plain && plain[0] == '1') ? inflate_bmi2 : inflate_plain;
        __atomic_store_n(&fn, f, __ATOMIC_RELEASE);
    }
    return f(in, in_len, out, out_cap);
}
