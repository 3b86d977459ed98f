#!/usr/bin/env python3
"""One-off soak on the GPU box: the triage kernels against the oracle's fetch_func restatement on random BAM records --
random flags, positions, insert sizes, CIGARs (every op code, clips anywhere), base codes, aux areas (every tag type,
B arrays, RG / MQ of right and wrong types, junk tails), a few truncated records.
    python profiles/triage_fuzz.py [first_seed] [n_rounds]"""
import os, random, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from indelminer_amd import capi
from tests import test_gpu_triage as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
NAMES = ["lib1", "lib10", "generic", "li", "x" * 40, "lib1b"]
RANGES = [500, 600, 700, 800, 900, 1000]


def rand_aux(rng):
    out = b""
    for _ in range(rng.choice([0, 1, 1, 2, 3, 5])):
        tag = rng.choice([b"RG", b"MQ", b"XY", b"NM", b"AS", b"MQ"])
        typ = rng.choice("AcCsSiIfZZHB")
        if tag == b"RG" and rng.random() < 0.8: typ = "Z"
        if tag == b"MQ" and rng.random() < 0.7: typ = rng.choice("CcSsIi")
        if typ in "AcC": val = bytes([rng.randrange(256)])
        elif typ in "sS": val = struct.pack("<H", rng.randrange(65536))
        elif typ in "iIf": val = struct.pack("<I", rng.randrange(2**32))
        elif typ in "ZH":
            val = (rng.choice(NAMES * 4 + ["nope", "l", "lib", ""]) if tag == b"RG" else "".join(rng.choice("abc12") for _ in range(rng.randrange(6)))).encode() + b"\0"
        else:
            sub = rng.choice("cCsSiIf"); cnt = rng.randrange(5)
            val = sub.encode() + struct.pack("<I", cnt) + bytes(rng.randrange(256) for _ in range(cnt * {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]))
        out += tag + typ.encode() + val
    if rng.random() < 0.1:
        out += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 4)))       # junk tail
    return out


def rand_rec(rng):
    l_seq = rng.choice([100, 100, 100, 76, 33, 150, 255, 1, 300, 129])
    flag = rng.choice([0x1 | 0x2, 0x1 | 0x2 | 0x10, 0x1 | 0x2 | 0x20, 0x1, 0x1 | 0x20, 0x1 | 0x10, 0x1 | 0x4 | 0x40, 0x1 | 0x4 | 0x20 | 0x80,
                       0x1 | 0x8, 0x1 | 0x4 | 0x8, 0x2, rng.randrange(0x1000)])
    if rng.random() < 0.05:
        flag |= rng.choice([0x100, 0x200, 0x400, 0x800])
    ops = []
    left = l_seq
    kind = rng.random()
    if flag & 0x4:
        ops = []
    elif kind < 0.4:
        ops = [(l_seq, 0)]
    elif kind < 0.75:              # what an aligner writes: clips at the ends, a few I / D inside
        if rng.random() < 0.5: ops.append((rng.randrange(1, 40), 4)); left -= ops[-1][0]
        tail = rng.randrange(1, 40) if rng.random() < 0.5 and left > 45 else 0
        body = left - tail
        while body > 12 and rng.random() < 0.6 and len(ops) < 9:
            m = rng.randrange(5, max(6, body - 5)); ops.append((m, rng.choice([0, 7, 8]))); body -= m
            if rng.random() < 0.5: ops.append((rng.randrange(1, 30), 2))
            elif body > 3: i = rng.randrange(1, min(20, body)); ops.append((i, 1)); body -= i
        if body > 0: ops.append((body, 0))
        if tail: ops.append((tail, 4))
    else:
        while left > 0 and len(ops) < rng.choice([2, 3, 5, 9]):
            op = rng.choice([0, 0, 0, 1, 2, 4, 4, 7, 8, 3, 5, 6, 9, 12])
            l = rng.randrange(1, max(2, left))
            ops.append((l, op))
            if op in (0, 1, 4, 7, 8):
                left -= l
        if left > 0:
            ops.append((left, 0))
    if ops and rng.random() < 0.04:
        ops.append((rng.randrange(1, 60), rng.choice([0, 0, 1, 4])))          # a CIGAR that reaches past l_seq: the qualities are read as bases
    codes = [rng.choice([1, 2, 4, 8, 15]) for _ in range(l_seq + (l_seq & 1))]
    if rng.random() < 0.05:
        codes[rng.randrange(l_seq)] = rng.choice([0, 3, 5, 7, 9, 14])
    seq = bytes((codes[2 * i] << 4) | codes[2 * i + 1] for i in range((l_seq + 1) // 2))
    rec = T._rec(flag, tid=0, pos=rng.randrange(0, 1900), mtid=rng.choice([0, 0, 0, 1]), mpos=rng.randrange(0, 1900),
                 isize=rng.choice([300, -300, 650, 5000, -5000, 2000000, 0, rng.randrange(-3000, 3000)]), mapq=rng.choice([0, 5, 10, 60]),
                 cigar=tuple(ops), seq=seq, qual=bytes(rng.choice([0x28, 0x28, 0x28, 0x11, 0x30, 0xFF]) for _ in range(l_seq)) if rng.random() < 0.3 else None, tags=rand_aux(rng), qname=("q%d" % rng.randrange(10 ** rng.randrange(1, 6))).encode() + b"\0", l_seq=l_seq,
                 pad=bytes(rng.randrange(256) for _ in range(3)))
    if rng.random() < 0.01:
        rec = rec[:rng.randrange(8, len(rec)) // 4 * 4]
    return rec


ctx = capi.Context(0)
ctx.set_reference([b"ACGT" * 500])
ctx.set_insert_ranges(NAMES, RANGES)
for seed in range(first, first + rounds):
    rng = random.Random(seed)
    recs = [rand_rec(rng) for _ in range(3000)]
    raw = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    off = np.zeros(len(recs) + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in recs], out=off[1:])
    q = rng.choice([0, 10, 30])
    pipe = capi.Pipeline(ctx, len(recs), len(raw), cap_cand=len(recs), maxpedelsize=rng.choice([1000, 1000000]), qthreshold=q)
    from tests.support import oraclebind as ob
    tri = ob.triage_records(raw, off, NAMES, RANGES, qthreshold=q, maxpedelsize=pipe.tp.maxpedelsize)
    pipe.upload(raw, off); pipe.triage(); pipe.fetch_counts()
    h_cls = pipe.d_class.download(np.uint8, len(recs))
    o_cls = np.array([21 if (t.cls == 3 and t.n_ev > capi.MAX_EV) else t.cls for t, _ in tri], dtype=np.uint8)
    bad = np.nonzero(h_cls != o_cls)[0]
    if len(bad):
        for i in bad[:5]:
            r = recs[i]
            tid, pos, lq, mapq, b, nc, flag, ls, mtid, mpos, isz = struct.unpack("<iiBBHHHiiii", r[:32])
            cig = [(w >> 4, w & 15) for w in struct.unpack("<%dI" % nc, r[32 + lq:32 + lq + 4 * nc])] if 32 + lq + 4 * nc <= len(r) else "truncated"
            aux = r[32 + lq + 4 * nc + (ls + 1) // 2 + ls:]
            print("MISMATCH seed %d record %d: hip %d oracle %d | len %d flag 0x%x mapq %d l_seq %d mtid %d isize %d cigar %r aux %r" %
                  (seed, i, h_cls[i], o_cls[i], len(r), flag, mapq, ls, mtid, isz, cig, aux), flush=True)
        sys.exit(1)
    try:
        tri, cand = T._compare_triage(pipe, raw, off, NAMES, RANGES, tri=tri)
    except AssertionError as ex:
        # The one known way the two record forms differ: an aux area that ends in 1-3 bytes which are not a whole field (a truncated
        # record, a junk tail).  Records travel at 4-byte aligned offsets with up to three zero bytes behind them, and the tag walk
        # takes a tail of fewer than 4 bytes for that padding: whether tail + padding reaches 4 depends on the record's length, which
        # leaving the qualities out changes.  Anything else is a failure.
        def short_tail(r):
            tid, pos, lq, mapq, b, nc, flag, ls, mtid, mpos, isz = struct.unpack("<iiBBHHHiiii", r[:32])
            s_ = 32 + lq + 4 * nc + (ls + 1) // 2 + ls
            if ls < 0 or s_ > len(r):
                return False
            sizes = {"A": 1, "c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4, "d": 8}
            while s_ + 4 <= len(r):
                typ = chr(r[s_ + 2]); s_ += 3
                if typ in "ZH":
                    while s_ < len(r) and r[s_]: s_ += 1
                    s_ += 1
                elif typ == "B":
                    if s_ + 5 > len(r): return False
                    es = sizes.get(chr(r[s_]), 0); s_ += 5 + es * struct.unpack("<I", r[s_ + 1:s_ + 5])[0]
                elif typ in sizes: s_ += sizes[typ]
                else: return False
            return 0 < len(r) - s_ < 4
        lst = ex.args[0] if ex.args and isinstance(ex.args[0], list) else None
        if lst and all(short_tail(recs[i]) for i, _, _ in lst):
            print("seed %d: %d record(s) with an aux tail of 1-3 bytes classed differently with and without qualities (known ambiguity): %r" % (seed, len(lst), lst[:3]), flush=True)
            continue
        print("seed %d: %r" % (seed, ex.args), flush=True)
        raise
    cl = {}
    for t, _ in tri:
        cl[t.cls] = cl.get(t.cls, 0) + 1
    print("seed %d ok: %d candidates, classes %r" % (seed, len(cand), dict(sorted(cl.items()))), flush=True)
ctx.close()
