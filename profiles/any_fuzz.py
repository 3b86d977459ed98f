#!/usr/bin/env python3
"""Soak on the GPU box: the realign kernels -- laid-out and general pass -- against the CPU oracle on random parameter sets that
straddle every border between them (read lengths 4..2600, -g 0..400, every -k, short contigs, anchors on the contig's ends).
    python profiles/any_fuzz.py [first_seed] [n_rounds]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from indelminer_amd import capi
from tests.support import gpucmp, oraclebind as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = capi.Context(0)
total = bad_total = 0
by_status = {}
for seed in range(first, first + rounds):
    rng = random.Random(seed)
    g = rng.choice([0, 0, 0, 1, 2, 3, 7, 12, 31, 32, 33, 60, 61, 64, 100, 250, 400])
    kw = dict(klength=rng.choice([2, 3, 4, 5, 6, 6, 7, 8, 9, 11, 13, 14, 15]), numgaps=g,
              maxdelsize=rng.choice([50, 300, 1000, 2500, 6000]), ethreshold=rng.choice([1, 5, 10, 25]))
    clen = rng.choice([1200, 3000, 9000, 40000])
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    if rng.random() < 0.3:      # a low-complexity stretch: bands without votes, repeated k-mers
        at = rng.randrange(clen - 400)
        unit = rng.choice(["A", "AC", "ACG", "TTTG"])
        contig = contig[:at] + (unit * 300)[:300] + contig[at + 300:]
    assert len(contig) == clen
    cases = []
    for _ in range(rng.choice([20, 48, 90])):
        L = min(rng.choice([4, 17, 36, 100, 101, 150, 255, 256, 257, 300, 511, 1019, 1020, 1021, 1500, 2047, 2049, 2600]), clen - 10)
        Rm = rng.choice([60, 200, 705, 1500, 3000])
        anchor = rng.choice([0, clen - 1, rng.randint(0, clen - 1)])
        p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
        cut = rng.randint(1, max(1, L - 1))
        d = rng.randint(1, 80)
        typ = rng.random()
        if typ < 0.45:
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.8:
            read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
        else:
            read = contig[p:p + L]
        read = "".join((rng.choice("ACGTN") if rng.random() < rng.choice([0, 0.004, 0.02]) else ch) for ch in read)
        if len(read) < 4:
            read = contig[:4]
        cases.append(dict(anchor=anchor, range_max=Rm, read=read))
    ctx.set_reference([contig.encode()])
    n = len(cases)
    rc, out = ctx.realign_batch(capi.params(**kw), [c["read"].encode() for c in cases], np.zeros(n, np.int32),
                                np.array([c["anchor"] for c in cases], np.int32), np.array([c["range_max"] for c in cases], np.int32),
                                allow=(capi.E_ABORT, capi.E_OVERFLOW))
    P = ob.params(**kw)
    bad = []
    for i, c in enumerate(cases):
        st, res = ob.realign(P, contig.encode(), clen, c["anchor"], c["range_max"], c["read"])
        by_status[st] = by_status.get(st, 0) + 1
        msg = gpucmp.hip_vs_oracle(out[i], st, res)
        if msg:
            bad.append((i, msg, len(c["read"])))
    total += n; bad_total += len(bad)
    print("seed %d %r contig %d: %d reads, %d differ%s" % (seed, kw, clen, n, len(bad), (" FIRST " + repr(bad[0])) if bad else ""), flush=True)
print("TOTAL %d reads, %d differ; oracle statuses %r" % (total, bad_total, by_status))
ctx.close()
sys.exit(1 if bad_total else 0)
