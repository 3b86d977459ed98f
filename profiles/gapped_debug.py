#!/usr/bin/env python3
"""Diagnostic: runs the fuzzed gapped parity cases and prints, per mismatch, the kernel's failure code
(reserved[6]) and both results.  usage: python profiles/gapped_debug.py [trial ...]"""
import random
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from indelminer_amd import capi
from tests.support import oraclebind as ob, gpucmp

ctx = capi.Context(0)
rng = random.Random(99)
want = [int(a) for a in sys.argv[1:]]
for trial in range(14):
    kw = dict(klength=rng.choice([4, 6, 6, 8, 12]), numgaps=rng.choice([1, 2, 3, 5, 8, 12]),
              maxdelsize=rng.choice([300, 1000]), ethreshold=rng.choice([5, 10]))
    clen = rng.choice([900, 4000, 20000])
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    cases = []
    for _ in range(40):
        L = min(rng.choice([36, 76, 100, 150, 250]), clen - 10)
        Rm = rng.choice([200, 705])
        anchor = rng.randint(0, clen - 1)
        p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
        cut = rng.randint(1, max(1, L - 1))
        d = rng.randint(1, 12)
        typ = rng.random()
        if typ < 0.45:
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.85:
            read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
        else:
            read = contig[p:p + L]
        read = "".join((rng.choice("ACGT") if rng.random() < 0.01 else ch) for ch in read)
        if len(read) < 4:
            read = contig[:4]
        cases.append(dict(anchor=anchor, range_max=Rm, read=read))
    if want and trial not in want:
        continue
    ctx.set_reference([contig.encode()])
    reads = [c["read"].encode() for c in cases]
    rc, out = ctx.realign_batch(capi.params(**kw), reads, np.zeros(len(reads), np.int32), [c["anchor"] for c in cases],
                                [c["range_max"] for c in cases], allow=(capi.E_ABORT, capi.E_OVERFLOW))
    P = ob.params(**kw)
    nbad = 0
    for i, c in enumerate(cases):
        st, res = ob.realign(P, contig.encode(), len(contig), c["anchor"], c["range_max"], c["read"])
        msg = gpucmp.hip_vs_oracle(out[i], st, res)
        if msg:
            nbad += 1
            r = out[i]
            print("trial %d %r case %d L=%d: %s | code %d | hip band %s | oracle n_band %d pieces %s" %
                  (trial, kw, i, len(c["read"]), msg, int(r["reserved"][6]),
                   [(int(b["r1"]), int(b["r2"]), int(b["q1"]), int(b["q2"]), int(b["low"])) for b in r["band"]],
                   res.n_band, [(res.piece[j].r1, res.piece[j].r2, res.piece[j].q1, res.piece[j].q2, res.piece[j].low, res.piece[j].up,
                                 [(w >> 4, w & 15) for w in list(res.piece[j].ops)[:res.piece[j].n_ops]]) for j in range(res.n_band)]))
            if r["status"] == 1:
                print("    hip ops", [(int(w) >> 4, int(w) & 15) for w in r["ops"][:int(r["n_ops"])]], "ref_start", int(r["ref_start"]))
            if st == 1:
                print("    ora ops", [(w >> 4, w & 15) for w in list(res.ops)[:res.n_ops]], "ref_start", res.ref_start)
    print("trial %d %r: %d of %d differ" % (trial, kw, nbad, len(cases)), flush=True)
