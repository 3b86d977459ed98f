#!/usr/bin/env python3
"""Soak on the GPU box: im_support_batch (both forms of the support kernel) against the oracle's full-matrix Smith-Waterman +
traceback on random tasks whose windows / queries straddle the LDS form's bounds (4095 bytes, 1020 bases).
    python profiles/support_fuzz.py [first_seed] [n_rounds]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from indelminer_amd import capi
from tests.support import oraclebind as ob

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = ob.lib()
ctx = capi.Context(0)
total = bad = big = 0
for seed in range(first, first + rounds):
    rng = np.random.default_rng(seed)
    targets, queries = [], []
    for it in range(24):
        len1 = int(rng.choice([60, 300, 1500, 4094, 4095, 4096, 4097, 5000, 9000, 15000]))
        len2 = int(rng.choice([30, 100, 150, 300, 1019, 1020, 1021, 1024, 1500, 2300]))
        t = rng.choice(list(b"ACGT"), size=len1).astype(np.uint8)
        p = int(rng.integers(0, max(1, len1 - len2)))
        q = t[p:p + len2].copy()
        if len(q) < len2:
            q = np.concatenate([q, rng.choice(list(b"ACGT"), size=len2 - len(q)).astype(np.uint8)])
        typ = rng.random()
        if typ < 0.35 and len2 > 20:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 40))
            q = np.concatenate([q[:cut], q[cut + d:], rng.choice(list(b"ACGT"), size=d).astype(np.uint8)])
        elif typ < 0.7 and len2 > 20:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 40))
            q = np.concatenate([q[:cut], rng.choice(list(b"ACGT"), size=d).astype(np.uint8), q[cut:]])[:len2]
        elif typ < 0.8:
            q = rng.choice(list(b"ACGT"), size=len2).astype(np.uint8)
        sub = rng.random(len(q)) < rng.choice([0, 0.01, 0.05, 0.2])
        q[sub] = rng.choice(list(b"ACGTN"), size=int(sub.sum())).astype(np.uint8)
        if it % 7 == 0:
            t[len(t) // 3] = ord("n"); q[len(q) // 2] = ord("a")        # the comparison folds case (toupper), the substitution count does not
        targets.append(t.tobytes()); queries.append(q.tobytes())
    got = ctx.support_batch(targets, queries)
    nb = 0
    for k, (t, q) in enumerate(zip(targets, queries)):
        s, i, a = C.c_int32(), C.c_int32(), C.c_int32()
        L.imo_sw_indel(t, len(t), q, len(q), C.byref(s), C.byref(i), C.byref(a))
        ok = tuple(int(x) for x in got[k][:3]) == (s.value, i.value, a.value) and int(got[k][3]) == capi.ST_EVIDENCE
        big += len(t) > 4095 or len(q) > 1020
        if not ok:
            nb += 1
            print("  seed %d task %d (%d x %d): hip %r oracle %r" % (seed, k, len(t), len(q), tuple(int(x) for x in got[k]), (s.value, i.value, a.value)), flush=True)
    total += len(targets); bad += nb
    print("seed %d: %d tasks, %d differ" % (seed, len(targets), nb), flush=True)
print("TOTAL %d tasks (%d in the second form), %d differ" % (total, big, bad))
ctx.close()
sys.exit(1 if bad else 0)
