#!/usr/bin/env python3
"""Throughput of the annotate-mode Smith-Waterman (im_support_batch, K7) against its CPU restatement (oracle
imo_sw_indel, one thread): tasks shaped like check_for_indel's (100-base read vs a mutated window of
L + 2 * indel bases, src/variant.c:1427-1559)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from indelminer_amd import capi  # noqa: E402

rng = np.random.default_rng(9)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
targets, queries = [], []
for it in range(N):
    d = int(rng.integers(1, 50))
    len1 = 100 + 2 * d
    t = rng.choice(list(b"ACGT"), size=len1).astype(np.uint8)
    p = int(rng.integers(0, 2 * d + 1))
    q = t[p:p + 100].copy()
    if rng.random() < 0.5:
        cut = int(rng.integers(20, 80))
        q = np.concatenate([q[:cut], q[cut + min(d, 15):], rng.choice(list(b"ACGT"), size=min(d, 15)).astype(np.uint8)])
    sub = rng.random(len(q)) < 0.01
    q[sub] = rng.choice(list(b"ACGT"), size=int(sub.sum())).astype(np.uint8)
    targets.append(t.tobytes()); queries.append(q.tobytes())
cells = sum((len(a) + 1) * (len(b) + 1) for a, b in zip(targets, queries))
ctx = capi.Context(0)
ctx.support_batch(targets[:100], queries[:100])
best = None
for _ in range(3):
    t0 = time.perf_counter()
    out = ctx.support_batch(targets, queries)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
print("GPU  im_support_batch: %d tasks  %.2f ms  %.2f M tasks/s  %.1f GCUPS  (host buffers, incl. copies and the Python wrapper)"
      % (N, best * 1e3, N / best / 1e6, cells / best / 1e9))
try:
    from tests.support import oraclebind as ob
    L = ob.lib()
    m = min(N, 3000)
    s = C.c_int32(); i = C.c_int32(); a = C.c_int32()
    t0 = time.perf_counter()
    bad = 0
    for k in range(m):
        L.imo_sw_indel(targets[k], len(targets[k]), queries[k], len(queries[k]), C.byref(s), C.byref(i), C.byref(a))
        bad += int((s.value, i.value, a.value) != tuple(int(x) for x in out[k][:3]))
    dt = time.perf_counter() - t0
    c3 = sum((len(a_) + 1) * (len(b_) + 1) for a_, b_ in zip(targets[:m], queries[:m]))
    print("CPU  oracle restatement, 1 thread: %d tasks  %.2f ms  %.3f M tasks/s  %.2f GCUPS   mismatches vs GPU: %d"
          % (m, dt * 1e3, m / dt / 1e6, c3 / dt / 1e9, bad))
except Exception as e:  # oracle not built on this box
    print("oracle not available:", e)
