#!/usr/bin/env python3
"""Throughput of the host-buffer entry point im_realign_batch (pack + PCIe both ways + kernel), by batch size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from indelminer_amd import capi, synth  # noqa: E402

refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
n0 = len(cand["index"])
for mult in (1, 8, 32):
    reads = [bytes(b) for b in cand["bases"]] * mult
    tid = [0] * (n0 * mult)
    anchor = list(cand["anchor"]) * mult
    rng = list(cand["range_max"]) * mult
    ctx.realign_batch(capi.params(), reads[:1000], tid[:1000], anchor[:1000], rng[:1000])
    best = None
    for _ in range(3):
        t = time.perf_counter()
        rc, out = ctx.realign_batch(capi.params(), reads, tid, anchor, rng)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print("n=%7d  %8.2f ms  %6.2f M reads/s (includes the ctypes marshalling of the Python wrapper)  evidence %d" % (len(reads), best * 1e3, len(reads) / best / 1e6, int((out["status"] == 1).sum())))
