#!/usr/bin/env python3
"""Step time of the bench workload launched kernel by kernel vs replayed as one HIP graph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import capi, synth  # noqa: E402
import bench  # noqa: E402

refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
sh = bench.Shard(ctx, refs[0], cand, 100)
for _ in range(5):
    sh.step()
sh.sync()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    sh.step()
sh.sync()
t1 = time.perf_counter()
print("kernel by kernel: %.2f us per step" % ((t1 - t0) / K * 1e6))
with capi.Graph.capture(ctx) as g:
    sh.step()
for _ in range(5):
    g.launch()
sh.sync()
t0 = time.perf_counter()
for _ in range(K):
    g.launch()
sh.sync()
t1 = time.perf_counter()
print("one graph per step: %.2f us per step" % ((t1 - t0) / K * 1e6))
