#!/usr/bin/env python3
"""Host issue time vs device time of the pipelined bench step (realign on one stream, clustering on another)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import capi, synth  # noqa: E402
import bench  # noqa: E402,F401
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import legacy_shard  # noqa: E402

refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
if len(sys.argv) > 1:                       # fewer reads per step: is the loop bound by the host's launch rate?
    import numpy as np
    n_sub = int(sys.argv[1]); n_all = len(cand["index"])
    cand = {k: (v[:n_sub] if isinstance(v, np.ndarray) and v.shape[:1] == (n_all,) else v) for k, v in cand.items()}
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
sh = legacy_shard.Shard(ctx, refs[0], cand, 100)
for _ in range(5):
    sh.step()
sh.sync()
K = 300
t0 = time.perf_counter()
for _ in range(K):
    sh.step()
t1 = time.perf_counter()
sh.sync()
t2 = time.perf_counter()
print("host issue %.2f us per step, issue + drain %.2f us per step" % ((t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
