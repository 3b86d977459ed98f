#!/usr/bin/env python3
"""Throughput of the -g > 0 realign path (HIP events around im_dev_realign) for a few band widths.

    python profiles/gapped_probe.py [n_reads] [g ...]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from indelminer_amd import capi, synth  # noqa: E402
import bench  # noqa: E402,F401
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import legacy_shard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24464
GS = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 5, 12]
refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
n_all = len(cand["index"])
L = capi.lib()
reps = (n + n_all - 1) // n_all
sub = {k: (np.concatenate([v] * reps)[:n] if isinstance(v, np.ndarray) and v.shape[:1] == (n_all,) else v) for k, v in cand.items()}
sh = legacy_shard.Shard(ctx, refs[0], sub, 100)
for g in GS:
    P = capi.params(numgaps=g)
    t = capi.Timer(ctx)
    ctx._check(L.im_dev_realign(ctx.h, C.byref(P), C.byref(sh.batch), ctx.stream))
    ts = []
    for _ in range(3):
        t.start(ctx.stream)
        ctx._check(L.im_dev_realign(ctx.h, C.byref(P), C.byref(sh.batch), ctx.stream))
        t.stop(ctx.stream)
        ts.append(t.elapsed_ms())
    ms = min(ts)
    res = sh.results()
    print("g=%2d n=%6d  %9.3f ms  %9.3f Mreads/s   evidence reads %d" % (g, n, ms, n / ms / 1e3, int((res["status"] == 1).sum())), flush=True)
