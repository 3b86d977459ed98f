#!/usr/bin/env python3
"""realign_kernel throughput by -k (HIP events around im_dev_realign, 48 184 candidate reads of the bench contig): k <= 6 takes the
direct 4 KiB table (k = 6 its specialised form), k > 6 the 512-slot hash.
    python profiles/k_probe.py [k ...]"""
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

KS = [int(a) for a in sys.argv[1:]] or [4, 5, 6, 7, 8, 10, 12, 15]
refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
n_all = len(cand["index"])
n = 4 * n_all
L = capi.lib()
sub = {k: (np.concatenate([v] * 4)[:n] if isinstance(v, np.ndarray) and v.shape[:1] == (n_all,) else v) for k, v in cand.items()}
sh = legacy_shard.Shard(ctx, refs[0], sub, 100)
for k in KS:
    P = capi.params(klength=k)
    t = capi.Timer(ctx)
    ctx._check(L.im_dev_realign(ctx.h, C.byref(P), C.byref(sh.batch), ctx.stream))
    ts = []
    for _ in range(5):
        t.start(ctx.stream)
        ctx._check(L.im_dev_realign(ctx.h, C.byref(P), C.byref(sh.batch), ctx.stream))
        t.stop(ctx.stream)
        ts.append(t.elapsed_ms())
    ms = min(ts)
    res = sh.results()
    print("k=%2d n=%6d  %8.3f ms  %8.1f Mreads/s   evidence reads %d" % (k, n, ms, n / ms / 1e3, int((res["status"] == 1).sum())), flush=True)
