#!/usr/bin/env python3
"""realign_kernel duration vs batch size (HIP events): separates per-read latency from throughput."""
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

refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
n_all = len(cand["index"])
L = capi.lib()
SIZES = [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 2048, 4096, 6144, 8192, 12046, 24092, 48184, 96368, 192736]
for n in SIZES:
    reps = (n + n_all - 1) // n_all
    sub = {k: (np.concatenate([v] * reps)[:n] if isinstance(v, np.ndarray) and v.shape[:1] == (n_all,) else v) for k, v in cand.items()}
    sh = legacy_shard.Shard(ctx, refs[0], sub, 100)
    t = capi.Timer(ctx)
    for _ in range(3):
        ctx._check(L.im_dev_realign(ctx.h, C.byref(sh.P), C.byref(sh.batch), ctx.stream))
    ts = []
    for _ in range(10):
        t.start(ctx.stream)
        ctx._check(L.im_dev_realign(ctx.h, C.byref(sh.P), C.byref(sh.batch), ctx.stream))
        t.stop(ctx.stream)
        ts.append(t.elapsed_ms())
    ms = min(ts)
    print("n=%7d  %8.3f ms  %8.1f Mreads/s  %6.2f us per read-slot-round" % (n, ms, n / ms / 1e3, ms * 1e3 / max(1, (n + 4095) // 4096)))
