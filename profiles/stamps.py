#!/usr/bin/env python3
"""Per-phase shader-cycle shares of realign_kernel from the diagnostic (-DIM_STAMPS) build.
Run on the GPU box:  INDELMINER_AMD_LIB=indelminer_amd/libindelminer_amd_stamps.so python profiles/stamps.py
Shares only -- the stamped build is slower than the product and its run time is never quoted."""
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

NAMES = {0: "setup (loads, staging, geometry)", 1: "p1 table build", 2: "p1 histogram clear", 3: "p1 vote",
         4: "p1 evaluate", 5: "p1 argmax reduce", 6: "p1 diagonal scan", 7: "case selection",
         8: "p2 table build", 9: "p2 histogram clear", 10: "p2 vote", 11: "p2 evaluate", 12: "p2 argmax reduce",
         13: "p2 diagonal scan", 14: "combine + segments + evidence"}

refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
cand = synth.candidates(rd)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()])
sh = legacy_shard.Shard(ctx, refs[0], cand, 100)
L = capi.lib()
acc = (C.c_ulonglong * 32)()
sh.step(); sh.sync()
L.im_debug_stamps_read(acc, 1)
for _ in range(10):
    sh.step()
sh.sync()
L.im_debug_stamps_read(acc, 1)
v = np.array(list(acc), dtype=np.float64)
tot = v.sum()
print("phase shares of wave-0-lane cycles, %d reads x 10 launches" % sh.n)
for i in range(15):
    print("  %-36s %6.2f %%   %8.0f cycles/read" % (NAMES[i], 100 * v[i] / tot, v[i] / (10 * sh.n)))
print("  total %.0f cycles/read" % (tot / (10 * sh.n)))
