#!/usr/bin/env python3
"""What each stage of bench.py's step costs INSIDE the overlapped pipeline: the step with one stage (or all but one) taken out
of every buffer set's launch list, four streams as in the bench.  A stage's marginal cost under overlap is what the step gains
when it goes -- not its duration alone on the device.

    python profiles/step_probe.py [steps]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from indelminer_amd import capi, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30, read_len=100)
ctx = capi.Context(0)
ctx.set_reference([refs[0].tobytes()] * bench.PIPELINE_DEPTH)
ctx.set_insert_ranges(["generic"], [rd.range_max])
ps = bench.PipeStep(ctx, rd, bench.PIPELINE_DEPTH)
ps.step(); ps.sync()                      # every buffer set holds a complete pass: partial passes below re-use what is there
for _ in range(bench.PIPELINE_DEPTH):
    ps.step()
ps.sync()
L = capi.lib()
names = ["im_depth_reset", "im_dev_triage", "im_dev_realign_n", "im_depth_scan", "im_dev_flush_groupby"]
full = [list(s["calls"]) for s in ps.sets]
kinds = [[getattr(fn, "__name__", "?").replace("im_dev_", "").replace("im_", "") for fn, _ in calls] for calls in full]
assert all(getattr(fn, "__name__", "?") in names for fn, _ in full[0]), [getattr(fn, "__name__", "?") for fn, _ in full[0]]

def run(keep):
    for s, calls, kd in zip(ps.sets, full, kinds):
        s["calls"] = [c for c, k in zip(calls, kd) if keep(k)]
        s["graph"] = None
    ps.capture()
    for _ in range(20):
        ps.step()
    ps.sync()
    t = time.perf_counter()
    for _ in range(steps):
        ps.step()
    ps.sync()
    return (time.perf_counter() - t) / steps * 1e6


out = {"stages": sorted(set(kinds[0])), "us_per_step": {}}
out["us_per_step"]["all"] = run(lambda k: True)
for st in sorted(set(kinds[0])):
    out["us_per_step"]["without " + st] = run(lambda k, st=st: k != st)
    out["us_per_step"]["only " + st] = run(lambda k, st=st: k == st)
out["us_per_step"]["all again"] = run(lambda k: True)
print(json.dumps(out, indent=1))
