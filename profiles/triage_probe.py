#!/usr/bin/env python3
"""The three triage launches alone on the device (HIP events, nothing else running): configs[1] chunk and the config-3 shard,
with and without the pileup depth scatter, records with and without qualities.  python profiles/triage_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from indelminer_amd import capi, rawrec, synth  # noqa: E402

L = capi.lib()
ONLY = sys.argv[1] if len(sys.argv) > 1 else None          # "configs[1]" / "shard3": that input alone, records without qualities, with depth
for name, kw in (("configs[1]", dict(seed=1, ref_len=1_000_000)), ("shard3", dict(seed=2, ref_len=6_250_000, big_every=7))):
    if ONLY and name != ONLY:
        continue
    refs, rd = synth.simulate(coverage=30, read_len=100, **kw)
    ctx = capi.Context(0)
    ctx.set_reference([refs[0].tobytes()])
    ctx.set_insert_ranges(["generic"], [rd.range_max])
    ctx.depth_enable()
    # AUX=1: every record carries what an aligner writes in front of MQ / RG (NM:C MD:Z AS:C XS:C MC:Z, 31 bytes) and an RG:Z tag
    AUX = os.environ.get("AUX") == "1"
    for qual in ((False,) if ONLY else (False, True)):
        raw, off = rawrec.records(rd, qual=qual, rg="generic" if AUX else None,
                                  aux_prefix=(b"NMC\x00" + b"MDZ100\x00" + b"ASC\x64" + b"XSC\x00" + b"MCZ100M\x00") if AUX else b"")
        for want_depth in ((True,) if ONLY else (True, False)):
            pipe = capi.Pipeline(ctx, rd.n, len(raw), cap_cand=max(4096, rd.n // 8), read_len_max=100, want_depth=want_depth)
            pipe.upload(raw, off)
            tp = capi.TriageParams(pipe.tp.qthreshold, pipe.tp.ethreshold_vcfcheck, pipe.tp.maxpedelsize, 1 if want_depth else 0, 0, 1)
            tm = capi.Timer(ctx)
            ts = []
            for _ in range(8):
                ctx._check(L.im_stream_sync(ctx.h, ctx.stream))
                tm.start(ctx.stream)
                ctx._check(L.im_dev_triage(ctx.h, C.byref(tp), C.byref(pipe.recs), C.byref(pipe.cands), pipe.d_ts.ptr, pipe.ts_bytes, ctx.stream))
                tm.stop(ctx.stream)
                ts.append(tm.elapsed_ms())
            ms = float(np.median(ts[2:]))
            print("%-10s qual=%d depth=%d  records %7d  bytes %9d  %7.1f us  %6.2f G records/s  %5.2f TB/s"
                  % (name, qual, want_depth, rd.n, len(raw), ms * 1e3, rd.n / ms / 1e6, len(raw) / ms / 1e9), flush=True)
            del pipe
    ctx.close()
