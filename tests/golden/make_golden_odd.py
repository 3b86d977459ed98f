#!/usr/bin/env python3
"""Writes tests/golden/odd_inputs.json: for a fixed list of seeds of tests/golden/odd_inputs.py, what the COMPILED REFERENCE
(oracle/_ref/indelminer, built from /root/reference by oracle/Makefile) does with the input -- exit status and the md5 of its
stdout (also of what it printed in front of an abort; not where it died of a signal, its buffered output is lost then).  Run in the build container; the -m gpu test regenerates the inputs from the seeds on the GPU box
and holds the product (real kernels) to these results.
    python tests/golden/make_golden_odd.py"""
import hashlib, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden.odd_inputs import make_input

REF = os.path.join(ROOT, "oracle", "_ref", "indelminer")
# 60037 / 60058 / 60233: what profiles/ref_diff_fuzz.py found in round 3 (a read group on counted reads only; an RG tag that is no
# string in the estimation pass; -c ctg0 with records placed beyond the contig's end)
SEEDS = list(range(20000, 20070)) + [60037, 60058, 60233]

out = {}
for seed in SEEDS:
    d = tempfile.mkdtemp(prefix="odd%d_" % seed, dir="/tmp")
    try:
        cmd, info = make_input(seed, d)
        r = subprocess.run([REF] + cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        err = r.stderr.decode(errors="replace").strip().splitlines()
        out[str(seed)] = {"cmd": cmd, "rc": r.returncode if r.returncode >= 0 else 1, "signal": r.returncode < 0,
                          "md5": hashlib.md5(r.stdout).hexdigest() if r.returncode >= 0 else None, "bytes": len(r.stdout),
                          "last_stderr_line": err[-1][:80] if r.returncode != 0 and err else ""}
        print(seed, out[str(seed)]["rc"], out[str(seed)]["bytes"], " ".join(cmd), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "odd_inputs.json"), "w"), indent=0, sort_keys=True)
print("%d inputs, %d the reference completes" % (len(out), sum(1 for v in out.values() if v["rc"] == 0)))
