#!/usr/bin/env python3
"""Golden of BASELINE configs[3]'s reference (3 Gb, 24 contigs, human length spread) at a low coverage: the input comes from
tests/support/simgen.c (regenerated from its arguments wherever the test runs), the VCF's digest from tests/shim/indelminer_shim
-- this host driver over the CPU oracle, record at a time -- because the compiled reference cannot finish an input with 240 Mb
contigs (its per-candidate strlen of the contig, src/alignment.c:771).    python tests/golden/make_golden_wgs.py 3"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.support.shimbuild import build_shim  # noqa: E402

cov = sys.argv[1]
gen = os.path.join(ROOT, "tests", "support", "simgen")
subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-pthread", "-o", gen, gen + ".c", "-lz", "-lm"])
args = ["--human", "3000000000", "--coverage", cov, "--seed", "3"]
with tempfile.TemporaryDirectory(dir=os.environ.get("IM_GOLDEN_TMP", "/tmp")) as td:
    out = subprocess.run([gen, "--prefix", os.path.join(td, "w"), "--threads", "8"] + args, stdout=subprocess.PIPE, check=True)
    info = json.loads(out.stdout.decode())
    t = time.perf_counter()
    p = subprocess.run([build_shim(), "-i", "w.cfg", "w.fa", "s=w.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, INDELMINER_PIPELINE="host"))
    assert p.returncode == 0
    doc = {"what": "BASELINE configs[3] at %sx: simgen %s; digest of the VCF of tests/shim/indelminer_shim -i w.cfg (record-at-a-time path over the CPU "
                   "oracle, %.0f s in the build container)" % (cov, " ".join(args), time.perf_counter() - t),
           "simgen": args, "records_in_bam": info["records"], "bam_bytes": info["bam_bytes"], "md5": hashlib.md5(p.stdout).hexdigest(),
           "vcf_records": sum(1 for l in p.stdout.splitlines() if not l.startswith(b"#"))}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "large_wgs%sx.json" % cov), "w"), indent=1)
    print(json.dumps(doc))
