#!/usr/bin/env python3
"""Golden of the deep-locus input (tests/support/deeplocus.py), made by the REFERENCE compiled in place (oracle/_ref/indelminer):
the VCF whose DP= at the stacked deletion shows samtools' pileup cap.   python tests/golden/make_golden_deep.py"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.support import deeplocus  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "indelminer")
for name, depth in (("deep_locus_9000", 9000), ("deep_locus_7900", 7900)):
    with tempfile.TemporaryDirectory() as td:
        n, where = deeplocus.write(td, depth=depth)
        q = subprocess.run([REF_BIN, "-i", "cfg.txt", "ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert q.returncode == 0, q.stderr.decode()[-2000:]
        open(os.path.join(ROOT, "tests", "golden", "vcf", name + ".vcf"), "wb").write(q.stdout)
        body = [l for l in q.stdout.splitlines() if not l.startswith(b"#")]
        near = [l for l in body if abs(int(l.split(b"\t")[1]) - where["deletion_start"]) < 60]
        print(name, n, "reads,", where, len(body), "records; at the stack:", [(int(l.split(b"\t")[1]), l.split(b"DP=")[1].split(b";")[0].decode()) for l in near])
