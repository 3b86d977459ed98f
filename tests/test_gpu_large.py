"""BASELINE configs[2] at full size through the product driver on one MI355X: 8 contigs x 6.25 Mb, 30x,
15 M reads, every seventh planted event a 150-900 bp deletion (COMPOSITE calls), crossing ~150 READCHUNK
flushes with the global read counter carried over the contigs (src/indelminer.c:617,764).  The expected
digest was made by the reference itself in the build container (tests/golden/make_golden_large.py).

config4like carries what distinguishes BASELINE configs[3] (3 Gb, 24 contigs) at a coverage the reference can finish:
24 contigs with a human-like length spread, the longest above 2^27 bases, 2.28e9 reference bytes in total -- past
2^31, so every contig offset on the device is 64-bit -- and the read counter carried over all 24 contigs."""
import hashlib
import importlib.util
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _mg():
    spec = importlib.util.spec_from_file_location("make_golden_large", os.path.join(GOLD, "make_golden_large.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["config3", "config4like"])
def test_large_config_matches_reference_digest(tmp_path, name):
    from indelminer_amd import build
    mg = _mg()
    want = json.load(open(os.path.join(GOLD, "large_%s.json" % name)))
    n, flags = mg.materialise(name, str(tmp_path))
    assert n == want["reads"]
    build.build()
    prod = build.build_host()
    p = subprocess.run([prod] + flags + ["ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = mg.digest(p.stdout)
    for k in ("records", "composite", "insertions", "bytes", "md5"):
        assert got[k] == want[k], (k, got[k], want[k])
