#!/usr/bin/env python3
"""config4like through the product with different walker counts (debugging aid)"""
import hashlib, importlib.util, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import build
spec = importlib.util.spec_from_file_location("mg", os.path.join(ROOT, "tests", "golden", "make_golden_large.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
with tempfile.TemporaryDirectory() as td:
    n, flags = mg.materialise("config4like", td)
    for env in ({}, {"INDELMINER_WALKERS": "6", "INDELMINER_REPLAYERS": "6"}, {"INDELMINER_ONEPASS": "1"}, {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "2", "INDELMINER_CLAIM_BASES": "1"},
                {"INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_REPLAYERS": "2"}, {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "8", "INDELMINER_CLAIM_BASES": "1"}):
        p = subprocess.run([build.HOST_BIN] + flags + ["ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        print(env, "rc", p.returncode, hashlib.md5(p.stdout).hexdigest(), p.stderr.decode().splitlines()[-1][:600], flush=True)
