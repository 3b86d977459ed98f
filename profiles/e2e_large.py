#!/usr/bin/env python3
"""BASELINE config 3, whole programs (default: in small): 8 contigs x 1 Mb (argv[2] = 6250000 for the full 50 Mb), 100 bp pairs at 30x (2.4 M reads), every seventh
planted event a 150-900 bp deletion (PAIRED_READ / COMPOSITE calls), discovery without a config file (so both
programs also run the insert-length pass).  Product driver on the GPU vs the reference binary compiled in place
(oracle/_ref/indelminer), VCF compared byte for byte."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import bamwrite, build, rawrec, synth  # noqa: E402

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000      # 6250000 = BASELINE config 3 at full size
t = time.perf_counter()
refs, rd = synth.simulate(seed=2, ref_len=ref_len, coverage=30, n_contigs=n_contigs, big_every=7)
print("simulated %d reads on %d contigs in %.1f s" % (rd.n, n_contigs, time.perf_counter() - t), flush=True)
ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
with tempfile.TemporaryDirectory() as td:
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
    t = time.perf_counter()
    rawrec.write_bam_fast(td + "/aln.bam", contigs, rd)
    print("BAM written in %.1f s, %d bytes" % (time.perf_counter() - t, os.path.getsize(td + "/aln.bam")), flush=True)
    cmd = ["ref.fa", "s=aln.bam"]
    tp = None
    # E2E_ENVS="A=1 B=2|C=3": every set of settings in turn (alternating, so that a drifting box does not favour one)
    envsets = [dict(kv.split("=", 1) for kv in es.split()) for es in os.environ.get("E2E_ENVS", "").split("|")]
    for thr in os.environ.get("E2E_THREADS", "").split(","):     # inflate workers per reader (INDELMINER_THREADS), "" = default
        for rep in range(int(os.environ.get("E2E_REPEAT", "1")) * len(envsets)):      # the first run also pages the binary and the inputs in
            env = dict(os.environ, INDELMINER_TIMING="1", **envsets[rep % len(envsets)])
            if envsets[rep % len(envsets)]: print("settings:", envsets[rep % len(envsets)], flush=True)
            if thr:                                   # "4" = inflate workers; "4:3" = inflate workers : walkers
                env["INDELMINER_THREADS"] = thr.split(":")[0]
                if ":" in thr:
                    env["INDELMINER_WALKERS"] = thr.split(":")[1]
            t = time.perf_counter()
            p = subprocess.run([build.HOST_BIN] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t
            import hashlib
            print("product run (inflate workers %s): %.2f s  rc %d  md5 %s" % (thr or "default", dt, p.returncode, hashlib.md5(p.stdout).hexdigest()), flush=True)
            tp = dt if tp is None else min(tp, dt)
            if os.environ.get("E2E_SETUP_LINES") == "1":      # the set-up phases of every run, not only of the last one
                for l in p.stderr.decode().splitlines():
                    if l.startswith("[timing]") and ("buffers" in l or "GPU context" in l or "walk of all" in l or "read FASTA" in l):
                        print("       ", l, flush=True)
    body = [l for l in p.stdout.splitlines() if not l.startswith(b"#")]
    print("product   rc %d  %.2f s  %d VCF records (%d COMPOSITE, %d PAIRED_READ only)" %
          (p.returncode, tp, len(body), sum(b"COMPOSITE" in l for l in body), sum(b"PAIRED_READ" in l and b"COMPOSITE" not in l for l in body)), flush=True)
    for l in p.stderr.decode().splitlines():
        if l.startswith("[timing]"):
            print("   ", l)
    if os.environ.get("E2E_SKIP_REF") == "1":
        import hashlib
        print("reference skipped (E2E_SKIP_REF=1); product VCF md5 %s" % hashlib.md5(p.stdout).hexdigest())
    elif os.path.exists(ref_bin):
        t = time.perf_counter()
        q = subprocess.run([ref_bin] + cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        tq = time.perf_counter() - t
        print("reference rc %d  %.2f s   VCF identical: %s   speed-up %.1fx   product %.2f M reads/s" %
              (q.returncode, tq, q.stdout == p.stdout and q.returncode == 0, tq / tp, rd.n / tp / 1e6))
    else:
        print("reference binary not present (oracle/_ref/indelminer)")
    for l in p.stderr.decode().splitlines():
        if l.startswith("[timing]") and ("pass A" in l or "insert" in l or "GPU" in l):
            pass
