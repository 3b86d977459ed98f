import os, random, subprocess, sys, tempfile, difflib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from indelminer_amd import bamwrite, build, rawrec, synth
seed=37
rng = random.Random(seed)
nc = rng.choice([1, 2, 3, 5, 7])
lens = [rng.choice([20_000, 60_000, 150_000, 400_000]) for _ in range(nc)]
cov=rng.choice([8, 20, 30, 45]); be=rng.choice([0,3,7]); sp=rng.choice([1000,2000])
refs, rd = synth.simulate(seed=seed, ref_lens=lens, coverage=cov, big_every=be, indel_spacing=sp)
def run(td, env):
    p = subprocess.run([build.HOST_BIN, "ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    return p.returncode, p.stdout
with tempfile.TemporaryDirectory() as td:
    contigs = [("c%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
    rawrec.write_bam_fast(td + "/aln.bam", contigs, rd)
    rc, want = run(td, {"INDELMINER_PIPELINE": "host"})
    for env in ({"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "3", "INDELMINER_CLAIM_BASES": "100000"},
                {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "3", "INDELMINER_CLAIM_BASES": "100000"},
                {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "1", "INDELMINER_CLAIM_BASES": "100000"},
                {"INDELMINER_WALKERS": "1", "INDELMINER_CLAIM_BASES": "100000", "INDELMINER_FLUSH_MODE": "per-flush"},
                {"INDELMINER_WALKERS": "1", "INDELMINER_CLAIM_BASES": "100000", "INDELMINER_VERIFY_TRIAGE": "1"},
                {"INDELMINER_WALKERS": "1", "INDELMINER_CLAIM_BASES": "100000", "AMD_SERIALIZE_KERNEL": "3"},
                {"INDELMINER_WALKERS": "4"}, {}):
        rc, got = run(td, env)
        same = got == want
        print(env, rc, same, flush=True)
        if not same:
            a = want.decode().splitlines(); b = got.decode().splitlines()
            d = [l for l in difflib.unified_diff(a, b, lineterm="", n=0) if not l.startswith(("---", "+++", "@@"))]
            print("   %d diff lines; first: %s" % (len(d), " | ".join(x[:160] for x in d[:4])), flush=True)
