#!/usr/bin/env python3
"""One-off soak on the GPU box: tumour / normal pairs (normal + somatic indels), discovery on the tumour, annotate mode on the
normal (README.md:116 of the reference) -- the product's device pipeline against its own record-at-a-time host path.
    python profiles/annotate_soak.py [first_seed] [n]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import bamwrite, build, rawrec, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def run(td, args, env):
    p = subprocess.run([build.HOST_BIN] + args, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
    return p.returncode, p.stdout, p.stderr[-300:]


for seed in range(first, first + n):
    rng = random.Random(seed)
    nc = rng.choice([1, 2, 3, 4])
    kw = dict(seed=seed, ref_len=rng.choice([150_000, 300_000]), coverage=rng.choice([15, 30]), n_contigs=nc, big_every=rng.choice([0, 5]))
    with tempfile.TemporaryDirectory() as td:
      try:
        for prefix, extra in (("normal_", {}), ("tumor_", dict(read_seed=seed + 999, somatic_spacing=rng.choice([15_000, 30_000])))):
            refs, rd = synth.simulate(**dict(kw, **extra))
            contigs = [("c%d" % i, len(r)) for i, r in enumerate(refs)]
            bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
            rawrec.write_bam_fast(td + "/%saln.bam" % prefix, contigs, rd)
      except ValueError as ex:      # the simulator refuses some seeds (a planted event too close to a contig's start)
        print("seed %d skipped: %s" % (seed, ex), flush=True)
        continue
      if True:
        open(td + "/cfg.txt", "w").write("IL generic 300 700\n")
        rc, tumor, err = run(td, ["-i", "cfg.txt", "ref.fa", "t=tumor_aln.bam"], {})
        assert rc == 0, err
        open(td + "/tumor.vcf", "wb").write(tumor)
        args = ["-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "tumor.vcf", "normal=normal_aln.bam"]
        rc0, want, err0 = run(td, args, {"INDELMINER_PIPELINE": "host"})
        assert rc0 == 0, err0
        rc, got, err = run(td, args, {})
        if rc != 0 or got != want:
            print("MISMATCH seed %d rc %d: %s" % (seed, rc, err.decode(errors="replace")), flush=True)
            sys.exit(1)
        body = [l for l in want.splitlines() if not l.startswith(b"#")]
    print("seed %d ok: %d contigs, %d records, %d tagged ;normal" % (seed, nc, len(body), sum(1 for l in body if l.endswith(b";normal"))), flush=True)
