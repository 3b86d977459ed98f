#!/usr/bin/env python3
"""Goldens of the large BASELINE configurations, made by the REFERENCE compiled in place
(oracle/_ref/indelminer, oracle/Makefile) in the build container.  Only digests are committed
(tests/golden/large_*.json): the inputs are regenerated from their seeds wherever the tests run.

  python tests/golden/make_golden_large.py config3        # 8 x 6.25 Mb, 30x, 15 M reads (~5 min of reference time)
  python tests/golden/make_golden_large.py config4like    # 24 contigs, human-like length spread, > 2^31 reference bytes, low coverage
  python tests/golden/make_golden_large.py config5        # tumour + normal, 8 x 6.25 Mb each, discovery then annotate mode (~25 min of reference time)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from indelminer_amd import bamwrite, rawrec, synth  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "indelminer")

LARGE = {
    # BASELINE configs[2]: 50 Mb in 8 contigs, 30x 100 bp PE; every seventh planted event a 150-900 bp deletion
    "config3": dict(sim=dict(seed=2, ref_len=6_250_000, coverage=30, n_contigs=8, big_every=7), flags=[]),
    # the distinguishing properties of BASELINE configs[3] (3 Gb / 24 contigs) at low coverage: 24 contigs with a human-like
    # length spread, the longest above 2^27 bases, the reference above 2^31 bytes in total (64-bit offsets on the device),
    # the read counter carried over all 24 contigs; coverage 0.01x and sparse events keep the reference's per-candidate
    # strlen of the contig (src/alignment.c:771) affordable
    "config4like": dict(sim=dict(seed=3, ref_lens=[140_000_000 - 3_900_000 * i for i in range(24)], coverage=0.01,
                                 indel_spacing=5_000, big_every=7), flags=["-e", "1"]),      # support 1 is enough at this depth
}


# BASELINE configs[4]: tumour / normal at the size of configs[2] -- the normal is the donor genome with its germline indels, the tumour
# the same genome (same seed: same germline events) read with another read stream and with somatic indels on top, one per ~50 kb
TN = {
    "normal": dict(seed=4, ref_len=6_250_000, coverage=30, n_contigs=8),
    "tumor": dict(seed=4, ref_len=6_250_000, coverage=30, n_contigs=8, read_seed=55, somatic_spacing=50_000),
}


def materialise_tn(td):
    """ref.fa, tumor_aln.bam, normal_aln.bam (+ indexes) and cfg.txt into td; returns (tumour reads, normal reads)"""
    n = {}
    for who in ("tumor", "normal"):
        refs, rd = synth.simulate(**TN[who])
        contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
        if who == "tumor":
            bamwrite.write_fasta(os.path.join(td, "ref.fa"), contigs, refs)
        rawrec.write_bam_fast(os.path.join(td, "%s_aln.bam" % who), contigs, rd)
        n[who] = int(rd.n)
        rmax = int(rd.range_max)
        del refs, rd
    open(os.path.join(td, "cfg.txt"), "w").write("IL generic 300 %d\n" % rmax)
    return n["tumor"], n["normal"]


TN_DISCOVER = ["-i", "cfg.txt", "ref.fa", "t=tumor_aln.bam"]
TN_ANNOTATE = ["-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "tumor.vcf", "normal=normal_aln.bam"]


def make_tn():
    """discovery on the tumour, then annotate mode on the normal (README.md:116), by the compiled reference"""
    with tempfile.TemporaryDirectory() as td:
        t = time.perf_counter()
        nt_, nn_ = materialise_tn(td)
        print("config5: %d + %d reads written in %.1f s" % (nt_, nn_, time.perf_counter() - t), flush=True)
        t = time.perf_counter()
        q = subprocess.run([REF_BIN] + TN_DISCOVER, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t_disc = time.perf_counter() - t
        assert q.returncode == 0, q.stderr.decode()[-2000:]
        open(os.path.join(td, "tumor.vcf"), "wb").write(q.stdout)
        print("config5: discovery %.1f s, %d bytes" % (t_disc, len(q.stdout)), flush=True)
        t = time.perf_counter()
        a = subprocess.run([REF_BIN] + TN_ANNOTATE, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t_ann = time.perf_counter() - t
        assert a.returncode == 0, a.stderr.decode()[-2000:]
        body = [l for l in a.stdout.splitlines() if not l.startswith(b"#")]
        out = dict(tumor=digest(q.stdout), annotate=digest(a.stdout), tagged_normal=sum(1 for l in body if l.endswith(b";normal")),
                   tumor_reads=nt_, normal_reads=nn_, reference_discovery_wall_s=round(t_disc, 1), reference_annotate_wall_s=round(t_ann, 1),
                   sim=TN, made_by="oracle/_ref/indelminer (the reference's own sources, oracle/Makefile), 1 thread, build container")
        print(json.dumps(out))
        with open(os.path.join(ROOT, "tests", "golden", "large_config5.json"), "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")


def materialise(name, td):
    """writes ref.fa / aln.bam (+ .bai) for the named configuration into td; returns (n_reads, flags)"""
    cfg = LARGE[name]
    refs, rd = synth.simulate(**cfg["sim"])
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(os.path.join(td, "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(os.path.join(td, "aln.bam"), contigs, rd)
    return rd.n, cfg["flags"]


def digest(vcf_bytes):
    body = [l for l in vcf_bytes.splitlines() if not l.startswith(b"#")]
    return dict(md5=hashlib.md5(vcf_bytes).hexdigest(), records=len(body),
                composite=sum(b"COMPOSITE" in l for l in body), insertions=sum(b"INSERTION" in l for l in body),
                bytes=len(vcf_bytes))


def main():
    name = sys.argv[1]
    if name == "config5":
        return make_tn()
    with tempfile.TemporaryDirectory() as td:
        t = time.perf_counter()
        n, flags = materialise(name, td)
        print("%s: %d reads written in %.1f s" % (name, n, time.perf_counter() - t), flush=True)
        t = time.perf_counter()
        q = subprocess.run([REF_BIN] + flags + ["ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t
        assert q.returncode == 0, q.stderr.decode()[-2000:]
        out = digest(q.stdout)
        out.update(reads=int(n), reference_wall_s=round(dt, 1), flags=flags, sim=LARGE[name]["sim"],
                   made_by="oracle/_ref/indelminer (the reference's own sources, oracle/Makefile), 1 thread, build container")
        print(json.dumps(out))
        with open(os.path.join(ROOT, "tests", "golden", "large_%s.json" % name), "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    main()
