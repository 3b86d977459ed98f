#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the REAL reference.

Run in the build container only (needs oracle/_ref/libimref.so, i.e.
`make -C oracle ref`, which compiles the reference in place from
/root/reference).  Commits data only: inputs and the reference's outputs.

  realign_testdata.json   every read of test_data/alignments.bam that reaches
                          attempt_pe_alignment at default flags (697 reads) and
                          what the reference returned for it
  realign_synth.json      seeded synthetic contigs/reads for several (-k,-g,-s,-n)
                          settings, same content
  vcf/*.vcf               stdout of the reference binary for a matrix of flags
"""
import ctypes as C
import json
import os
import random
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.support import bamlite, candidates, refbind  # noqa: E402


def ref_case(R, buf, anchor, range_max, read):
    out = R.realign(buf, anchor, range_max, read)
    if out is None:
        return None
    return [dict(cls=e["cls"], b1=e["b1"], b2=e["b2"],
                 aln1=[s[:4] for s in e["aln1"]], aln2=[s[:4] for s in e["aln2"]], aln3=[s[:4] for s in e["aln3"]])
            for e in out]


def make_testdata(R):
    td = os.path.join(HERE, "test_data")
    _, seqs = bamlite.read_fasta(os.path.join(td, "reference.fa"))
    _, _, recs = bamlite.read_bam(os.path.join(td, "alignments.bam"))
    cands = candidates.select(recs, 705)          # IL generic 202 705 (indelminer.config)
    buf = C.create_string_buffer(seqs[0].encode())
    R.set_params(6, 0, 1000, 10)
    cases = []
    for c in cands:
        cases.append(dict(qname=c["qname"], anchor=c["anchor"], range_max=c["range_max"], read=c["read"],
                          qual=c["qual"], kind=c["kind"], ref=ref_case(R, buf, c["anchor"], c["range_max"], c["read"])))
    json.dump(dict(params=dict(klength=6, numgaps=0, maxdelsize=1000, ethreshold=10), cases=cases),
              open(os.path.join(HERE, "realign_testdata.json"), "w"), separators=(",", ":"))
    print("realign_testdata.json:", len(cases), "cases,", sum(1 for c in cases if c["ref"]), "with evidence")


def synth_reads(rng, contig, n, maxdel, Rm):
    clen = len(contig)
    out = []

    def randseq(m):
        return "".join(rng.choice("ACGT") for _ in range(m))

    def mutate(s, rate):
        return "".join((rng.choice("ACGT") if rng.random() < rate else ch) for ch in s)

    while len(out) < n:
        L = rng.choice([100, 100, 100, 76, 150, 36, 250])
        anchor = rng.randint(0, clen - 1)
        p = max(0, min(clen - L - 60, anchor + rng.randint(-Rm - 200, Rm + 200)))
        typ = rng.random()
        if typ < 0.45:
            d = rng.choice([1, 2, 3, 5, 10, 20, 50, 100, 300, maxdel - 1, maxdel + 10])
            cut = rng.randint(1, L - 1)
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.75:
            d = rng.choice([1, 2, 3, 5, 10, 20, 40])
            cut = rng.randint(1, L - 1)
            ins = randseq(d) if rng.random() < 0.7 else contig[max(0, p + cut - d):p + cut]
            read = (contig[p:p + cut] + ins + contig[p + cut:p + L])[:L]
        elif typ < 0.85:
            read = contig[p:p + L]
        elif typ < 0.93:
            c1 = rng.randint(5, L // 2)
            c2 = rng.randint(L // 2, L - 5)
            read = contig[p:p + c1] + contig[p + c1 + 3:p + c2] + "ACG" + contig[p + c2:p + L]
        else:
            read = randseq(L)
        if len(read) < 20:
            continue
        read = mutate(read, rng.choice([0, 0, 0.01, 0.03, 0.08]))
        if rng.random() < 0.05:
            read = read[:10] + "N" + read[11:]
        out.append((anchor, read))
    return out


def make_synth(R):
    groups = []
    settings = [(6, 0, 1000, 10), (6, 0, 200, 10), (4, 0, 1000, 10), (5, 0, 50, 6), (8, 0, 1000, 10), (12, 0, 1000, 10),
                (6, 1, 1000, 10), (6, 2, 1000, 10), (8, 5, 1000, 10), (10, 3, 200, 15)]
    for gi, (k, g, maxdel, eth) in enumerate(settings):
        rng = random.Random(1000 + gi)
        parts = []
        clen = rng.choice([3000, 5000, 20000])
        while sum(map(len, parts)) < clen:
            m = rng.randint(50, 600)
            if rng.random() < 0.15:
                unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 6)))
                parts.append((unit * (m // len(unit) + 1))[:m])
            else:
                parts.append("".join(rng.choice("ACGT") for _ in range(m)))
        contig = "".join(parts)[:clen]
        if gi % 3 == 2:
            contig = contig[:100] + "N" * 5 + contig[105:]
        Rm = rng.choice([300, 500, 705])
        R.set_params(k, g, maxdel, max(eth, k))
        buf = C.create_string_buffer(contig.encode())
        cases = []
        for anchor, read in synth_reads(rng, contig, 120, maxdel, Rm):
            cases.append(dict(anchor=anchor, range_max=Rm, read=read, ref=ref_case(R, buf, anchor, Rm, read)))
        groups.append(dict(params=dict(klength=k, numgaps=g, maxdelsize=maxdel, ethreshold=max(eth, k)),
                           contig=contig, cases=cases))
        print("synth group", gi, (k, g, maxdel, eth), "evidence in", sum(1 for c in cases if c["ref"]), "of", len(cases))
    json.dump(groups, open(os.path.join(HERE, "realign_synth.json"), "w"), separators=(",", ":"))


def make_vcfs():
    td = os.path.join(HERE, "test_data")
    outdir = os.path.join(HERE, "vcf")
    os.makedirs(outdir, exist_ok=True)
    matrix = {
        "default_config": ["-i", "indelminer.config"],
        "default_noconfig": [],
        "detailed": ["-i", "indelminer.config", "-o", "detailed"],
        "q0": ["-i", "indelminer.config", "-q", "0"],
        "all": ["-i", "indelminer.config", "-a"],
        "e1": ["-i", "indelminer.config", "-e", "1"],
        "b40_n15": ["-i", "indelminer.config", "-b", "40", "-n", "15"],
        "s50": ["-i", "indelminer.config", "-s", "50"],
        "k8": ["-i", "indelminer.config", "-k", "8"],
        "f2": ["-i", "indelminer.config", "-f", "2"],
        "region": ["-i", "indelminer.config", "-c", "reference:1-5000"],
        "g2": ["-i", "indelminer.config", "-g", "2"],
    }
    for name, flags in matrix.items():
        cmd = [refbind.BIN] + flags + ["reference.fa", "sample=alignments.bam"]
        r = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        open(os.path.join(outdir, name + ".vcf"), "wb").write(r.stdout)
        print("vcf/%s.vcf" % name, "rc", r.returncode, len(r.stdout.splitlines()), "lines")
    # annotate mode (README.md:116)
    cmd = [refbind.BIN, "-i", "indelminer.config", "-q", "0", "-a", "-e", "1", "reference.fa",
           os.path.join(outdir, "default_config.vcf"), "normal=alignments.bam"]
    r = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    open(os.path.join(outdir, "annotate.vcf"), "wb").write(r.stdout)
    print("vcf/annotate.vcf rc", r.returncode, len(r.stdout.splitlines()), "lines")


SYNTH_E2E = {
    # name: simulate() arguments.  The BAM is regenerated from the seed wherever the test runs.
    "synth_2ctg_composite": dict(seed=3, ref_len=200_000, coverage=30, n_contigs=2, big_every=5),
    "synth_1mb_30x": dict(seed=1, ref_len=1_000_000, coverage=30, n_contigs=1, big_every=7),
}

# BASELINE config 5 in small: tumour = normal's genome and germline indels + somatic indels
SYNTH_TN = {
    "normal": dict(seed=4, ref_len=300_000, coverage=30, n_contigs=2),
    "tumor": dict(seed=4, ref_len=300_000, coverage=30, n_contigs=2, read_seed=55, somatic_spacing=15_000),
}


def write_dataset(td, kw, prefix=""):
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(**kw)
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
    bamwrite.write_bam(td + "/%saln.bam" % prefix, contigs, rd)
    open(td + "/cfg.txt", "w").write("IL generic 300 700\n")


def make_tn_vcfs():
    """Discovery on the tumour, then annotate mode (-q 0 -a -e 1) on the normal (README.md:116)."""
    import tempfile
    outdir = os.path.join(HERE, "vcf")
    with tempfile.TemporaryDirectory() as td:
        write_dataset(td, SYNTH_TN["tumor"], "tumor_")
        write_dataset(td, SYNTH_TN["normal"], "normal_")
        r = subprocess.run([refbind.BIN, "-i", "cfg.txt", "ref.fa", "t=tumor_aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        open(os.path.join(outdir, "synth_tn_tumor.vcf"), "wb").write(r.stdout)
        open(td + "/tumor.vcf", "wb").write(r.stdout)
        a = subprocess.run([refbind.BIN, "-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "tumor.vcf", "normal=normal_aln.bam"],
                           cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        open(os.path.join(outdir, "synth_tn_annotate.vcf"), "wb").write(a.stdout)
        body = [l for l in a.stdout.splitlines() if not l.startswith(b"#")]
        print("vcf/synth_tn_tumor.vcf rc %d; vcf/synth_tn_annotate.vcf rc %d: %d records, %d tagged ;normal"
              % (r.returncode, a.returncode, len(body), sum(1 for l in body if l.endswith(b";normal"))))


def make_synth_vcfs():
    """Reference stdout on seeded synthetic BAMs (insertions, COMPOSITE calls, > READCHUNK reads)."""
    import tempfile
    from indelminer_amd import bamwrite, synth
    outdir = os.path.join(HERE, "vcf")
    for name, kw in SYNTH_E2E.items():
        with tempfile.TemporaryDirectory() as td:
            refs, rd = synth.simulate(**kw)
            contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
            bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
            bamwrite.write_bam(td + "/aln.bam", contigs, rd)
            open(td + "/cfg.txt", "w").write("IL generic 300 700\n")
            for suffix, flags in (("", ["-i", "cfg.txt"]), ("_noconfig", [])):
                r = subprocess.run([refbind.BIN] + flags + ["ref.fa", "s=aln.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                open(os.path.join(outdir, name + suffix + ".vcf"), "wb").write(r.stdout)
                print("vcf/%s%s.vcf rc %d, %d lines" % (name, suffix, r.returncode, len(r.stdout.splitlines())))


if __name__ == "__main__":
    if not refbind.available():
        sys.exit("oracle/_ref/libimref.so missing: run `make -C oracle ref` in the build container")
    R = refbind.Ref()
    make_testdata(R)
    make_synth(R)
    make_vcfs()
    make_synth_vcfs()
    make_tn_vcfs()
