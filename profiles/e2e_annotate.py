#!/usr/bin/env python3
"""BASELINE config 5 in small, whole programs: tumour = normal's genome and germline indels + one somatic indel per
50 kb; discovery on the tumour BAM, then annotate mode (-q 0 -a -e 1) of that VCF against the normal BAM.  Product
driver on the GPU vs the reference binary compiled in place; both outputs compared byte for byte."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from indelminer_amd import bamwrite, build, synth  # noqa: E402

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
with tempfile.TemporaryDirectory() as td:
    n_reads = {}
    for name, kw in (("normal", {}), ("tumor", dict(read_seed=55, somatic_spacing=50_000))):
        refs, rd = synth.simulate(seed=4, ref_len=ref_len, coverage=30, n_contigs=n_contigs, **kw)
        contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
        bamwrite.write_fasta(td + "/ref.fa", contigs, refs)
        bamwrite.write_bam(td + "/%s.bam" % name, contigs, rd)
        n_reads[name] = rd.n
    open(td + "/cfg.txt", "w").write("IL generic 300 700\n")
    print("reads: tumour %d, normal %d" % (n_reads["tumor"], n_reads["normal"]), flush=True)
    res = {}
    for who, binary in (("product", build.HOST_BIN), ("reference", ref_bin)):
        if not os.path.exists(binary):
            continue
        t = time.perf_counter()
        d = subprocess.run([binary, "-i", "cfg.txt", "ref.fa", "t=tumor.bam"], cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t1 = time.perf_counter() - t
        open(td + "/%s_tumor.vcf" % who, "wb").write(d.stdout)
        t = time.perf_counter()
        a = subprocess.run([binary, "-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "%s_tumor.vcf" % who, "normal=normal.bam"],
                           cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        t2 = time.perf_counter() - t
        body = [l for l in a.stdout.splitlines() if not l.startswith(b"#")]
        res[who] = (d.stdout, a.stdout, t1, t2)
        print("%-9s discovery rc %d %.2f s | annotate rc %d %.2f s = %.2f M normal reads/s | %d records, %d tagged ;normal"
              % (who, d.returncode, t1, a.returncode, t2, n_reads["normal"] / t2 / 1e6, len(body), sum(l.endswith(b";normal") for l in body)), flush=True)
    if len(res) == 2:
        print("discovery VCF identical: %s   annotated VCF identical: %s   annotate speed-up %.1fx"
              % (res["product"][0] == res["reference"][0], res["product"][1] == res["reference"][1], res["reference"][3] / res["product"][3]))
