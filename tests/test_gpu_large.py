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


def test_large_config5_tumor_normal_matches_reference_digests(tmp_path):
    """BASELINE configs[4] at the size of configs[2]: discovery on the tumour, annotate mode (-q 0 -a -e 1) on the normal with the
    tumour's VCF -- both outputs against the digests of what the compiled reference printed (make_golden_large.py config5)"""
    from indelminer_amd import build
    mg = _mg()
    want = json.load(open(os.path.join(GOLD, "large_config5.json")))
    n_t, n_n = mg.materialise_tn(str(tmp_path))
    assert (n_t, n_n) == (want["tumor_reads"], want["normal_reads"])
    build.build()
    prod = build.build_host()
    p = subprocess.run([prod] + mg.TN_DISCOVER, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    got = mg.digest(p.stdout)
    for k in ("records", "composite", "insertions", "bytes", "md5"):
        assert got[k] == want["tumor"][k], (k, got[k], want["tumor"][k])
    open(str(tmp_path / "tumor.vcf"), "wb").write(p.stdout)
    a = subprocess.run([prod] + mg.TN_ANNOTATE, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0, a.stderr.decode()[-2000:]
    got = mg.digest(a.stdout)
    for k in ("records", "bytes", "md5"):
        assert got[k] == want["annotate"][k], (k, got[k], want["annotate"][k])
    assert sum(1 for l in a.stdout.splitlines() if l.endswith(b";normal")) == want["tagged_normal"]


def test_wgs_scale_reference_at_3x(tmp_path):
    """BASELINE configs[3]'s reference -- 3.0e9 bases in 24 contigs with the human length spread -- at 3x (9.0e7 reads, a 3.6 GB BAM), generated on
    the box by tests/support/simgen.c, through the product: the VCF's digest is the CPU shim's (the record-at-a-time path over the
    oracle, tests/golden/large_wgs3x.json, made by tests/golden/make_golden_wgs.py); the same bytes with the insert lengths estimated in the same pass and with the contigs
    cut into many more pieces.  (At 30x this input is profiles/wgs_run.py: minutes of GPU-box time, not a test.)"""
    import hashlib
    from indelminer_amd import build
    want = json.load(open(os.path.join(GOLD, "large_wgs3x.json")))
    gen = os.path.join(ROOT, "tests", "support", "simgen")
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-pthread", "-o", gen, gen + ".c", "-lz", "-lm"])
    out = subprocess.run([gen, "--prefix", str(tmp_path / "w"), "--threads", "16"] + want["simgen"], stdout=subprocess.PIPE, check=True)
    info = json.loads(out.stdout.decode())
    assert info["records"] == want["records_in_bam"] and info["bam_bytes"] == want["bam_bytes"]
    build.build()
    prod = build.build_host()
    for flags, env in ((["-i", "w.cfg"], {}), ([], {}), ([], {"INDELMINER_ONEPASS": "0"}), (["-i", "w.cfg"], {"INDELMINER_PIECE_BYTES": "3000000", "INDELMINER_WALKERS": "12"})):
        p = subprocess.run([prod] + flags + ["w.fa", "s=w.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert hashlib.md5(p.stdout).hexdigest() == want["md5"], (flags, env)
        assert sum(1 for l in p.stdout.splitlines() if not l.startswith(b"#")) == want["vcf_records"]


def test_config4like_candidates_match_oracle_record_by_record(gpu_ctx):
    """the 24-contig, 2.28e9-byte reference resident in HBM at once (contig offsets beyond 2^31, the longest contig beyond
    2^27 bases): device triage of every delivered record, then the realign kernel on every candidate of every contig,
    each checked against the CPU oracle on that read's own contig"""
    import numpy as np
    from indelminer_amd import capi, rawrec, synth
    from tests.support import gpucmp, oraclebind as ob
    from tests.test_gpu_triage import _compare_triage
    mg = _mg()
    refs, rd = synth.simulate(**mg.LARGE["config4like"]["sim"])
    assert len(refs) == 24 and max(len(r) for r in refs) > 2**27 and sum(len(r) for r in refs) > 2**31
    raw, off = rawrec.records(rd)
    contigs = [r.tobytes() for r in refs]
    gpu_ctx.set_reference(contigs)
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n // 4)
    tri, cand = _compare_triage(pipe, raw, off, ["generic"], [rd.range_max])
    assert len({tri[i][0].tid for i in cand}) == 24
    pipe.realign()
    pipe.sync()
    out = pipe.d_res.download(capi.RESULT_DTYPE, len(cand))
    P = ob.params()
    found = 0
    for j, i in enumerate(cand):
        t, b = tri[i]
        st, res = ob.realign(P, contigs[t.tid], len(contigs[t.tid]), t.anchor, t.range_max, b)
        msg = gpucmp.hip_vs_oracle(out[j], st, res)
        assert msg is None, (j, i, t.tid, t.anchor, msg)
        found += st == 1 and res.n_ev > 0
    assert found > 1000
