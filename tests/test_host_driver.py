"""The C host driver end to end: BAM + FASTA + config in, VCF bytes out, against what the
reference binary printed for the same inputs (tests/golden/vcf/, made by make_golden.py).

Without a GPU the driver is linked against tests/shim/im_shim.c (the C ABI implemented with
the CPU oracle) -- that exercises the HOST logic: BGZF/BAM/BAI/FASTA readers, fetch_func's
dispatch rules, READCHUNK flush replay, paired-read evidence, merge, filters, VCF writer.
With a GPU (-m gpu) the product binary (linked against the HIP library) must print the
same bytes."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TD = os.path.join(GOLD, "test_data")
SHIM = os.path.join(ROOT, "tests", "shim", "indelminer_shim")

FLAG_MATRIX = {
    "default_config": ["-i", "indelminer.config"],
    "default_noconfig": [],
    "q0": ["-i", "indelminer.config", "-q", "0"],
    "all": ["-i", "indelminer.config", "-a"],
    "e1": ["-i", "indelminer.config", "-e", "1"],
    "b40_n15": ["-i", "indelminer.config", "-b", "40", "-n", "15"],
    "s50": ["-i", "indelminer.config", "-s", "50"],
    "k8": ["-i", "indelminer.config", "-k", "8"],
    "f2": ["-i", "indelminer.config", "-f", "2"],
    "region": ["-i", "indelminer.config", "-c", "reference:1-5000"],
    "g2": ["-i", "indelminer.config", "-g", "2"],
    "detailed": ["-i", "indelminer.config", "-o", "detailed"],
}


def _build_shim():
    from tests.support.shimbuild import build_shim
    return build_shim()


def _product():
    from indelminer_amd import build
    build.build()
    return build.build_host()


def _run(binary, flags, cwd, ref="reference.fa", bam="alignments.bam", env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([binary] + flags + [ref, "sample=" + bam], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout


def _golden(name):
    return open(os.path.join(GOLD, "vcf", name + ".vcf"), "rb").read()


def _synth_dir(tmp_path_factory, name):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from indelminer_amd import bamwrite, synth
    d = tmp_path_factory.mktemp(name)
    refs, rd = synth.simulate(**mg.SYNTH_E2E[name])
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(d / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(d / "aln.bam"), contigs, rd)
    (d / "cfg.txt").write_text("IL generic 300 700\n")
    return str(d)


@pytest.fixture(scope="module")
def synth_small(tmp_path_factory):
    return _synth_dir(tmp_path_factory, "synth_2ctg_composite")


@pytest.fixture(scope="module")
def synth_tn(tmp_path_factory):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    d = tmp_path_factory.mktemp("tn")
    mg.write_dataset(str(d), mg.SYNTH_TN["tumor"], "tumor_")
    mg.write_dataset(str(d), mg.SYNTH_TN["normal"], "normal_")
    return str(d)


def _annotate(binary, cwd):
    """Discovery on the tumour, then annotate mode on the normal (README.md:116)."""
    r = subprocess.run([binary, "-i", "cfg.txt", "ref.fa", "t=tumor_aln.bam"], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    open(os.path.join(cwd, "tumor.vcf"), "wb").write(r.stdout)
    a = subprocess.run([binary, "-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "tumor.vcf", "normal=normal_aln.bam"],
                       cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert a.returncode == 0, a.stderr.decode()[-2000:]
    return r.stdout, a.stdout


@pytest.fixture(scope="module")
def synth_1mb(tmp_path_factory):
    return _synth_dir(tmp_path_factory, "synth_1mb_30x")


# ---------------------------------------------------------------- host logic (CPU, shim)

@pytest.mark.parametrize("name", sorted(FLAG_MATRIX))
def test_host_logic_test_data(name):
    assert _run(_build_shim(), FLAG_MATRIX[name], TD) == _golden(name)


def test_host_logic_expected_vcf_tie_order():
    """test_data/indelminer.expected.vcf differs from this toolchain's reference build in one BF
    field, a within-cluster tie order; INDELMINER_TIE_ORDER=expected reproduces it."""
    out = _run(_build_shim(), ["-i", "indelminer.config"], TD, env={"INDELMINER_TIE_ORDER": "expected"})
    assert out == open(os.path.join(TD, "indelminer.expected.vcf"), "rb").read()


def test_host_logic_synthetic_two_contigs_composite(synth_small):
    out = _run(_build_shim(), ["-i", "cfg.txt"], synth_small, "ref.fa", "aln.bam")
    assert out == _golden("synth_2ctg_composite")
    assert out.count(b"COMPOSITE;") > 10 and out.count(b"INSERTION;") > 10
    assert _run(_build_shim(), [], synth_small, "ref.fa", "aln.bam") == _golden("synth_2ctg_composite_noconfig")


def test_host_logic_synthetic_1mb_crosses_flushes(synth_1mb):
    """300 000 reads: three READCHUNK flushes with markers (src/indelminer.c:617-670)."""
    assert _run(_build_shim(), ["-i", "cfg.txt"], synth_1mb, "ref.fa", "aln.bam") == _golden("synth_1mb_30x")


def test_host_io_readahead_is_transparent(synth_1mb, synth_tn):
    """The BGZF read-ahead ring (inflate worker threads, hostio.c) must not change a byte: whole-contig runs,
    region runs (a seek, then a run of blocks long enough to start the pool, ending mid-stream) and annotate mode
    (one fetch per known variant: seek after seek with the pool already running) with 0, 1 and 6 workers."""
    shim = _build_shim()
    runs = [(synth_1mb, ["-i", "cfg.txt"], "ref.fa", "aln.bam"),
            (synth_1mb, ["-i", "cfg.txt", "-c", "ctg0:250000-700000"], "ref.fa", "aln.bam"),
            (synth_1mb, ["-c", "ctg0:600001-990000"], "ref.fa", "aln.bam")]
    for cwd, flags, fa, bam in runs:
        outs = [_run(shim, flags, cwd, fa, bam, env={"INDELMINER_THREADS": t}) for t in ("0", "1", "6")]
        assert outs[0] == outs[1] == outs[2] and outs[0].count(b"\n") > 20, flags
    ann = []
    for t in ("0", "6"):
        os.environ["INDELMINER_THREADS"] = t
        try:
            ann.append(_annotate(shim, synth_tn))
        finally:
            del os.environ["INDELMINER_THREADS"]
    assert ann[0] == ann[1]


def test_host_logic_annotate_test_data():
    out = subprocess.run([_build_shim(), "-i", "indelminer.config", "-q", "0", "-a", "-e", "1", "reference.fa",
                          os.path.join(GOLD, "vcf", "default_config.vcf"), "normal=alignments.bam"], cwd=TD,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0 and out.stdout == _golden("annotate")


def test_host_logic_annotate_tumor_normal(synth_tn):
    """BASELINE config 5 in small: somatic indels stay untagged, germline ones are tagged either by
    re-discovery or by is_indel_supported's Smith-Waterman."""
    tumor, ann = _annotate(_build_shim(), synth_tn)
    assert tumor == _golden("synth_tn_tumor")
    assert ann == _golden("synth_tn_annotate")
    body = [l for l in ann.splitlines() if not l.startswith(b"#")]
    assert 0 < sum(1 for l in body if not l.endswith(b";normal")) < len(body)


def test_product_binary_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([_product(), "-i", "indelminer.config", "reference.fa", "s=alignments.bam"], cwd=TD,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"no CPU path" in r.stderr
    assert b"#CHROM" not in r.stdout


def test_host_parallel_walkers_give_the_single_walk(tmp_path):
    """Several contigs walked at once, flush points placed afterwards from the walk's logs (group_resolve_flushes): any
    number of walkers and any grouping of contigs into claims must print what the one-record-at-a-time host path prints
    (and the reference, where its binary is present).  Six contigs, ~360 000 reads: the READCHUNK flush points fall inside
    contigs 1, 3 and 4, discordant pairs keep the pair table busy."""
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=21, ref_len=200_000, coverage=30, n_contigs=6, big_every=4)
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    shim = _build_shim()
    want = _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"COMPOSITE") > 20
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, [], str(tmp_path), ref="ref.fa", bam="aln.bam") == want
    for walkers, claim, replayers in (("1", None, None), ("2", "1", "1"), ("6", "1", "4"), ("4", "450000", "2")):
        env = {"INDELMINER_WALKERS": walkers}
        if claim:
            env["INDELMINER_CLAIM_BASES"] = claim
        if replayers:
            env["INDELMINER_REPLAYERS"] = replayers        # groups replayed at once, printed in contig order
        assert _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env=env) == want, env
    # -o detailed numbers its blocks across the whole run: the replay stays on one thread there
    det = _run(shim, ["-o", "detailed"], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert _run(shim, ["-o", "detailed"], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_WALKERS": "4", "INDELMINER_CLAIM_BASES": "1"}) == det


def test_host_one_pass_estimates_during_the_walk(synth_small, synth_1mb, tmp_path):
    """no config file (the default; INDELMINER_ONEPASS=0 is the pre-pass of the reference's layout): no estimation pass of its own -- the walk collects the insert-length extrema, candidates' ranges and
    the pair table are applied when the last contig is in -- and the bytes of the two-pass run come out (the goldens were made by
    the reference without a config file)"""
    shim = _build_shim()
    for d_, golden in ((synth_small, "synth_2ctg_composite_noconfig"),):
        for env in ({"INDELMINER_ONEPASS": "0"}, {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "1"},
                    {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_REPLAYERS": "1"}):
            assert _run(shim, [], d_, ref="ref.fa", bam="aln.bam", env=env) == _golden(golden), env
    want = _run(shim, [], synth_1mb, ref="ref.fa", bam="aln.bam")
    assert _run(shim, [], synth_1mb, ref="ref.fa", bam="aln.bam", env={"INDELMINER_ONEPASS": "0"}) == want


def test_host_replays_that_finish_out_of_order(tmp_path):
    """replay workers finish in any order; a walker must wait for the very group buffer it is about to reuse, not for a count
    of finished replays (found by profiles/pipeline_soak.py on the GPU box: one walker, three replay workers, a small group's
    replay overtaking a large one's -- the large group's output went missing).  The delay hook makes every other replay slow."""
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=61, ref_lens=[30_000, 30_000, 120_000, 30_000, 120_000, 30_000, 60_000], coverage=30, big_every=5)
    contigs = [("c%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    shim = _build_shim()
    want = _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    for walkers in ("1", "2"):
        env = {"INDELMINER_WALKERS": walkers, "INDELMINER_REPLAYERS": "3", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_DEBUG_REPLAY_DELAY_MS": "150"}
        assert _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env=env) == want, env


def test_host_unknown_read_group_exits_like_the_reference(tmp_path):
    """a read group the config file does not name: the reference exits in must_find_hashtable (src/indelminer.c:374-376); the device
    triage flags the record, the driver replays it for the reference's own message -- same exit code, same stdout (the header)"""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=5, ref_len=30_000, coverage=10)
    rd.rg_names = ["libA", "libB"]
    rd.rg_idx = (rd.pair_id % 2).astype(np.int64)
    contigs = [("ctg0", len(refs[0]))]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    (tmp_path / "cfg.txt").write_text("IL libA 300 700\n")
    outs = []
    bins = [_build_shim()]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        bins.append(ref_bin)
    for b in bins:
        r = subprocess.run([b, "-i", "cfg.txt", "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        outs.append((r.returncode, r.stdout, r.stderr.decode().strip().splitlines()[-1]))
    assert outs[0][0] == 1 and outs[0][2] == "indelminer: did not find libB in the hash"
    assert all(o == outs[0] for o in outs)


def test_host_iupac_base_in_a_plain_proper_pair_exits_like_the_reference(tmp_path):
    """the reference builds the segment list of EVERY proper pair (new_readaln, src/readaln.c:186-240) and bit2char exits on a base
    code other than A C G T N -- also in a read that is no candidate at all (100M).  Same exit code, same message, same stdout."""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=5, ref_len=30_000, coverage=10)
    idx = [i for i in range(rd.n) if (rd.flag[i] & 0x3) == 0x3 and rd.ncig[i] == 1][50]
    rd.seq = rd.seq.copy()
    rd.seq[idx, 10] = ord("M")
    contigs = [("ctg0", len(refs[0]))]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    code = bamwrite._SEQ_CODE.copy()
    try:
        bamwrite._SEQ_CODE[ord("M")] = 3            # IUPAC M
        bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    finally:
        bamwrite._SEQ_CODE[:] = code
    (tmp_path / "cfg.txt").write_text("IL generic 300 700\n")
    outs = []
    bins = [(_build_shim(), {}), (_build_shim(), {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "1"})]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        bins.append((ref_bin, {}))
    for b, env in bins:
        r = subprocess.run([b, "-i", "cfg.txt", "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        outs.append((r.returncode, r.stdout, r.stderr.decode().strip().splitlines()[-1]))
    assert outs[0][0] == 1 and outs[0][2] == "indelminer: Unhandled base encoding : 3:3"
    assert all(o == outs[0] for o in outs)


def _abort_in_a_later_group(binary, tmp_path):
    """a base code the reference refuses, in a plain proper pair of the THIRD contig, every contig a claim of its own and three
    walkers: the groups in front go out, then the run is handed to a record-at-a-time child that drops the bytes already printed
    (DESIGN.md section 4b).  Same stdout, exit status and message as the record-at-a-time run and as the reference."""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=9, ref_len=40_000, coverage=12, n_contigs=4, big_every=5)
    idx = [i for i in range(rd.n) if rd.tid[i] == 2 and (rd.flag[i] & 0x3) == 0x3 and rd.ncig[i] == 1]
    idx = idx[len(idx) // 2]
    rd.seq = rd.seq.copy()
    rd.seq[idx, 10] = ord("M")
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    code = bamwrite._SEQ_CODE.copy()
    try:
        bamwrite._SEQ_CODE[ord("M")] = 3
        bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    finally:
        bamwrite._SEQ_CODE[:] = code
    (tmp_path / "cfg.txt").write_text("IL generic 300 700\n")
    shim = binary
    runs = [(shim, {"INDELMINER_PIPELINE": "host"}), (shim, {"INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "3"}),
            (shim, {"INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "2", "INDELMINER_REPLAYERS": "1"}), (shim, {})]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        runs.append((ref_bin, {}))
    outs = []
    for b, env in runs:
        r = subprocess.run([b, "-i", "cfg.txt", "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env))
        outs.append((r.returncode, r.stdout, r.stderr.decode().strip().splitlines()[-1]))
    assert outs[0][0] == 1 and outs[0][2] == "indelminer: Unhandled base encoding : 3:3"
    body = [l for l in outs[0][1].splitlines() if not l.startswith(b"#")]
    assert len(body) > 20 and {l.split(b"\t")[0] for l in body} >= {b"ctg0", b"ctg1"}      # two whole contigs and a part of the third
    assert all(o == outs[0] for o in outs)


def test_host_abort_in_a_later_group_prints_what_the_reference_has_printed(tmp_path):
    _abort_in_a_later_group(_build_shim(), tmp_path)


@pytest.mark.gpu
def test_product_abort_in_a_later_group_prints_what_the_reference_has_printed(tmp_path):
    _abort_in_a_later_group(_product(), tmp_path)


def _many_indels_in_one_read(binary, tmp_path, envs, expect_handoff=False):
    """a proper-pair read whose CIGAR carries FIVE insertions / deletions that pass check_variants' end-distance rule
    (src/indelminer.c:285-337, no bound there): the kernels hold IM_MAX_EV = 4 per read (IM_REC_ERR_LIMIT), the run goes to the
    record-at-a-time child, which has no such bound -- the reference's output comes out whole"""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=14, ref_len=60_000, coverage=12, n_contigs=3, big_every=5)
    width = 11
    op = np.zeros((rd.n, width), dtype=rd.cig_op.dtype); ln = np.zeros((rd.n, width), dtype=rd.cig_len.dtype)
    op[:, :rd.cig_op.shape[1]] = rd.cig_op; ln[:, :rd.cig_len.shape[1]] = rd.cig_len
    idx = [i for i in range(rd.n) if rd.tid[i] == 1 and (rd.flag[i] & 0x3) == 0x3 and rd.ncig[i] == 1 and 20_000 < rd.pos[i] < 30_000][7]
    # 100 read bases: 16M 1I 16M 2D 16M 1I 16M 2D 16M 1I 17M
    ops = [(0, 16), (1, 1), (0, 16), (2, 2), (0, 16), (1, 1), (0, 16), (2, 2), (0, 16), (1, 1), (0, 17)]
    assert sum(l for o, l in ops if o != 2) == rd.seq.shape[1]
    for j, (o, l) in enumerate(ops):
        op[idx, j] = o; ln[idx, j] = l
    rd.cig_op, rd.cig_len = op, ln
    rd.ncig = rd.ncig.copy(); rd.ncig[idx] = len(ops)
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    (tmp_path / "cfg.txt").write_text("IL generic 300 700\n")
    d = str(tmp_path)
    want = _run(_build_shim(), ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam") == want
    assert want.count(b"\nctg1\t") > 5 and want.count(b"\nctg2\t") > 5
    for env in envs:
        assert _run(binary, ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam", env=env) == want, env
    if expect_handoff:
        assert b"[handoff]" in _stderr_of(binary, ["-i", "cfg.txt"], d, env={"INDELMINER_DEBUG_HANDOFF": "1"})


def test_host_read_with_more_indels_than_the_kernels_hold(tmp_path):
    _many_indels_in_one_read(_build_shim(), tmp_path, ({}, {"INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "3"}))


def test_host_leaked_handoff_variable_drops_nothing(synth_small):
    """INDELMINER_SKIP_STDOUT is how a hand-over child learns how many bytes its parent has printed; found in the environment of a
    run that nobody handed anything to (no INDELMINER_HANDOFF_PARENT naming the parent process), it must not eat output"""
    shim = _build_shim()
    want = _golden("synth_2ctg_composite")
    for env in ({"INDELMINER_SKIP_STDOUT": "5000"}, {"INDELMINER_SKIP_STDOUT": "5000", "INDELMINER_HANDOFF_PARENT": "1"}):
        assert _run(shim, ["-i", "cfg.txt"], synth_small, ref="ref.fa", bam="aln.bam", env=env) == want


def _speculation_fails_dir(tmp_path):
    """a proper pair in the last third of the contig with an insert size beyond every one in the first pieces: the insert-length table
    made from the first claims does not hold for the whole file"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=33, ref_len=300_000, coverage=20, n_contigs=2, big_every=4)
    proper = ((rd.flag & 0x3) == 0x3) & (rd.tid == 1) & (rd.pos > 200_000) & (rd.isize > 0)
    i = int(np.nonzero(proper)[0][5])
    mate = int(np.nonzero((rd.pair_id == rd.pair_id[i]) & (np.arange(rd.n) != i))[0][0])
    rd.isize = rd.isize.copy()
    rd.isize[i] = int(rd.isize.max()) + 40
    rd.isize[mate] = -int(rd.isize[i])
    contigs = [("ctg%d" % k, len(r)) for k, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    return str(tmp_path)


def _one_pass_speculation(binary, tmp_path):
    """One pass without a config file stages its groups behind the walk on the table made from the first claims and holds its output
    back; the table of the whole file decides at the end.  When it differs the run is taken again with the pre-pass: the bytes of the
    pre-pass run either way, and of the reference."""
    d = _speculation_fails_dir(tmp_path)
    want = _run(_build_shim(), [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, [], d, ref="ref.fa", bam="aln.bam") == want
    small = {"INDELMINER_PIECE_BYTES": "60000", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "2", "INDELMINER_TIMING": "1",
             "INDELMINER_SPECULATE": "1"}         # by itself only on inputs of gigabytes
    e = dict(os.environ, **small)
    r = subprocess.run([binary, "ref.fa", "s=aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode == 0 and r.stdout == want, r.stderr.decode()[-1500:]
    assert b"provisional insert lengths did not hold" in r.stderr
    for env in ({}, {"INDELMINER_ONEPASS": "0"}, dict(small, INDELMINER_SPECULATE="0")):
        assert _run(binary, [], d, ref="ref.fa", bam="aln.bam", env=env) == want, env
    # and a speculation that holds: the same input without the late large insert (synth_small's goldens), pieces for two walkers
    return small


def test_host_one_pass_speculation(tmp_path, synth_small, synth_1mb):
    shim = _build_shim()
    small = _one_pass_speculation(shim, tmp_path)
    for d, golden in ((synth_small, "synth_2ctg_composite_noconfig"), (synth_1mb, "synth_1mb_30x_noconfig")):
        r = subprocess.run([shim, "ref.fa", "s=aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **small))
        assert r.returncode == 0 and r.stdout == _golden(golden)
        if b"did not hold" not in r.stderr:
            assert b"provisional insert lengths from the first claims" in r.stderr       # staged behind the walk, and it held


def test_host_two_walkers_meet_the_same_unknown_read_group(tmp_path):
    """No config file; a read group that occurs on counted reads but on no proper pair is missing from the estimated table and
    the reference dies AT the first such read (must_find_hashtable, src/indelminer.c:369-376) -- header and earlier flushes out.
    Here such reads lie in every piece behind the first ones, several walkers meet them at once under the speculation
    (INDELMINER_SPECULATE=1) and each wants to end the run: ONE of them hands the run over, the others wait (end_lock;
    profiles/pipeline_soak.py seed 94036 had one thread exit under the other's hand-over child).  Status and bytes of the
    record-at-a-time run -- and of the compiled reference -- every time."""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=94, ref_len=800_000, coverage=30, big_every=5)
    rd.rg_names = ["libA", "libB"]
    rd.rg_idx = np.where(((rd.flag & 0x2) == 0) & (rd.pos > 500_000), 1, 0).astype(np.int64)      # behind the first READCHUNK flush
    contigs = [("ctg0", len(refs[0]))]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    shim = _build_shim()
    one = subprocess.run([shim, "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, INDELMINER_PIPELINE="host"))
    assert one.returncode == 1 and one.stderr.decode().strip().splitlines()[-1] == "indelminer: did not find libB in the hash"
    assert one.stdout.startswith(b"##fileformat")     # the header (and what the flushes in front of the read printed) is out
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        r = subprocess.run([ref_bin, "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert (r.returncode, r.stdout) == (one.returncode, one.stdout)
    env = dict(os.environ, INDELMINER_SPECULATE="1", INDELMINER_WALKERS="4", INDELMINER_PIECE_BYTES="60000", INDELMINER_CLAIM_BASES="1")
    for attempt in range(6):
        r = subprocess.run([shim, "ref.fa", "s=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert (r.returncode, r.stdout) == (one.returncode, one.stdout), (attempt, r.returncode, r.stderr.decode()[-800:])


def test_host_contigs_without_reads(tmp_path):
    """contigs that deliver no record at all -- the first, one in the middle, the last -- between contigs that do: claims,
    groups and flush placement must not mind (the read counter and the marker floor simply pass through them)"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=31, ref_len=150_000, coverage=30, n_contigs=5, big_every=4)
    keep = np.isin(rd.tid, [1, 3])
    for name, col in list(vars(rd).items()):
        if isinstance(col, np.ndarray) and len(col) == len(keep):
            setattr(rd, name, col[keep])
    rd.n = int(keep.sum())
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    shim = _build_shim()
    want = _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 50
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, [], str(tmp_path), ref="ref.fa", bam="aln.bam") == want
    for env in ({}, {"INDELMINER_WALKERS": "5", "INDELMINER_CLAIM_BASES": "1"}, {"INDELMINER_WALKERS": "1"},
                {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "5", "INDELMINER_CLAIM_BASES": "1"}):
        assert _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env=env) == want, env


def test_host_read_groups_estimated_by_several_threads(tmp_path):
    """no config file: the insert-length table is estimated -- by several threads over different contigs -- and must list the
    read groups in the order ONE process meets them (the table's prefix-match / last-hit look-up depends on insertion
    order: "li" is a prefix of "lib1" is a prefix of "lib10"); records without an RG tag are "generic".  Pipeline with
    several walkers == one-record-at-a-time host path == the reference."""
    import numpy as np
    from indelminer_amd import bamwrite, synth
    refs, rd = synth.simulate(seed=41, ref_len=60_000, coverage=25, n_contigs=4, big_every=4)
    rd.rg_names = ["lib10", "li", "", "lib1"]
    # a pair shares its read group; the groups first appear in different contigs and in an order that is not alphabetical
    first_tid = {0: 0, 1: 0, 2: 1, 3: 2}
    g = (rd.pair_id % 4).astype(np.int64)
    for k, t in first_tid.items():
        g[(g == k) & (rd.tid < t)] = 0
    rd.rg_idx = g
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    shim = _build_shim()
    want = _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host", "INDELMINER_ESTIMATE_SERIAL": "1"})
    assert want.count(b"\n") > 40
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, [], str(tmp_path), ref="ref.fa", bam="aln.bam") == want
    for env in ({}, {"INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1"}, {"INDELMINER_PIPELINE": "host"},
                {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1"}):
        assert _run(shim, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env=env) == want, env

    # The table itself (stderr): "lib1" is first met after "lib10", shares its hash bin and is a prefix of it -- the sequential
    # look-up never gives it an entry, its sizes widen lib10's range (src/bamoperations.c:48-57, src/hashtable.c:62-81).
    def table(binary, env):
        r = subprocess.run([binary, "ref.fa", "sample=aln.bam"], cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0
        lines = r.stderr.decode().splitlines()
        at = next(i for i, l in enumerate(lines) if l.startswith("Read-group"))
        rows = []
        for l in lines[at + 1:]:
            f = l.split("\t")
            if len(f) == 3 and f[1].lstrip("-").isdigit():
                rows.append(tuple(f))
            elif rows:
                break
        return sorted(rows)
    serial = table(shim, {"INDELMINER_PIPELINE": "host", "INDELMINER_ESTIMATE_SERIAL": "1"})
    assert [r[0] for r in serial] == ["generic", "li", "lib10"]
    if os.path.exists(ref_bin):
        assert table(ref_bin, {}) == serial
    for env in ({}, {"INDELMINER_WALKERS": "3"}, {"INDELMINER_ONEPASS": "0"}):
        assert table(shim, env) == serial, env


def _stale_dir(tmp_path, ref_len=250_000):
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=51, ref_len=ref_len, coverage=30, n_contigs=4, big_every=3)
    both = ((rd.flag & 0x4) == 0) & ((rd.flag & 0x8) == 0)
    second_of_discordant = both & ((rd.flag & 0x2) == 0) & (rd.pos > rd.mpos)
    drop = second_of_discordant & (rd.pair_id % 3 == 0) & np.isin(rd.tid, [0, 2])
    assert drop.sum() > 20
    keep = ~drop
    for name, col in list(vars(rd).items()):
        if isinstance(col, np.ndarray) and len(col) == len(keep):
            setattr(rd, name, col[keep])
    rd.n = int(keep.sum())
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    return str(tmp_path)


def _apply_keep(rd, keep):
    import numpy as np
    for name, col in list(vars(rd).items()):
        if isinstance(col, np.ndarray) and len(col) == len(keep):
            setattr(rd, name, col[keep])
    rd.n = int(keep.sum())


def _many_waiting_dir(tmp_path, ref_len=250_000):
    """contig 0 leaves more than forty first mates waiting in the pair table's bookkeeping whose |isize| does NOT pass
    src/indelminer.c:519 (ordinary pairs made 'not proper', second mate removed), all in front of the few that do pass
    and really pin the later markers: a summary that keeps 'the 32 smallest starts' loses the real ones"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=52, ref_len=ref_len, coverage=30, n_contigs=4, big_every=3)
    both = ((rd.flag & 0x4) == 0) & ((rd.flag & 0x8) == 0)
    proper_first = both & ((rd.flag & 0x2) != 0) & (rd.tid == 0) & (rd.pos < rd.mpos) & (rd.pos < 30_000) & (((rd.flag & 0x10) != 0) != ((rd.flag & 0x20) != 0))
    pick = np.nonzero(proper_first)[0][::7][:60]
    assert len(pick) >= 45
    ids = set(rd.pair_id[pick].tolist())
    in_pick = np.isin(rd.pair_id, list(ids)) & (rd.tid == 0)
    rd.flag = np.where(in_pick, rd.flag & ~0x2, rd.flag).astype(rd.flag.dtype)
    drop = in_pick & (rd.pos > rd.mpos)
    real = both & ((rd.flag & 0x2) == 0) & ~in_pick & (rd.pos > rd.mpos) & (rd.pair_id % 2 == 0) & np.isin(rd.tid, [0, 1]) & (rd.mpos > 40_000)
    assert real.sum() >= 5
    _apply_keep(rd, ~(drop | real))
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    open(str(tmp_path / "cfg.txt"), "w").write("IL generic 300 %d\n" % rd.range_max)
    return str(tmp_path)


def _shared_names_dir(tmp_path, ref_len=200_000):
    """a first mate left waiting in contig 0 and a complete discordant pair under the SAME name in contig 2: the reference's
    one pair table (readpairs is never reset) hands the old entry to the new pair's second mate"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=53, ref_len=ref_len, coverage=30, n_contigs=3, big_every=3)
    both = ((rd.flag & 0x4) == 0) & ((rd.flag & 0x8) == 0)
    disc = both & ((rd.flag & 0x2) == 0) & (((rd.flag & 0x10) != 0) != ((rd.flag & 0x20) != 0)) & (np.abs(rd.isize) > rd.range_max)
    a = np.nonzero(disc & (rd.tid == 0) & (rd.pos < rd.mpos))[0]
    c = np.nonzero(disc & (rd.tid == 2) & (rd.pos < rd.mpos))[0]
    assert len(a) >= 3 and len(c) >= 3
    lost, twin = int(rd.pair_id[a[1]]), int(rd.pair_id[c[1]])
    drop = (rd.pair_id == lost) & (rd.tid == 0) & (rd.pos > rd.mpos)        # contig 0: the second mate never comes
    rd.pair_id = np.where((rd.pair_id == twin) & (rd.tid == 2), lost, rd.pair_id).astype(rd.pair_id.dtype)     # contig 2: same name
    _apply_keep(rd, ~drop)
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    return str(tmp_path)


PIECE_ENVS = [{"INDELMINER_PIECE_BYTES": "150000"}, {"INDELMINER_PIECE_BYTES": "1000000", "INDELMINER_WALKERS": "3"},
              {"INDELMINER_PIECE_BYTES": "40000", "INDELMINER_WALKERS": "6", "INDELMINER_REPLAYERS": "4"},
              {"INDELMINER_PIECE_BYTES": "300000", "INDELMINER_ONEPASS": "1"}, {"INDELMINER_PIECE_BYTES": "300000", "INDELMINER_FLUSH_MODE": "seq"},
              {"INDELMINER_PIECE_BYTES": "500000", "INDELMINER_WALKERS": "1", "INDELMINER_THREADS": "0"}]


def test_host_fasta_read_in_parallel(synth_small, tmp_path):
    """the memory-mapped parallel FASTA reader (large references) against the serial one: same run, byte for byte -- also with
    lower-case bases, IUPAC codes, characters the reference's reader drops, a '>' inside a header line and blank lines"""
    shim = _build_shim()
    want = _golden("synth_2ctg_composite")
    for nt in ("2", "5", "13"):
        env = {"INDELMINER_FASTA_PARALLEL_FROM": "0", "INDELMINER_FASTA_THREADS": nt}
        assert _run(shim, ["-i", "cfg.txt"], synth_small, ref="ref.fa", bam="aln.bam", env=env) == want
    # an untidy copy of the same reference
    import shutil
    for f in ("aln.bam", "aln.bam.bai", "cfg.txt"):
        shutil.copy(os.path.join(synth_small, f), str(tmp_path / f))
    lines = open(os.path.join(synth_small, "ref.fa")).read().split("\n")
    out = []
    for i, ln in enumerate(lines):
        if ln.startswith(">"):
            out.append(ln + " some > text")
        else:
            out.append((ln.lower() if i % 3 == 0 else ln) + ("" if i % 5 else " \t") )
            if i % 7 == 0:
                out.append("")
    open(str(tmp_path / "ref.fa"), "w").write("\n".join(out))
    serial = _run(shim, ["-i", "cfg.txt"], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_FASTA_THREADS": "1"})
    assert serial == want
    assert _run(shim, ["-i", "cfg.txt"], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_FASTA_PARALLEL_FROM": "0", "INDELMINER_FASTA_THREADS": "6"}) == want


def test_host_contigs_walked_in_pieces(synth_small, synth_1mb):
    """A contig is cut into pieces (records by start position) that walkers take at the same time; the pair table, the read
    counter and the evidence no flush has consumed yet carry over from piece to piece on the main thread: the bytes of the whole-contig
    run whatever the piece size -- pieces of a few hundred reads up to a third of the contig, READCHUNK flushes inside and across
    them, COMPOSITE calls whose mates lie in different pieces; also with the insert lengths estimated by the same pass"""
    shim = _build_shim()
    for d, golden, flags in ((synth_1mb, "synth_1mb_30x", ["-i", "cfg.txt"]), (synth_1mb, "synth_1mb_30x_noconfig", []),
                             (synth_small, "synth_2ctg_composite", ["-i", "cfg.txt"]), (synth_small, "synth_2ctg_composite_noconfig", [])):
        want = _golden(golden)
        for env in (PIECE_ENVS if d is synth_small else PIECE_ENVS[::2] if flags else PIECE_ENVS[1::2]):     # the 1 Mb input: three settings per mode
            if flags and "INDELMINER_ONEPASS" in env:
                continue
            assert _run(shim, flags, d, ref="ref.fa", bam="aln.bam", env=env) == want, (golden, env)


def test_host_region_runs_take_the_pipeline(synth_1mb, synth_small):
    """-c: the pipeline over the pieces of one stretch -- its first piece takes the reads that reach into it from the front (bam_fetch),
    mates outside the stretch are looked up in the file (find_mate_rln), DP= comes from the file around each variant -- against the
    record-at-a-time path and, where it is present, the compiled reference"""
    shim = _build_shim()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    cases = [(synth_1mb, "ctg0:200,001-640,000"), (synth_1mb, "ctg0:777000"), (synth_small, "ctg1:50000-250000"), (synth_small, "ctg0")]
    for d, region in cases:
        for flags in (["-i", "cfg.txt", "-c", region], ["-c", region]):
            want = _run(shim, flags, d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
            assert want.count(b"\n") > 40
            if os.path.exists(ref_bin) and flags[0] == "-i":
                assert _run(ref_bin, flags, d, ref="ref.fa", bam="aln.bam") == want, region
            for env in (({}, {"INDELMINER_PIECE_BYTES": "40000"}) if flags[0] == "-i" else ({"INDELMINER_PIECE_BYTES": "120000", "INDELMINER_WALKERS": "3"},)):
                assert _run(shim, flags, d, ref="ref.fa", bam="aln.bam", env=env) == want, (region, flags, env)


def test_host_pieces_with_markers_pinned_low(tmp_path):
    """first mates that wait for ever pin every later marker (also of later contigs): nearly all evidence then waits for its contig's
    last flush -- the frozen entries skip the piece-to-piece chain and one entry carries their smallest key for the cuts in between;
    and a detailed run (blocks numbered across the run, replay on the main thread)"""
    d = _stale_dir(tmp_path)
    shim = _build_shim()
    want = _run(shim, [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    det = _run(shim, ["-o", "detailed"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    for env in PIECE_ENVS:
        assert _run(shim, [], d, ref="ref.fa", bam="aln.bam", env=env) == want, env
    assert _run(shim, ["-o", "detailed"], d, ref="ref.fa", bam="aln.bam", env=PIECE_ENVS[0]) == det
    d2 = _many_waiting_dir(tmp_path / "w" if (tmp_path / "w").mkdir() is None else tmp_path)
    want2 = _run(shim, ["-i", "cfg.txt"], d2, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    for env in PIECE_ENVS[:2]:
        assert _run(shim, ["-i", "cfg.txt"], d2, ref="ref.fa", bam="aln.bam", env=env) == want2, env


def test_host_names_shared_between_contigs_go_to_the_one_table(tmp_path):
    """the pipeline walks contigs independently; when a record goes through the pair table under the name of an entry an
    earlier contig left waiting, the run is handed to the record-at-a-time path, which keeps the reference's one table"""
    d = _shared_names_dir(tmp_path)
    shim = _build_shim()
    want = _run(shim, [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        # the compiled reference pairs them too -- and on this input dies of SIGSEGV while it prints the evidence it made of reads
        # of two contigs; where it survives, its bytes are the record-at-a-time path's
        r = subprocess.run([ref_bin, "ref.fa", "sample=aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode < 0 or r.stdout == want
    for env in ({}, {"INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1"}, {"INDELMINER_ONEPASS": "0"}):
        assert _run(shim, [], d, ref="ref.fa", bam="aln.bam", env=env) == want, env
    # and without the hand-over the run says why it stops instead of printing something else
    e = dict(os.environ, INDELMINER_NO_HANDOFF="1")
    r = subprocess.run([shim, "ref.fa", "s=aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode != 0 and b"shared between contigs" in r.stderr


def test_host_stale_pair_table_entries_pin_the_markers(tmp_path):
    """first mates of discordant pairs whose second mate never comes stay in the reference's pair table for the rest of the
    run and pin every later flush marker at their start (find_marker, src/indelminer.c:211-233) -- also in LATER contigs.
    The walkers keep a table per contig and hand the leftovers on as a marker floor: same output as the one-record-at-a-time
    host path and the reference, whatever the number of walkers."""
    d = _stale_dir(tmp_path)
    shim = _build_shim()
    want = _run(shim, [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 100
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, [], d, ref="ref.fa", bam="aln.bam") == want
    for env in ({}, {"INDELMINER_WALKERS": "4", "INDELMINER_CLAIM_BASES": "1"}, {"INDELMINER_WALKERS": "1"},
                {"INDELMINER_WALKERS": "2", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_FLUSH_MODE": "per-flush"},
                {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "4", "INDELMINER_CLAIM_BASES": "1"}):
        assert _run(shim, [], d, ref="ref.fa", bam="aln.bam", env=env) == want, env


def _long_read_dir(tmp_path, read_len=300, ref_len=20_000, coverage=4, seed=5):
    from indelminer_amd import bamwrite, synth
    isz = 3 * read_len
    refs, rd = synth.simulate(seed=seed, ref_len=ref_len, coverage=coverage, read_len=read_len, isize_mean=isz, isize_min=isz - 200, isize_max=isz + 200)
    contigs = [("ctg0", len(refs[0]))]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    bamwrite.write_bam(str(tmp_path / "aln.bam"), contigs, rd)
    (tmp_path / "cfg.txt").write_text("IL generic %d %d\n" % (isz - 200, isz + 200))
    return str(tmp_path)


def _long_read_library(binary, tmp_path, envs=({},)):
    """2 x 300 reads (the realign kernels' second lane layout, include/indelminer_amd.h im_expect_read_length): the reference has
    no bound on the read length (src/readaln.c:242-267), the run must print what it prints"""
    d = _long_read_dir(tmp_path, ref_len=150_000, coverage=20, seed=6)
    want = _run(_build_shim(), ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"SPLIT_READ") > 30
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert _run(ref_bin, ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam") == want
    for env in envs:
        assert _run(binary, ["-i", "cfg.txt"], d, ref="ref.fa", bam="aln.bam", env=env) == want, env
        assert _run(binary, ["-i", "cfg.txt", "-k", "9"], d, ref="ref.fa", bam="aln.bam", env=env) == \
            _run(_build_shim(), ["-i", "cfg.txt", "-k", "9"], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"}), env


def test_host_takes_a_2x300_library(tmp_path):
    _long_read_library(_build_shim(), tmp_path, envs=({}, {"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_WALKERS": "3"}))


def _beyond_the_laid_out_kernels(binary, tmp_path, envs=({},)):
    """reads beyond 1020 bases, reads beyond 255 bases with -g > 0, bands wider than a wave (-g > 60): the reference has no bound
    (src/readaln.c:242-267, src/indelminer.c:934,948) and until round 4 the driver turned such runs away at start-up; they take
    the realign kernels' general pass now (im_realign_any.hip) and the run must print what the reference prints"""
    shim = _build_shim()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    for sub, read_len, flags in (("a", 1100, []), ("b", 300, ["-g", "2"]), ("c", 150, ["-g", "80"]), ("d", 1400, ["-g", "3", "-k", "8"])):
        (tmp_path / sub).mkdir()
        d = _long_read_dir(tmp_path / sub, read_len=read_len, ref_len=60_000, coverage=12 if read_len < 1000 else 20, seed=40 + read_len)
        want = _run(shim, ["-i", "cfg.txt"] + flags, d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
        assert want.count(b"SPLIT_READ") > 5, (sub, want.count(b"SPLIT_READ"))
        if os.path.exists(ref_bin):
            assert _run(ref_bin, ["-i", "cfg.txt"] + flags, d, ref="ref.fa", bam="aln.bam") == want, sub
        for env in envs:
            assert _run(binary, ["-i", "cfg.txt"] + flags, d, ref="ref.fa", bam="aln.bam", env=env) == want, (sub, env)


def _long_read_tumour_normal(binary, tmp_path, envs=({},)):
    """tumour / normal pair of a 2 x 1100 library: discovery realigns through the general pass, annotate mode (-q 0 -a -e 1) aligns
    queries beyond 1020 bases (the support kernel's second form).  Both VCFs = the CPU shim's = the compiled reference's."""
    from indelminer_amd import bamwrite, synth
    d = str(tmp_path)
    L, isz = 1100, 3300
    base = dict(seed=77, ref_len=80_000, coverage=16, read_len=L, isize_mean=isz, isize_min=isz - 200, isize_max=isz + 200, indel_spacing=4000)
    for prefix, kw in (("tumor_", dict(base, read_seed=91, somatic_spacing=12_000)), ("normal_", base)):
        refs, rd = synth.simulate(**kw)
        contigs = [("ctg0", len(refs[0]))]
        bamwrite.write_fasta(d + "/ref.fa", contigs, refs)
        bamwrite.write_bam(d + "/%saln.bam" % prefix, contigs, rd)
    open(d + "/cfg.txt", "w").write("IL generic %d %d\n" % (isz - 200, isz + 200))
    def run_pair(b, env):
        e = dict(os.environ, **env)
        r = subprocess.run([b, "-i", "cfg.txt", "ref.fa", "t=tumor_aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        open(os.path.join(d, "tumor.vcf"), "wb").write(r.stdout)
        a = subprocess.run([b, "-i", "cfg.txt", "-q", "0", "-a", "-e", "1", "ref.fa", "tumor.vcf", "normal=normal_aln.bam"], cwd=d,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert a.returncode == 0, a.stderr.decode()[-1500:]
        return r.stdout, a.stdout
    want_t, want_a = run_pair(_build_shim(), {"INDELMINER_PIPELINE": "host"})
    body = [ln for ln in want_a.split(b"\n") if ln and not ln.startswith(b"#")]
    assert len(body) > 10 and 0 < sum(1 for ln in body if ln.endswith(b";normal")) < len(body)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert run_pair(ref_bin, {}) == (want_t, want_a)
    for env in envs:
        assert run_pair(binary, env) == (want_t, want_a), env


def test_host_long_read_tumour_normal(tmp_path):
    _long_read_tumour_normal(_build_shim(), tmp_path)


def test_host_runs_beyond_the_laid_out_kernels(tmp_path):
    _beyond_the_laid_out_kernels(_build_shim(), tmp_path, envs=({}, {"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_WALKERS": "3"}))


def _coverage_dir(tmp_path):
    """uneven coverage: a stretch of contig 0 without reads, contig 2 thinned to a third, contig 3 empty"""
    import numpy as np
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=23, ref_len=120_000, coverage=14, n_contigs=4, big_every=4)
    rng = np.random.default_rng(3)
    drop = ((rd.tid == 0) & (rd.pos >= 40_000) & (rd.pos < 60_000)) | ((rd.tid == 2) & (rng.random(rd.n) < 0.67)) | (rd.tid == 3)
    _apply_keep(rd, ~drop)
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    return str(tmp_path)


def _coverage_table(stderr):
    import re
    m = re.search(rb"ChromosomeID\tMean-coverage\n-+\t-+\n((?:\d+\t\d+\n)*)-+\t-+\n", stderr)
    return m.group(1) if m else None


def _stderr_of(binary, flags, cwd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([binary] + flags + ["ref.fa", "s=aln.bam"], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stderr


def _coverage_tables(binary, tmp_path, envs):
    """the mean-coverage table the reference prints on stderr when it has no config file (estimate_average_coverage,
    src/bamoperations.c:88-147: a pileup of every contig): here the sum of the records' reference spans over the length of their
    union, collected by whichever pass sees every record -- the table must be the reference's in every mode"""
    d = _coverage_dir(tmp_path)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    want = _coverage_table(_stderr_of(ref_bin if os.path.exists(ref_bin) else _build_shim(), [], d, env={"INDELMINER_PIPELINE": "host"}))
    rows = [int(l.split(b"\t")[1]) for l in want.splitlines()]
    assert len(rows) == 4 and rows[3] == 0 and 0 < rows[2] < rows[1] and rows[0] > 0, rows
    for env in envs:
        assert _coverage_table(_stderr_of(binary, [], d, env=env)) == want, env
    if os.path.exists(ref_bin):
        assert _coverage_table(_stderr_of(binary, ["-c", "ctg2"], d)) == _coverage_table(_stderr_of(ref_bin, ["-c", "ctg2"], d))
    (tmp_path / "cfg.txt").write_text("IL generic 300 700\nRC ctg2 41\nRC ctg0 7\n")
    assert _coverage_table(_stderr_of(binary, ["-i", "cfg.txt"], d)) == b"0\t7\n1\t0\n2\t41\n3\t0\n"
    return d, want


def test_host_coverage_table(tmp_path):
    _coverage_tables(_build_shim(), tmp_path, ({}, {"INDELMINER_ESTIMATE_SERIAL": "1"}, {"INDELMINER_ONEPASS": "0"}, {"INDELMINER_PIPELINE": "host"},
                                              {"INDELMINER_PIECE_BYTES": "60000", "INDELMINER_WALKERS": "3"},
                                              {"INDELMINER_PIECE_BYTES": "60000", "INDELMINER_WALKERS": "3", "INDELMINER_ONEPASS": "1"}))


def _large_known_indels(binary, tmp_path, envs=({},)):
    """annotate mode with split-read indels far beyond what discovery reports with the default -s: the reference realigns every
    overlapping read against its span widened by the indel on both sides (src/variant.c:1246-1312: any size up to -p), i.e. windows
    of 5 to 13 kb here -- beyond the LDS form of the support kernel (IM_MAX_SW_TARGET), which until round 4 made the driver turn
    the variant file away.  Known deletions of 2500 / 6000 bases and an insertion of 1800 in the reference's own test data."""
    from tests.support import bamlite
    _, seqs = bamlite.read_fasta(os.path.join(TD, "reference.fa"))
    seq = seqs[0]
    ins = "".join("ACGT"[(i * 7 + i // 3) % 4] for i in range(1800))
    vcf = tmp_path / "known.vcf"
    lines = ["##fileformat=VCFv4.1", "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"]
    for pos, dl in ((1000, 2500), (6000, 6000)):
        lines.append("reference\t%d\t.\t%s\t%s\t.\t.\tDELETION;SPLIT_READ;NS=3;END=%d;BP_END=%d;UTAILS=3"
                     % (pos, seq[pos - 1:pos + dl], seq[pos - 1], pos + dl, pos + dl))
    lines.append("reference\t14000\t.\t%s\t%s\t.\t.\tINSERTION;SPLIT_READ;NS=4;END=14001;BP_END=14001;UTAILS=4" % (seq[13999], seq[13999] + ins))
    # END == POS: bam_fetch over an empty interval delivers no read (reg2bins, bam_index.c:559), the variant stays unsupported
    lines.append("reference\t15000\t.\t%s\t%s\t.\t.\tINSERTION;SPLIT_READ;NS=4;END=15000;BP_END=15000;UTAILS=4" % (seq[14999], seq[14999] + ins[:30]))
    vcf.write_text("\n".join(lines) + "\n")
    flags = ["-i", "indelminer.config", "-q", "0", "-a", "-e", "1", "reference.fa", str(vcf), "normal=alignments.bam"]
    def run(b, env=None):
        r = subprocess.run([b] + flags, cwd=TD, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stderr[-600:]
        return r.stdout
    want = run(_build_shim(), {"INDELMINER_PIPELINE": "host"})
    assert len([ln for ln in want.split(b"\n") if ln and not ln.startswith(b"#")]) == 4
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "indelminer")
    if os.path.exists(ref_bin):
        assert run(ref_bin) == want
    for env in envs:
        assert run(binary, env) == want, env


def test_host_takes_large_known_indels(tmp_path):
    _large_known_indels(_build_shim(), tmp_path)



# ---------------------------------------------------------------- product binary on the GPU

@pytest.mark.gpu
def test_product_parallel_walkers_and_replayers(tmp_path):
    """the product on the GPU with 1 / 6 / default walkers, contigs one per claim or grouped, 1 / 4 replay workers, a stream
    per walker or one shared stream: always the bytes of the reference's run (six contigs, flush points inside three of them,
    groups of very different sizes through the same device buffers)"""
    from indelminer_amd import bamwrite, rawrec, synth
    refs, rd = synth.simulate(seed=21, ref_len=200_000, coverage=30, n_contigs=6, big_every=4)
    lens = [200_000, 30_000, 200_000, 8_000, 120_000, 200_000]          # uneven contigs: uneven groups
    refs = [r[:l] for r, l in zip(refs, lens)]
    import numpy as np
    keep = rd.pos + 400 < np.array(lens)[rd.tid]
    keep &= rd.mpos + 400 < np.array(lens)[rd.tid]
    for name, col in list(vars(rd).items()):
        if isinstance(col, np.ndarray) and len(col) == len(keep):
            setattr(rd, name, col[keep])
    rd.n = int(keep.sum())
    contigs = [("ctg%d" % i, len(r)) for i, r in enumerate(refs)]
    bamwrite.write_fasta(str(tmp_path / "ref.fa"), contigs, refs)
    rawrec.write_bam_fast(str(tmp_path / "aln.bam"), contigs, rd)
    want = _run(_build_shim(), [], str(tmp_path), ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"COMPOSITE") > 10
    prod = _product()
    for env in ({}, {"INDELMINER_WALKERS": "1"}, {"INDELMINER_WALKERS": "6", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_REPLAYERS": "4"},
                {"INDELMINER_WALKERS": "3", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_REPLAYERS": "1", "INDELMINER_STREAMS": "shared"},
                {"INDELMINER_WALKERS": "4", "INDELMINER_CLAIM_BASES": "250000", "INDELMINER_VERIFY_TRIAGE": "1"},
                # inputs below 64 MB put every pipeline on the context's stream: streams of their own (the large-input layout) by hand,
                # with a ring of four small chunks
                {"INDELMINER_STREAMS": "own", "INDELMINER_WALKERS": "6", "INDELMINER_CLAIM_BASES": "1", "INDELMINER_REPLAYERS": "4"},
                {"INDELMINER_STREAMS": "own", "INDELMINER_CHUNK_MB": "1", "INDELMINER_CHUNKS": "4", "INDELMINER_PIECE_BYTES": "100000"},
                {"INDELMINER_ONEPASS": "0"}, {"INDELMINER_ONEPASS": "1", "INDELMINER_WALKERS": "6", "INDELMINER_CLAIM_BASES": "1"},
                {"INDELMINER_PIECE_BYTES": "200000", "INDELMINER_WALKERS": "5"}, {"INDELMINER_PIECE_BYTES": "60000", "INDELMINER_ONEPASS": "1"},
                {"INDELMINER_PIECE_BYTES": "100000", "INDELMINER_FLUSH_MODE": "seq", "INDELMINER_THREADS": "0"}):
        assert _run(prod, [], str(tmp_path), ref="ref.fa", bam="aln.bam", env=env) == want, env


@pytest.mark.gpu
def test_product_stale_pair_table_entries(tmp_path):
    """markers pinned low by first mates that wait for ever: pending ranges grow past what a flush keeps in registers
    (flush_seq_kernel's re-reading path) and past the one-launch limit (per-flush launches); product vs the host path"""
    d = _stale_dir(tmp_path, ref_len=1_200_000)
    want = _run(_build_shim(), [], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 400
    prod = _product()
    for env in ({}, {"INDELMINER_WALKERS": "1"}, {"INDELMINER_FLUSH_MODE": "per-flush"}, {"INDELMINER_ONEPASS": "0"},
                {"INDELMINER_PIECE_BYTES": "500000"}, {"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_WALKERS": "6"}):
        assert _run(prod, [], d, ref="ref.fa", bam="aln.bam", env=env) == want, env


@pytest.mark.gpu
def test_product_contigs_walked_in_pieces_and_region_runs(synth_small, synth_1mb):
    """the product on the GPU: contigs cut into pieces of many sizes (evidence, pair table and read counter carried from piece to
    piece), and -c stretches through the same pipeline"""
    prod = _product()
    for d, golden, flags in ((synth_1mb, "synth_1mb_30x", ["-i", "cfg.txt"]), (synth_1mb, "synth_1mb_30x_noconfig", []),
                             (synth_small, "synth_2ctg_composite", ["-i", "cfg.txt"])):
        for env in PIECE_ENVS:
            if flags and "INDELMINER_ONEPASS" in env:
                continue
            assert _run(prod, flags, d, ref="ref.fa", bam="aln.bam", env=env) == _golden(golden), (golden, env)
    shim = _build_shim()
    for d, region in ((synth_1mb, "ctg0:200,001-640,000"), (synth_small, "ctg1:50000-250000")):
        want = _run(shim, ["-i", "cfg.txt", "-c", region], d, ref="ref.fa", bam="aln.bam", env={"INDELMINER_PIPELINE": "host"})
        for env in ({}, {"INDELMINER_PIECE_BYTES": "120000", "INDELMINER_WALKERS": "3"}):
            assert _run(prod, ["-i", "cfg.txt", "-c", region], d, ref="ref.fa", bam="aln.bam", env=env) == want, (region, env)


@pytest.mark.gpu
def test_product_runs_beyond_the_laid_out_kernels(tmp_path):
    _beyond_the_laid_out_kernels(_product(), tmp_path, envs=({}, {"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_WALKERS": "3"}, {"INDELMINER_PIPELINE": "host"}))


@pytest.mark.gpu
def test_product_read_with_more_indels_than_the_kernels_hold(tmp_path):
    _many_indels_in_one_read(_product(), tmp_path, ({}, {"INDELMINER_CLAIM_BASES": "1", "INDELMINER_WALKERS": "3"}, {"INDELMINER_ONEPASS": "0"}), expect_handoff=True)


@pytest.mark.gpu
def test_product_one_pass_speculation(tmp_path, synth_small, synth_1mb):
    prod = _product()
    small = _one_pass_speculation(prod, tmp_path)
    for d, golden in ((synth_small, "synth_2ctg_composite_noconfig"), (synth_1mb, "synth_1mb_30x_noconfig")):
        r = subprocess.run([prod, "ref.fa", "s=aln.bam"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **small))
        assert r.returncode == 0 and r.stdout == _golden(golden)


@pytest.mark.gpu
def test_product_coverage_table(tmp_path):
    _coverage_tables(_product(), tmp_path, ({}, {"INDELMINER_ONEPASS": "0"}, {"INDELMINER_PIECE_BYTES": "60000", "INDELMINER_WALKERS": "3"}))


@pytest.mark.gpu
def test_product_takes_a_2x300_library(tmp_path):
    _long_read_library(_product(), tmp_path, envs=({}, {"INDELMINER_PIECE_BYTES": "150000", "INDELMINER_WALKERS": "3"}, {"INDELMINER_PIPELINE": "host"}))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FLAG_MATRIX))
def test_product_test_data(name):
    assert _run(_product(), FLAG_MATRIX[name], TD) == _golden(name)


@pytest.mark.gpu
def test_product_expected_vcf():
    out = _run(_product(), ["-i", "indelminer.config"], TD, env={"INDELMINER_TIE_ORDER": "expected"})
    assert out == open(os.path.join(TD, "indelminer.expected.vcf"), "rb").read()


@pytest.mark.gpu
def test_product_synthetic(synth_small, synth_1mb):
    assert _run(_product(), ["-i", "cfg.txt"], synth_small, "ref.fa", "aln.bam") == _golden("synth_2ctg_composite")
    assert _run(_product(), [], synth_small, "ref.fa", "aln.bam") == _golden("synth_2ctg_composite_noconfig")
    assert _run(_product(), ["-i", "cfg.txt"], synth_1mb, "ref.fa", "aln.bam") == _golden("synth_1mb_30x")


@pytest.mark.gpu
def test_product_takes_a_band_wider_than_a_wave():
    """-g 200 on the reference's own test data: every candidate through the general pass, output = the CPU shim's"""
    want = _run(_build_shim(), ["-i", "indelminer.config", "-g", "200"], TD, env={"INDELMINER_PIPELINE": "host"})
    assert want.count(b"\n") > 30
    assert _run(_product(), ["-i", "indelminer.config", "-g", "200"], TD) == want



@pytest.mark.gpu
def test_product_long_read_tumour_normal(tmp_path):
    _long_read_tumour_normal(_product(), tmp_path, envs=({}, {"INDELMINER_PIPELINE": "host"}))


@pytest.mark.gpu
def test_product_takes_large_known_indels(tmp_path):
    _large_known_indels(_product(), tmp_path, envs=({}, {"INDELMINER_PIPELINE": "host"}))


@pytest.mark.gpu
def test_product_annotate(synth_tn):
    out = subprocess.run([_product(), "-i", "indelminer.config", "-q", "0", "-a", "-e", "1", "reference.fa",
                          os.path.join(GOLD, "vcf", "default_config.vcf"), "normal=alignments.bam"], cwd=TD,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert out.returncode == 0 and out.stdout == _golden("annotate")
    tumor, ann = _annotate(_product(), synth_tn)
    assert tumor == _golden("synth_tn_tumor")
    assert ann == _golden("synth_tn_annotate")


@pytest.mark.gpu
def test_product_multi_gpu_path_one_rank(synth_small, tmp_path):
    """The multi-GPU flow of the product with a world of ONE rank on the one GPU of this box: RCCL unique id through the
    rendezvous file, communicator bring-up, the pre-walk, the all-gather of the shard summaries (RCCL, one rank), one
    VCF part per contig, the closing all-gather, rank 0's concatenation.  More ranks need more GPUs (RCCL refuses two
    ranks on one device); the two- and three-rank flows run against the CPU shim in tests/test_multi_rank_driver.py."""
    for flags, gold in ((["-i", "cfg.txt"], "synth_2ctg_composite"), ([], "synth_2ctg_composite_noconfig")):
        out = _run(_product(), flags, synth_small, "ref.fa", "aln.bam",
                   env={"INDELMINER_FORCE_MGPU": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                        "INDELMINER_RENDEZVOUS": str(tmp_path / "rdv")})
        assert out == _golden(gold)
        # the path of a contig walked by several ranks: walked groups held back, ONE sum of the depth arrays over the ranks
        # (ncclAllReduce, here over one rank), then the replays
        out = _run(_product(), flags, synth_small, "ref.fa", "aln.bam",
                   env={"INDELMINER_FORCE_MGPU": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "INDELMINER_MG_FORCE_SPLIT": "1",
                        "INDELMINER_PIECE_BYTES": "150000", "INDELMINER_RENDEZVOUS": str(tmp_path / "rdv2")})
        assert out == _golden(gold)
        # the road of a group walked for another rank's contig: host part serialised and uploaded, host part + device arrays through
        # ONE ncclSend / ncclRecv group (here from the rank to itself), unpacked and staged from the arrived arrays
        out = _run(_product(), flags, synth_small, "ref.fa", "aln.bam",
                   env={"INDELMINER_FORCE_MGPU": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "INDELMINER_MG_SELF_SHIP": "1",
                        "INDELMINER_PIECE_BYTES": "150000", "INDELMINER_RENDEZVOUS": str(tmp_path / "rdv3")})
        assert out == _golden(gold)


def _deep_locus(binary, tmp_path):
    """DP= at a locus deeper than samtools' pileup buffers (bam_pileup.c:172,244): 9000 plain pairs stacked on the base in front
    of a called deletion -- the reference's pileup stops taking the stack's records at 8000 nodes and prints DP=365 where a plain
    depth count gives 409; 7900 stacked pairs stay below the cap.  Goldens: the compiled reference (make_golden_deep.py)."""
    from tests.support import deeplocus
    for name, depth in (("deep_locus_9000", 9000), ("deep_locus_7900", 7900)):
        d = tmp_path / name
        d.mkdir()
        deeplocus.write(str(d), depth=depth)
        assert _run(binary, ["-i", "cfg.txt"], str(d), "ref.fa", "aln.bam") == _golden(name), name


def test_host_logic_dp_at_a_locus_deeper_than_the_pileup_buffers(tmp_path):
    _deep_locus(_build_shim(), tmp_path)


@pytest.mark.gpu
def test_product_dp_at_a_locus_deeper_than_the_pileup_buffers(tmp_path):
    _deep_locus(_product(), tmp_path)


def _odd_inputs(binary, tmp_path, envs, seeds=None):
    """the inputs of tests/golden/odd_inputs.py regenerated from their seeds; what the compiled reference did with each is in
    tests/golden/odd_inputs.json (make_golden_odd.py): same exit status; same stdout where the reference completes"""
    import hashlib, json, shutil
    from tests.golden.odd_inputs import make_input
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "odd_inputs.json")))
    n_ok = 0
    for seed in sorted(want, key=int):
        if seeds is not None and int(seed) not in seeds:
            continue
        w = want[seed]
        d = str(tmp_path / ("odd" + seed))
        os.makedirs(d)
        cmd, _ = make_input(int(seed), d)
        assert cmd == w["cmd"], seed
        for env in envs:
            r = subprocess.run([binary] + cmd, cwd=d, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
            assert r.returncode == w["rc"], (seed, env, r.returncode, r.stderr.decode(errors="replace")[-400:])
            if w["md5"] is not None:        # also what the reference printed in front of an abort (not where a signal took its buffer)
                assert hashlib.md5(r.stdout).hexdigest() == w["md5"], (seed, env, len(r.stdout), w["bytes"], r.stderr.decode(errors="replace")[-300:])
            n_ok += w["rc"] == 0
        shutil.rmtree(d)
    return n_ok


def test_host_odd_inputs_match_the_reference(tmp_path):
    """host logic on the CPU shim; a third of the list here, the whole list on the GPU below"""
    assert _odd_inputs(_build_shim(), tmp_path, [{}, {"INDELMINER_PIPELINE": "host"}], seeds=set(range(20000, 20070, 3)) | {60037, 60058, 60233}) > 20


@pytest.mark.gpu
def test_product_odd_inputs_match_the_reference(tmp_path):
    """the product with its real kernels: records made odd in every way the reference has an opinion on, random flags, read
    groups, with and without a configuration file -- exit status and stdout as the compiled reference gave them"""
    assert _odd_inputs(_product(), tmp_path, [{}, {"INDELMINER_WALKERS": "1", "INDELMINER_REPLAYERS": "1"}]) > 80
