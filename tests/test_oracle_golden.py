"""The CPU oracle (oracle/im_oracle.c) against the golden vectors produced by the
real reference (tests/golden/make_golden.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

from tests.support import bamlite, golden, oraclebind as ob


def test_oracle_matches_reference_on_test_data(golden_dir):
    g = golden.load("realign_testdata.json")
    _, seqs = bamlite.read_fasta(os.path.join(golden_dir, "test_data", "reference.fa"))
    contig = seqs[0].encode()
    P = ob.params(**g["params"])
    assert len(g["cases"]) == 697           # SURVEY.md Appendix B census
    n_ev = 0
    for c in g["cases"]:
        st, res, msg = golden.oracle_case(P, contig, c)
        assert msg is None, (c["qname"], msg)
        n_ev += res.n_ev if st == 1 else 0
    assert n_ev == 443                      # 440 deletions + 3 insertions in the -d trace


def test_oracle_matches_reference_on_synthetic():
    groups = golden.load("realign_synth.json")
    assert len(groups) >= 10
    for grp in groups:
        P = ob.params(**grp["params"])
        contig = grp["contig"].encode()
        for i, c in enumerate(grp["cases"]):
            st, res, msg = golden.oracle_case(P, contig, c)
            assert msg is None, (grp["params"], i, msg)


def test_oracle_flank_fields():
    """lflank/rflank/nd_* follow print_vcf_output's loops (src/variant.c:217-274)."""
    g = golden.load("realign_testdata.json")
    P = ob.params(**g["params"])
    _, seqs = bamlite.read_fasta(os.path.join(golden.GOLDEN, "test_data", "reference.fa"))
    contig = seqs[0].encode()
    checked = 0
    for c in g["cases"]:
        if not c["ref"]:
            continue
        st, res = ob.realign(P, contig, len(contig), c["anchor"], c["range_max"], c["read"])
        for k, e in enumerate(reversed(c["ref"])):
            lfl = sum(l for (op, l, *_r) in e["aln1"] if op in (0, 7, 8, 1))
            rfl = sum(l for (op, l, *_r) in e["aln3"] if op in (0, 7, 8, 1))
            ndp = sum(l for (op, l, *_r) in e["aln1"] + e["aln3"] if op in (8, 1, 2))
            ndf = ndp + sum(l for (op, l, *_r) in e["aln1"] + e["aln3"] if op == 4)
            o = res.ev[k]
            assert (lfl, rfl, ndp, ndf) == (o.lflank, o.rflank, o.nd_print, o.nd_filter)
            checked += 1
    assert checked == 443


def _cluster_oracle(cls, b1, b2, marker, tie_desc):
    import ctypes as C
    n = len(cls)
    cls = np.ascontiguousarray(cls, dtype=np.int32); b1 = np.ascontiguousarray(b1, dtype=np.int32)
    b2 = np.ascontiguousarray(b2, dtype=np.int32)
    order = np.zeros(max(n, 1), np.int32); first = np.zeros(max(n, 1), np.int32); count = np.zeros(max(n, 1), np.int32)
    used = np.zeros(max(n, 1), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    k = ob.lib().imo_cluster_sr(C.c_int32(n), p(cls), p(b1), p(b2), C.c_int32(marker), C.c_int32(tie_desc),
                                p(order), p(first), p(count), p(used))
    return order[:n], first[:k], count[:k], used[:n], k


def test_oracle_cluster_small():
    # arrival order: a(100,105) b(50,60) c(100,105) d(100,100 ins) e(50,60) f(300,400)
    cls = [1, 1, 1, 0, 1, 1]
    b1 = [100, 50, 100, 100, 50, 300]
    b2 = [105, 60, 105, 100, 60, 400]
    order, first, count, used, k = _cluster_oracle(cls, b1, b2, 2**31 - 1, 0)
    assert k == 4
    clusters = [list(order[f:f + c]) for f, c in zip(first, count)]
    assert clusters == [[1, 4], [3], [0, 2], [5]]
    order, first, count, used, k = _cluster_oracle(cls, b1, b2, 2**31 - 1, 1)
    assert [list(order[f:f + c]) for f, c in zip(first, count)] == [[4, 1], [3], [2, 0], [5]]
    # marker: nodes only for the sorted prefix before the first b2 >= marker (src/indelminer.c:140-142)
    order, first, count, used, k = _cluster_oracle(cls, b1, b2, 105, 0)
    assert k == 2 and list(used) == [0, 1, 0, 1, 1, 0]
