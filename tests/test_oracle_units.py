"""Function-level pins of the two oracle functions that were only covered through whole-VCF goldens: imo_cluster_sr
against the reference's process_evidence and imo_sw_indel against its realign_with_indel, on vectors those functions
produced themselves (tests/golden/make_golden_units.py through oracle/_ref/librefunits.so)."""
import ctypes as C
import json
import os

import numpy as np

from tests.support import oraclebind as ob
from tests.test_oracle_golden import _cluster_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def reference_clusters(case):
    """the reference's variants by (b1, b2, class): the list it returns is in component order (sort_nodes,
    src/indelminer.c:152-193) and sort_variants orders it later; imo_cluster_sr hands the clusters over in key order"""
    vs = [((v["class"], v["start"], v["stop"]), v["members"]) for v in case["variants"]]
    assert len({k for k, _ in vs}) == len(vs)            # one cluster per (class, b1, b2)
    return sorted(vs, key=lambda x: (x[0][1], x[0][2], x[0][0]))


def oracle_clusters(case, tie_desc):
    cls, b1, b2 = case["cls"], case["b1"], case["b2"]
    order, first, count, used, k = _cluster_oracle(cls, b1, b2, case["marker"], tie_desc)
    out = []
    for f, c in zip(first, count):
        m = [int(x) for x in order[f:f + c]]
        out.append(((cls[m[0]], b1[m[0]], b2[m[0]]), m))
    return out, [int(x) for x in used]


def test_cluster_sr_matches_process_evidence():
    cases = json.load(open(os.path.join(GOLD, "units_cluster.json")))["cases"]
    assert len(cases) >= 10
    ties = 0
    for case in cases:
        got, used = oracle_clusters(case, 0)
        want = reference_clusters(case)
        assert used == case["used"], case["seed"]
        assert got == want, (case["seed"], [x for x in zip(got, want) if x[0] != x[1]][:3])
        ties += sum(len(m) > 1 for _, m in want)
        # the other tie order (the one indelminer.expected.vcf was made with): the same clusters, members reversed
        rev, _ = oracle_clusters(case, 1)
        assert rev == [(k, m[::-1]) for k, m in want]
    assert ties > 300


def variant_window(c):
    """the window with the variant in it, as realign_with_indel builds it (src/variant.c:1259-1272) -- the caller's job at
    the im_support_batch seam (imhost.c: is_indel_supported)"""
    t = bytearray(c["contig"][c["rstart"]:c["rstop"]].encode())
    if c["is_deletion"]:
        a, b = c["vstart"] - c["rstart"], c["vstop"] - c["rstart"] - 1
        t = t[:a] + t[b:]
    else:
        a = c["vstart"] - c["rstart"]
        t = t[:a] + c["alternate"][1:].encode() + t[a:]
    return bytes(t)


def test_sw_indel_matches_realign_with_indel():
    cases = json.load(open(os.path.join(GOLD, "units_sw.json")))["cases"]
    L = ob.lib()
    kinds = set()
    for k, c in enumerate(cases):
        t = variant_window(c)
        q = c["read"][c["qstart"]:c["qstop"]].encode()
        s, i, a = C.c_int32(), C.c_int32(), C.c_int32()
        L.imo_sw_indel(t, len(t), q, len(q), C.byref(s), C.byref(i), C.byref(a))
        assert [s.value, i.value, a.value] == c["expect"], (k, c["expect"], (s.value, i.value, a.value))
        kinds.add((c["is_deletion"], c["expect"][1] == 0))
    assert len(kinds) == 4 and len(cases) >= 300
