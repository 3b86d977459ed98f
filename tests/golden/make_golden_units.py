#!/usr/bin/env python3
"""Function-level golden vectors made by the REAL reference's own static functions, called through
oracle/_ref/librefunits.so (oracle/ref_unit_pe.c / ref_unit_sw.c: the reference's indelminer.c / variant.c compiled
from where they lie plus one exported wrapper each; build container only):

  units_cluster.json  process_evidence (src/indelminer.c:117-209) on lists of split-read evidence: clusters in the order
                      the function returns them, members in its order, the isused flags; with and without a marker
  units_sw.json       realign_with_indel (src/variant.c:1246-1424): the three counts check_for_indel thresholds

  python tests/golden/make_golden_units.py
"""
import ctypes as C
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librefunits.so"))


def cluster_case(seed, n, span, marker):
    """split-read evidence the way realignment emits it: a few dozen sites hit several times (ties inside a cluster),
    insertions (b1 == b2) and deletions sharing a b1, scattered singletons"""
    rng = np.random.default_rng(seed)
    nsite = max(1, n // 4)
    sb1 = rng.integers(0, span, nsite)
    slen = np.where(rng.random(nsite) < 0.3, 0, rng.integers(1, 60, nsite))
    pick = rng.integers(0, nsite, n)
    b1 = sb1[pick].astype(np.int32)
    b2 = (sb1[pick] + slen[pick]).astype(np.int32)
    noise = rng.random(n) < 0.25
    b1[noise] = rng.integers(0, span, int(noise.sum()))
    b2[noise] = b1[noise] + rng.integers(0, 40, int(noise.sum()))
    cls = (b2 > b1).astype(np.int32)
    # same breakpoints, other class: a deletion and an insertion evidence never share a cluster (src/graph.c:122-127)
    flip = rng.random(n) < 0.03
    cls[flip & (b2 > b1)] = 1
    return cls, b1, b2, marker


def run_cluster(cls, b1, b2, marker):
    n = len(cls)
    z = np.zeros(max(n, 1), np.int32)
    out = np.zeros(6 * n + 16, np.int32)
    used = np.zeros(max(n, 1), np.uint8)
    nv = C.c_int(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    w = L.imref_process_evidence(C.c_int(n), p(z), p(np.ascontiguousarray(cls)), p(np.ascontiguousarray(b1)), p(np.ascontiguousarray(b2)),
                                 p(z), p(z), p(z), C.c_int(marker), p(out), C.c_int(len(out)), C.byref(nv), p(used))
    assert w >= 0
    vs, i = [], 0
    for _ in range(nv.value):
        typ, evt, start, stop, sup = (int(x) for x in out[i:i + 5])
        vs.append({"class": typ, "evdnctype": evt, "start": start, "stop": stop, "members": [int(x) for x in out[i + 5:i + 5 + sup]]})
        i += 5 + sup
    return vs, [int(x) for x in used[:n]]


def sw_case(rng):
    """a read against its own reference span widened by the indel, the way check_for_indel calls it (src/variant.c:1536-1546)"""
    clen = int(rng.integers(400, 900))
    contig = rng.choice(list(b"ACGT"), size=clen).astype(np.uint8)
    if rng.random() < 0.2:
        contig[rng.integers(0, clen, 3)] = ord("N")
    is_del = bool(rng.random() < 0.5)
    size = int(rng.integers(1, 30))
    vstart = int(rng.integers(120, clen - 200))          # 1-based position of the base in front of the event
    if is_del:
        vstop = vstart + size + 1
        alt = bytes(contig[vstart - 1:vstart])
        sample = np.concatenate([contig[:vstart], contig[vstart + size:]])
    else:
        vstop = vstart
        ins = rng.choice(list(b"ACGT"), size=size).astype(np.uint8)
        alt = bytes(contig[vstart - 1:vstart]) + ins.tobytes()
        sample = np.concatenate([contig[:vstart], ins, contig[vstart:]])
    rl = int(rng.choice([76, 100, 150]))
    # the read as the aligner placed it WITHOUT the indel: it starts left of the event on the reference
    pos = int(rng.integers(max(0, vstart - rl + 10), vstart - 5))
    kind = rng.random()
    if kind < 0.6:
        read = sample[pos:pos + rl].copy()               # carries the variant
    elif kind < 0.85:
        read = contig[pos:pos + rl].copy()               # does not
    else:
        read = rng.choice(list(b"ACGT"), size=rl).astype(np.uint8)
    sub = rng.random(len(read)) < rng.choice([0, 0.02, 0.08])
    read[sub] = rng.choice(list(b"ACGT"), size=int(sub.sum())).astype(np.uint8)
    clip = int(rng.choice([0, 0, 5, 20]))
    qstart, qstop = clip, len(read)
    rstart = max(0, pos - size)
    rstop = min(clen, pos + rl + size)
    return dict(contig=contig.tobytes().decode(), rstart=rstart, rstop=rstop, read=read.tobytes().decode(), qstart=qstart, qstop=qstop,
                is_deletion=int(is_del), vstart=vstart, vstop=vstop, alternate=alt.decode())


def run_sw(c):
    s, i, a = C.c_int(), C.c_int(), C.c_int()
    L.imref_realign_with_indel(c["contig"].encode(), C.c_int(c["rstart"]), C.c_int(c["rstop"]), c["read"].encode(), C.c_int(c["qstart"]),
                               C.c_int(c["qstop"]), C.c_int(c["is_deletion"]), C.c_uint(c["vstart"]), C.c_uint(c["vstop"]),
                               c["alternate"].encode(), C.byref(s), C.byref(i), C.byref(a))
    return [s.value, i.value, a.value]


def main():
    cases = []
    for seed, n, span, marker in [(1, 6, 40, 2**31 - 1), (2, 40, 300, 2**31 - 1), (3, 40, 300, 150), (4, 400, 5000, 2**31 - 1),
                                  (5, 400, 5000, 2600), (6, 3000, 100000, 2**31 - 1), (7, 3000, 100000, 41000), (8, 1, 10, 2**31 - 1),
                                  (9, 200, 60, 2**31 - 1), (10, 200, 60, 30)]:
        cls, b1, b2, marker = cluster_case(seed, n, span, marker)
        vs, used = run_cluster(cls, b1, b2, marker)
        cases.append({"seed": seed, "cls": cls.tolist(), "b1": b1.tolist(), "b2": b2.tolist(), "marker": marker, "variants": vs, "used": used})
    json.dump({"made_by": "process_evidence of the reference (oracle/_ref/librefunits.so), list built with sladdhead in arrival order",
               "cases": cases}, open(os.path.join(GOLD, "units_cluster.json"), "w"))
    rng = np.random.default_rng(77)
    sw = []
    for _ in range(300):
        c = sw_case(rng)
        c["expect"] = run_sw(c)
        sw.append(c)
    json.dump({"made_by": "realign_with_indel of the reference (oracle/_ref/librefunits.so)", "cases": sw},
              open(os.path.join(GOLD, "units_sw.json"), "w"))
    print("cluster cases %d (%d variants), sw cases %d" % (len(cases), sum(len(c["variants"]) for c in cases), len(sw)))


if __name__ == "__main__":
    main()
