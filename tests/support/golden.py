"""Loading the committed golden vectors and comparing results against them."""
import json
import os

from . import oraclebind as ob

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "golden")


def load(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def golden_vs_segments(gold, status, ref_start, ops, evs):
    """gold: the reference's evidence list (returned order) or None.
    ops: list of packed words; evs: list of (cls,b1,b2,seg).  Returns None if identical."""
    if gold is None:
        return None if status == 0 else "reference NULL, got status %d" % status
    if status != 1:
        return "reference has %d evidence, got status %d" % (len(gold), status)
    gold = list(reversed(gold))     # per-read list is built with sladdhead (src/alignment.c:465,471)
    if len(gold) != len(evs):
        return "evidence count %d vs %d" % (len(gold), len(evs))
    segs = []
    ref = ref_start
    for w in ops:
        op, ln = w & 15, w >> 4
        start = ref
        if op in (7, 8, 0, 2):
            ref += ln
        segs.append([op, ln, start, ref])
    for e, (cls, b1, b2, seg) in zip(gold, evs):
        if (e["cls"], e["b1"], e["b2"]) != (cls, b1, b2):
            return "evidence ref %r got %r" % ((e["cls"], e["b1"], e["b2"]), (cls, b1, b2))
        full = e["aln1"] + e["aln2"] + e["aln3"]
        if [list(s) for s in full] != segs:
            return "segments ref %r got %r" % (full, segs)
        if len(e["aln1"]) != seg:
            return "indel segment index %d vs %d" % (len(e["aln1"]), seg)
    return None


def oracle_case(P, contig_bytes, case):
    st, res = ob.realign(P, contig_bytes, len(contig_bytes), case["anchor"], case["range_max"], case["read"])
    ops = [res.ops[i] for i in range(res.n_ops)]
    evs = [(res.ev[i].cls, res.ev[i].b1, res.ev[i].b2, res.ev[i].seg) for i in range(res.n_ev)]
    return st, res, golden_vs_segments(case["ref"], st, res.ref_start, ops, evs)
