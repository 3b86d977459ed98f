"""Shared comparison helpers: reference evidence list vs oracle Result."""
from . import oraclebind as ob


def ref_vs_oracle(ref_out, st, res, read):
    """Returns None when identical, else a message."""
    if ref_out is None:
        if st == 1:
            return "reference NULL, oracle has %d evidence" % res.n_ev
        if st < 0:
            return "oracle status %d where the reference returned NULL" % st
        return None
    if st != 1:
        return "reference has %d evidence, oracle status %d" % (len(ref_out), st)
    segs = ob.segments(res)
    ref_out = list(reversed(ref_out))       # the per-read list is built with sladdhead (src/alignment.c:465,471)
    if len(ref_out) != res.n_ev:
        return "evidence count %d vs %d" % (len(ref_out), res.n_ev)
    for k, e in enumerate(ref_out):
        o = res.ev[k]
        if (e["cls"], e["b1"], e["b2"]) != (o.cls, o.b1, o.b2):
            return "evidence %d: ref %r oracle %r" % (k, (e["cls"], e["b1"], e["b2"]), (o.cls, o.b1, o.b2))
        full = e["aln1"] + e["aln2"] + e["aln3"]
        if len(full) != len(segs):
            return "segment count %d vs %d" % (len(full), len(segs))
        if len(e["aln1"]) != o.seg:
            return "indel segment index %d vs %d" % (len(e["aln1"]), o.seg)
        for (rop, rlen, rs, re_, rseq), (op, ln, s, en, roff) in zip(full, segs):
            if (rop, rlen, rs, re_) != (op, ln, s, en):
                return "segment ref %r oracle %r" % ((rop, rlen, rs, re_), (op, ln, s, en))
            want = "-" * ln if op == 2 else read[roff:roff + ln]
            if rseq != want:
                return "segment bases %r vs %r" % (rseq, want)
        # flank reductions as print_vcf_output would compute them (src/variant.c:217-274)
        lfl = sum(l for (op, l, *_rest) in e["aln1"] if op in (0, 7, 8, 1))
        rfl = sum(l for (op, l, *_rest) in e["aln3"] if op in (0, 7, 8, 1))
        ndp = sum(l for (op, l, *_rest) in e["aln1"] + e["aln3"] if op in (8, 1, 2))
        ndf = ndp + sum(l for (op, l, *_rest) in e["aln1"] + e["aln3"] if op == 4)
        if (lfl, rfl, ndp, ndf) != (o.lflank, o.rflank, o.nd_print, o.nd_filter):
            return "flanks ref %r oracle %r" % ((lfl, rfl, ndp, ndf), (o.lflank, o.rflank, o.nd_print, o.nd_filter))
    return None
