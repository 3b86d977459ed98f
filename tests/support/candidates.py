"""Which BAM records reach attempt_pe_alignment, and with what arguments.

Test-side restatement of fetch_func's candidate rules (src/indelminer.c:339-515,
SURVEY.md A.1) -- used to build parity inputs from a BAM.  Returns dicts with the
read bases exactly as the reference hands them over (reverse-complemented when
read and mate share an orientation, src/indelminer.c:404-409,479-484).
"""
_RC = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(_RC.get(c, "N") for c in reversed(s))


def select(recs, range_max, qthreshold=10):
    out = []
    for r in recs:
        f = r.flag
        if f & (0x100 | 0x200 | 0x400 | 0x800):
            continue
        if not (f & 0x1):
            continue
        aligned = not (f & 0x4)
        mate_aligned = not (f & 0x8)
        if aligned and mate_aligned and r.tid != r.mtid:
            continue
        is_rc = bool(f & 0x10)
        mate_rc = bool(f & 0x20)
        mq = r.aux_int("MQ")
        mmq = r.mapq if mq is None else mq
        if (not aligned) and mate_aligned:
            if mmq < qthreshold:
                continue
            seq = r.seq if mate_rc else revcomp(r.seq)
            out.append(dict(qname=r.qname, tid=r.mtid, anchor=r.mpos, range_max=range_max,
                            read=seq, qual=mmq, kind="unmapped"))
        elif aligned and mate_aligned and (f & 0x2):
            ops = [op for (_, op) in r.cigar]
            ndel = ops.count(2); nins = ops.count(1); nclip = ops.count(4)
            if ndel + nins + nclip == 0:
                continue
            three_prime = (ops[-1] == 4) if not is_rc else (ops[0] == 4)
            if (nclip == 0 or (nclip == 1 and three_prime)) and ndel == 0 and nins == 0:
                continue
            if mmq < qthreshold:
                continue
            seq = revcomp(r.seq) if (is_rc == mate_rc) else r.seq
            out.append(dict(qname=r.qname, tid=r.mtid, anchor=r.mpos, range_max=range_max,
                            read=seq, qual=r.mapq, kind="proper"))
    return out
