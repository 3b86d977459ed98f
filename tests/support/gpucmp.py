"""Field-by-field comparison of a HIP result record with the CPU oracle's result."""
from . import oraclebind as ob

_ST = {1: 1, 0: 0, -1: -1, -2: -2}


def hip_vs_oracle(rec, st, res, check_bands=True):
    """rec: one element of capi.RESULT_DTYPE; st/res: oracle status and Result."""
    if int(rec["status"]) != _ST.get(st, st):
        return "status hip %d oracle %d" % (rec["status"], st)
    if st < 0:
        return None
    if check_bands:
        nb = res.n_band
        if int(rec["n_band"]) != nb:
            return "n_band hip %d oracle %d" % (rec["n_band"], nb)
        for b in range(nb):
            h = rec["band"][b]
            o = res.piece[b]
            got = (int(h["r1"]), int(h["r2"]), int(h["q1"]), int(h["q2"]))
            want = (o.r1, o.r2, o.q1, o.q2)
            if got != want:
                return "band %d (r1,r2,q1,q2) hip %r oracle %r (low hip %d oracle %d)" % (b, got, want, h["low"], o.low)
            if int(h["low"]) != o.low and want != (0, 0, 0, 0):
                return "band %d low hip %d oracle %d" % (b, h["low"], o.low)
            if (int(h["win_bytes"]), int(h["piece_bytes"])) != (res.win_bytes[b], res.piece_bytes[b]):
                return "band %d bytes hip %r oracle %r" % (b, (h["win_bytes"], h["piece_bytes"]), (res.win_bytes[b], res.piece_bytes[b]))
    if st != 1:
        return None
    if int(rec["ref_start"]) != res.ref_start:
        return "ref_start hip %d oracle %d" % (rec["ref_start"], res.ref_start)
    hops = [int(x) for x in rec["ops"][:int(rec["n_ops"])]]
    oops = [res.ops[i] for i in range(res.n_ops)]
    if hops != oops:
        return "ops hip %r oracle %r" % ([(w >> 4, w & 15) for w in hops], [(w >> 4, w & 15) for w in oops])
    if int(rec["n_ev"]) != res.n_ev:
        return "n_ev hip %d oracle %d" % (rec["n_ev"], res.n_ev)
    for k in range(res.n_ev):
        h = rec["ev"][k]
        o = res.ev[k]
        for f in ("cls", "b1", "b2", "seg", "read_off", "lflank", "rflank", "nd_print", "nd_filter"):
            if int(h[f]) != getattr(o, f):
                return "ev[%d].%s hip %d oracle %d" % (k, f, h[f], getattr(o, f))
    return None
