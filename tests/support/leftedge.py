"""Constructed inputs for the left-edge case of local_align's reverse pass (src/localalign.c:144-176):
the band chosen by find_best_band has no k-mer vote at all (the read piece holds no k-mer that is unique in
it), select_band then takes the diagonal nearest the anchor, and for an anchor in front of or at the
window's left edge that diagonal hangs off the window: only the last bases of the piece face window
bases, the reverse pass walks up to the window's first column and -- having no `ib > 0` guard -- looks
at the byte in front of the window.  Piece 1 hangs off only at the contig's start (left1 = 0); piece 2
hangs off whenever the first piece ends where the second window begins (its anchor argument is r1 or r2,
src/alignment.c:605-717), which is the common geometry of a read with a low-complexity tail."""
import random

LOWC = ["A", "C", "AC", "AG", "CT", "ACG", "AAT", "TTTG", "GA"]


def _tail(rng, n):
    unit = rng.choice(LOWC)
    return (unit * (n // len(unit) + 2))[:n]


def cases(seed, n=120, clen=4000, head_lowc=True):
    """Returns (contig, [dict(anchor, range_max, read)]).  Three families:
      a) anchor within the first ~90 bases of the contig, read without unique k-mers (all low complexity);
      b) read = exact copy of a contig stretch + low-complexity tail (piece 2 without votes, anchored at r1);
      c) read = low-complexity head + exact copy (piece 2 = the head, anchored at r2)."""
    rng = random.Random(seed)
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    if head_lowc:
        contig = _tail(rng, 60) + contig[60:]
    out = []
    for i in range(n):
        L = rng.choice([76, 100, 100, 150])
        fam = i % 3
        if fam == 0:
            anchor = rng.randint(0, 90)
            read = _tail(rng, L)
            if rng.random() < 0.5:          # a few bases that do occur at the contig's start
                j = rng.randint(0, 10)
                read = read[:L - 12] + contig[j:j + 12]
        elif fam == 1:
            anchor = rng.randint(300, clen - 400)
            p = anchor + rng.randint(20, 500)
            m = rng.randint(30, L - 12)
            read = contig[p:p + m] + _tail(rng, L - m)
            if rng.random() < 0.5:
                read = read[:-6] + contig[p + m:p + m + 6]
        else:
            anchor = rng.randint(700, clen - 400)
            p = anchor - rng.randint(150, 600)
            m = rng.randint(30, L - 12)
            read = _tail(rng, L - m) + contig[p:p + m]
            if rng.random() < 0.5:
                q = max(0, p - 700)
                read = contig[q:q + 6] + read[6:]
        out.append(dict(anchor=anchor, range_max=705, read=read[:L]))
    return contig, out
