"""Seeded synthetic paired-end data in the shape BASELINE.json's configs name.

Host-side data tooling (numpy only, no compute path): a random reference, a
donor genome carrying planted 1-50 bp indels, 100 bp FR pairs at a stated
coverage with 0.5 % substitutions, and the alignments a BWA-like mapper would
emit for them:
  * indel with >= 20 read bases on both sides  -> CIGAR with I / D
  * otherwise the shorter side is soft-clipped
  * larger side < 30 bases                      -> read unmapped, mate mapped
MAPQ 60, MQ:i:60, no read group (-> "generic"), coordinate sorted.

`simulate()` returns every read as BAM-like columns; `candidates()` applies the
reference's own candidate rule (fetch_func, src/indelminer.c:339-515,
SURVEY.md A.1) and returns the struct-of-arrays batch the realign seam takes.
"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b

OP_M, OP_I, OP_D, OP_S = 0, 1, 2, 4


def random_genome(rng, length):
    return ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def plant_indels(rng, ref_len, spacing=2000, max_size=50, margin=1500, big_every=0):
    """Indel events on reference coordinates, about one per `spacing` bases.  With big_every = k
    every k-th event is a 150-900 bp deletion (discordant pairs -> PAIRED_READ / COMPOSITE calls)."""
    pos = np.arange(margin, ref_len - margin, spacing, dtype=np.int64)
    pos = pos + rng.integers(-spacing // 4, spacing // 4 + 1, size=len(pos))
    pos = np.clip(pos, 600, ref_len - 600) if spacing > 20000 else pos     # a wide jitter (sparse events) must not leave the contig
    size = rng.integers(1, max_size + 1, size=len(pos))
    is_ins = rng.random(len(pos)) < 0.5
    if big_every:
        big = (np.arange(len(pos)) % big_every) == (big_every - 1)
        size = np.where(big, rng.integers(150, 900, size=len(pos)), size)
        is_ins = np.where(big, False, is_ins)
    return pos, size, is_ins


def build_donor(rng, ref, pos, size, is_ins):
    """Apply the events.  Returns donor bases and, per event, its donor coordinate.  `rng` is the
    integer genome seed: inserted bases are a function of (seed, event position)."""
    pieces = []
    dpos = np.zeros(len(pos), dtype=np.int64)
    ins_seqs = []
    cur = 0
    dcur = 0
    for i in range(len(pos)):
        p = int(pos[i])
        pieces.append(ref[cur:p]); dcur += p - cur
        dpos[i] = dcur
        if is_ins[i]:
            s = ACGT[np.random.default_rng([int(rng), p]).integers(0, 4, size=int(size[i]), dtype=np.uint8)]
            ins_seqs.append(s)
            pieces.append(s); dcur += len(s)
            cur = p
        else:
            ins_seqs.append(None)
            cur = p + int(size[i])
    pieces.append(ref[cur:])
    return np.concatenate(pieces), dpos


class Reads:
    """Column store of simulated reads in coordinate-sorted (arrival) order."""
    pass


def simulate(seed, ref_len=1_000_000, coverage=30, read_len=100, isize_mean=500, isize_sd=50,
             isize_min=300, isize_max=700, sub_rate=0.005, indel_spacing=2000, n_contigs=1, big_every=0,
             read_seed=None, somatic_spacing=0, ref_lens=None):
    """Returns (refs, reads): refs = list of uint8 arrays (one per contig); reads = Reads.
    `seed` fixes the reference and its (germline) indels; `read_seed` (default: derived from seed)
    the sampled pairs; somatic_spacing > 0 adds extra indels about every that many bases, drawn from
    the read_seed stream -- a tumour/normal pair shares `seed` and differs in the other two
    (BASELINE config 5)."""
    rng = np.random.default_rng(seed)
    rrng = np.random.default_rng(seed + 7919 if read_seed is None else read_seed)
    L = read_len
    refs = []
    cols = {k: [] for k in ("tid", "pos", "flag", "mpos", "isize", "seq", "cig_op", "cig_len", "ncig", "mate_first", "pair_id")}
    pair_base = 0
    if ref_lens is not None:            # one length per contig (human-like spreads, BASELINE config 4)
        n_contigs = len(ref_lens)
    for tid in range(n_contigs):
        if ref_lens is not None:
            ref_len = int(ref_lens[tid])
        ref = random_genome(rng, ref_len)
        refs.append(ref)
        epos, esize, eins = plant_indels(rng, ref_len, indel_spacing, big_every=big_every)
        if somatic_spacing:
            spos, ssize, sins = plant_indels(rrng, ref_len, somatic_spacing, margin=1500 + indel_spacing // 2)
            far = np.array([np.min(np.abs(epos - p)) > 300 for p in spos], dtype=bool)
            epos = np.concatenate([epos, spos[far]]); esize = np.concatenate([esize, ssize[far]]); eins = np.concatenate([eins, sins[far]])
            o = np.argsort(epos, kind="stable")
            epos, esize, eins = epos[o], esize[o], eins[o]
        # inserted bases come from a stream keyed by the event position: same germline insertion in tumour and normal
        donor, edpos = build_donor(seed, ref, epos, esize, eins)
        dlen = len(donor)
        # donor -> reference shift after each event
        eshift = np.where(eins, -esize, esize).astype(np.int64)
        cshift = np.concatenate([[0], np.cumsum(eshift)])
        # donor interval each event occupies: deletion = point, insertion = [d, d+size)
        ed_lo = edpos
        ed_hi = np.where(eins, edpos + esize, edpos)

        n_pairs = int(round(coverage * ref_len / (2.0 * L)))
        isz = np.clip(np.rint(rrng.normal(isize_mean, isize_sd, n_pairs)), isize_min, isize_max).astype(np.int64)
        fs = rrng.integers(0, dlen - isize_max - 1, size=n_pairs)
        starts = np.concatenate([fs, fs + isz - L])                 # donor start of mate1 (fwd), mate2 (rev)
        is_rev = np.concatenate([np.zeros(n_pairs, bool), np.ones(n_pairs, bool)])
        n = 2 * n_pairs

        # bases in forward-reference orientation, with substitutions
        idx = starts[:, None] + np.arange(L)[None, :]
        seq = donor[idx]
        sub = rrng.random(seq.shape) < sub_rate
        seq[sub] = ACGT[rrng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]

        # event (if any) overlapping each read: first event with ed_hi > start (strictly inside the read)
        ev = np.searchsorted(ed_hi, starts, side="right")
        ev_ok = ev < len(ed_lo)
        evc = np.minimum(ev, len(ed_lo) - 1)
        hit = ev_ok & (ed_lo[evc] < starts + L) & (ed_hi[evc] > starts)
        # an insertion that starts exactly at the read start, or a deletion point at it, is no event for the read
        hit &= ~((ed_lo[evc] <= starts) & (ed_hi[evc] <= starts))
        # reference coordinate of a donor position that lies outside inserted bases
        nbefore = np.searchsorted(ed_hi, starts, side="right")      # events entirely before the read start
        # reads starting inside an insertion: treat the insertion as overlapping (handled below)
        pos = starts + cshift[nbefore]
        cig_op = np.zeros((n, 3), dtype=np.int8)
        cig_len = np.zeros((n, 3), dtype=np.int32)
        ncig = np.ones(n, dtype=np.int8)
        cig_op[:, 0] = OP_M; cig_len[:, 0] = L
        unmapped = np.zeros(n, bool)

        for r in np.nonzero(hit)[0]:
            e = int(evc[r]); s = int(starts[r])
            if eins[e]:
                a = max(0, int(ed_lo[e]) - s)                        # read bases before the insertion
                ins = min(s + L, int(ed_hi[e])) - max(s, int(ed_lo[e]))
                b = L - a - ins
                ref_left = s + int(cshift[e]) if a > 0 else None     # ref coord of read base 0 when it is not inserted
                ref_ins = int(epos[e])                               # insertion point on the reference
                if a >= 20 and b >= 20:
                    ops = [(a, OP_M), (ins, OP_I), (b, OP_M)]; p = ref_left
                elif max(a, b) < 30:
                    unmapped[r] = True; ops = []; p = -1
                elif a >= b:
                    ops = [(a, OP_M), (L - a, OP_S)]; p = ref_left
                else:
                    ops = [(L - b, OP_S), (b, OP_M)]; p = ref_ins
            else:
                a = int(ed_lo[e]) - s
                b = L - a
                ref_left = s + int(cshift[e])
                if a <= 0 or b <= 0:
                    continue
                if a >= 20 and b >= 20 and esize[e] <= 50:
                    ops = [(a, OP_M), (int(esize[e]), OP_D), (b, OP_M)]; p = ref_left
                elif a >= b:
                    ops = [(a, OP_M), (b, OP_S)]; p = ref_left
                else:
                    ops = [(a, OP_S), (b, OP_M)]; p = ref_left + a + int(esize[e])
            pos[r] = p
            ncig[r] = len(ops)
            for j, (ln, op) in enumerate(ops):
                cig_op[r, j] = op; cig_len[r, j] = ln

        mate = np.concatenate([np.arange(n_pairs, 2 * n_pairs), np.arange(0, n_pairs)])
        mpos = pos[mate].copy()
        m_unm = unmapped[mate]
        # SAM convention: an unmapped read sits at its mate's position
        pos = np.where(unmapped, mpos, pos)
        mpos = np.where(m_unm, pos, mpos)
        both_unm = unmapped & m_unm
        flag = np.full(n, 0x1, dtype=np.int32)
        flag |= np.where(np.arange(n) < n_pairs, 0x40, 0x80)
        flag |= np.where(is_rev, 0x10, 0) | np.where(is_rev[mate], 0x20, 0)
        flag |= np.where(unmapped, 0x4, 0) | np.where(m_unm, 0x8, 0)
        # reference span of each read for TLEN
        span = np.where(unmapped, 0, (cig_len * np.isin(cig_op, (OP_M, OP_D)) * (np.arange(3)[None, :] < ncig[:, None])).sum(1))
        end = pos + span
        lo = np.minimum(pos, pos[mate]); hi = np.maximum(end, end[mate])
        tl = (hi - lo).astype(np.int64)
        isize = np.where(unmapped | m_unm, 0, np.where(pos <= pos[mate], tl, -tl))
        # proper pair: both mapped and the observed template within the library's range
        flag |= np.where(~unmapped & ~m_unm & (tl <= isize_max), 0x2, 0)
        keep = ~both_unm
        order = np.argsort(pos[keep], kind="stable")
        sel = np.nonzero(keep)[0][order]
        cols["tid"].append(np.full(len(sel), tid, np.int32))
        cols["pos"].append(pos[sel].astype(np.int32)); cols["flag"].append(flag[sel])
        cols["mpos"].append(mpos[sel].astype(np.int32)); cols["isize"].append(isize[sel].astype(np.int32))
        cols["seq"].append(seq[sel]); cols["cig_op"].append(cig_op[sel]); cols["cig_len"].append(cig_len[sel])
        cols["ncig"].append(ncig[sel]); cols["mate_first"].append((np.arange(n) < n_pairs)[sel])
        cols["pair_id"].append((pair_base + (np.arange(n) % n_pairs))[sel])
        pair_base += n_pairs
    rd = Reads()
    for k, v in cols.items():
        setattr(rd, k, np.concatenate(v))
    rd.n = len(rd.pos)
    rd.read_len = L
    rd.range_max = isize_max
    rd.mapq = 60
    return refs, rd


def candidates(rd, qthreshold=10):
    """The reads fetch_func hands to attempt_pe_alignment (src/indelminer.c:384-515), in
    arrival order.  Returns a dict of arrays: index (into rd), tid, anchor, range_max, bases [n,L]."""
    f = rd.flag
    filt = (f & (0x100 | 0x200 | 0x400 | 0x800)) == 0
    paired = (f & 0x1) != 0
    aligned = (f & 0x4) == 0
    mate_al = (f & 0x8) == 0
    is_rc = (f & 0x10) != 0
    mate_rc = (f & 0x20) != 0
    mq_ok = rd.mapq >= qthreshold
    valid_op = np.arange(3)[None, :] < rd.ncig[:, None]
    ndel = ((rd.cig_op == OP_D) & valid_op).sum(1)
    nins = ((rd.cig_op == OP_I) & valid_op).sum(1)
    nclip = ((rd.cig_op == OP_S) & valid_op).sum(1)
    first_op = rd.cig_op[:, 0]
    last_op = rd.cig_op[np.arange(rd.n), np.maximum(rd.ncig - 1, 0)]
    three_prime = np.where(is_rc, first_op == OP_S, last_op == OP_S)
    boring = ((nclip == 0) | ((nclip == 1) & three_prime)) & (ndel == 0) & (nins == 0)
    proper = filt & paired & aligned & mate_al & ((f & 0x2) != 0) & ~boring & mq_ok
    unm = filt & paired & ~aligned & mate_al & mq_ok
    take = proper | unm
    idx = np.nonzero(take)[0]
    bases = rd.seq[idx].copy()
    # reverse-complement when read and mate share an orientation flag (proper pairs), or when the
    # mate of an unmapped read is forward (src/indelminer.c:404-409,479-484).  The simulator keeps
    # bases in forward-reference orientation, i.e. already what BAM stores for mapped reads; an
    # unmapped read is stored as sequenced, which for a reverse-strand read is the reverse
    # complement, so the two flips cancel and `bases` is what the reference would pass.
    return dict(index=idx.astype(np.int64), tid=rd.tid[idx], anchor=rd.mpos[idx],
                range_max=np.full(len(idx), rd.range_max, np.int32), bases=bases,
                n_reads=rd.n)
