"""Parity of the device-resident pipeline stages with the CPU oracle (bit-exact):
record triage (fetch_func's rules, a1), realign with kept CIGAR-derived evidence, the READCHUNK
flush cuts, the split-read group-by, and the genome-wide depth array."""
import os
import struct

import numpy as np
import pytest

from indelminer_amd import capi, rawrec, synth
from tests.support import oraclebind as ob

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compare_triage(pipe, raw, rec_off, rg_names, rg_range, tri=None, **kw):
    """runs the HIP triage over (raw, rec_off) and checks every output against the oracle"""
    n = len(rec_off) - 1
    tri = tri or ob.triage_records(raw, rec_off, rg_names, rg_range, **kw)
    pipe.upload(raw, rec_off)
    pipe.triage()
    c = pipe.fetch_counts()
    # the kernel holds IM_MAX_EV CIGAR-derived evidence per read: more is IM_REC_ERR_LIMIT there (the oracle has no limit)
    o_cls = np.array([21 if (t.cls == 3 and t.n_ev > capi.MAX_EV) else t.cls for t, _ in tri], dtype=np.uint8)
    h_cls = pipe.d_class.download(np.uint8, n)
    assert np.array_equal(h_cls, o_cls), np.nonzero(h_cls != o_cls)[0][:10]
    # a soft clip inside the CIGAR (19), a bad base code outside the CIGAR's reach or in an unaligned read (20) and the evidence
    # limit are found while the candidate is being written: it keeps its place in the batch.  What new_readaln refuses in a
    # proper pair (18, 20) is found before the record becomes a candidate.
    cand = [i for i in range(n) if tri[i][0].want]
    assert c[0] == len(cand)
    assert c[2] == int((o_cls != 0).sum()) and c[3] == int((o_cls >= 16).sum()) and c[4] == 0
    m = len(cand)
    assert np.array_equal(pipe.d_cand_rec.download(np.int32, max(m, 1))[:m], np.array(cand, np.int32))
    boff = pipe.d_boff.download(np.int64, max(m, 1))[:m]
    blen = pipe.d_len.download(np.int32, max(m, 1))[:m]
    bases = pipe.d_bases.download(np.uint8, pipe.cap_bases)
    tid = pipe.d_tid.download(np.int32, max(m, 1))[:m]
    anchor = pipe.d_anchor.download(np.int32, max(m, 1))[:m]
    rng = pipe.d_range.download(np.int32, max(m, 1))[:m]
    s_cls = pipe.d_cls.download(np.int32, pipe.n_slots).reshape(-1)
    s_b1 = pipe.d_b1.download(np.int32, pipe.n_slots)
    s_b2 = pipe.d_b2.download(np.int32, pipe.n_slots)
    pos = 0
    for j, i in enumerate(cand):
        t, b = tri[i]
        assert boff[j] == pos and boff[j] % 4 == 0
        if t.cls in (18, 19, 20):
            pos += (int(blen[j]) + 3) // 4 * 4
            continue
        assert blen[j] == t.l_seq
        assert bases[pos:pos + t.l_seq].tobytes() == b, (i, bases[pos:pos + t.l_seq].tobytes(), b)
        assert not bases[pos + t.l_seq:pos + (t.l_seq + 3) // 4 * 4].any()
        pos += (t.l_seq + 3) // 4 * 4
        assert (tid[j], anchor[j], rng[j]) == (t.tid, t.anchor, t.range_max)
        for k in range(capi.MAX_EV):
            sl = j * capi.MAX_EV + k
            if k < t.n_ev and t.n_ev <= capi.MAX_EV:
                assert (s_cls[sl], s_b1[sl], s_b2[sl]) == (t.ev_cls[k], t.ev_b1[k], t.ev_b2[k])
            else:
                assert s_cls[sl] == -1
    # the same records the way the product's walkers deliver them -- without their base qualities (bin = 0xFFFF; records whose CIGAR
    # reaches past l_seq keep them): every output must be what it was
    raw2, off2 = rawrec.strip_quals(raw, rec_off)
    assert len(raw2) < len(raw) or n == 0 or not any(t.cls in (1, 3) for t, _ in tri)
    pipe.upload(raw2, off2)
    pipe.triage()
    c2 = pipe.fetch_counts()
    h_cls2 = pipe.d_class.download(np.uint8, n)
    assert np.array_equal(h_cls2, h_cls), [(int(i), int(h_cls2[i]), int(h_cls[i])) for i in np.nonzero(h_cls2 != h_cls)[0][:8]]
    assert np.array_equal(c2[:5], c[:5]), (c2[:5], c[:5])
    assert np.array_equal(pipe.d_cand_rec.download(np.int32, max(m, 1))[:m], np.array(cand, np.int32))
    assert np.array_equal(pipe.d_boff.download(np.int64, max(m, 1))[:m], boff) and np.array_equal(pipe.d_len.download(np.int32, max(m, 1))[:m], blen)
    assert np.array_equal(pipe.d_bases.download(np.uint8, pipe.cap_bases)[:pos], bases[:pos])
    assert np.array_equal(pipe.d_anchor.download(np.int32, max(m, 1))[:m], anchor) and np.array_equal(pipe.d_range.download(np.int32, max(m, 1))[:m], rng)
    assert np.array_equal(pipe.d_cls.download(np.int32, pipe.n_slots).reshape(-1)[:m * capi.MAX_EV], s_cls[:m * capi.MAX_EV])
    assert np.array_equal(pipe.d_b1.download(np.int32, pipe.n_slots)[:m * capi.MAX_EV], s_b1[:m * capi.MAX_EV])
    assert np.array_equal(pipe.d_b2.download(np.int32, pipe.n_slots)[:m * capi.MAX_EV], s_b2[:m * capi.MAX_EV])
    return tri, cand


def _synth(seed=7, ref_len=60_000, coverage=20, **kw):
    refs, rd = synth.simulate(seed=seed, ref_len=ref_len, coverage=coverage, **kw)
    raw, off = rawrec.records(rd)
    return refs, rd, raw, off


def test_triage_matches_oracle_synthetic(gpu_ctx):
    refs, rd, raw, off = _synth(big_every=7)
    gpu_ctx.set_reference([r.tobytes() for r in refs])
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n)
    tri, cand = _compare_triage(pipe, raw, off, ["generic"], [rd.range_max])
    # the simulator's own candidate rule agrees (third opinion)
    sc = synth.candidates(rd)
    assert np.array_equal(sc["index"], np.array(cand))
    assert any(t.cls == capi.REC_PE for t, _ in tri)
    assert any(t.n_ev > 0 for t, _ in tri)


def test_triage_matches_oracle_test_data(gpu_ctx, golden_dir):
    raw, off, contigs = rawrec.records_from_bam(os.path.join(golden_dir, "test_data", "alignments.bam"))
    fa = open(os.path.join(golden_dir, "test_data", "reference.fa")).read().split("\n", 1)[1].replace("\n", "").upper()
    gpu_ctx.set_reference([fa.encode()])
    gpu_ctx.set_insert_ranges(["generic"], [705])
    n = len(off) - 1
    for q in (10, 0):
        pipe = capi.Pipeline(gpu_ctx, n, len(raw), cap_cand=n, qthreshold=q)
        tri, cand = _compare_triage(pipe, raw, off, ["generic"], [705], qthreshold=q)
        # SURVEY.md appendix B census: 697 candidates at -q 10; the 6 unmapped-read candidates only pass at -q 0
        assert len(cand) == (697 if q == 10 else 703)


def _rec(flag, tid=0, pos=100, mtid=0, mpos=300, isize=300, mapq=60, cigar=((100, 0),), seq=None, tags=b"", qname=b"q\0", l_seq=100, pad=b"\0\0\0", qual=None):
    seq = seq if seq is not None else bytes([0x12] * ((l_seq + 1) // 2))
    qual = qual if qual is not None else b"\x28" * l_seq
    cig = b"".join(struct.pack("<I", (l << 4) | o) for l, o in cigar)
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qname), mapq, 0, len(cigar), flag, l_seq, mtid, mpos, isize) + qname + cig + seq + qual + tags
    return body + pad[:(-len(body)) % 4]


def test_triage_edge_records(gpu_ctx):
    """read groups with the hashtable's prefix / last-hit semantics, MQ tag types, bad CIGARs, every skip rule"""
    gpu_ctx.set_reference([b"ACGT" * 500])
    names = ["lib1", "lib10", "generic", "li", "x" * 40, "lib1b"]
    ranges = [500, 600, 700, 800, 900, 1000]
    gpu_ctx.set_insert_ranges(names, ranges)
    P, S = 0x1 | 0x2, ((30, 4), (70, 0))
    recs = [
        _rec(P | 0x100), _rec(P | 0x200), _rec(P | 0x400), _rec(P | 0x800), _rec(0x2), _rec(P, mtid=1),         # skips
        _rec(P, cigar=S, tags=b"RGZlib1\0"), _rec(P, cigar=S, tags=b"RGZlib10\0"), _rec(P, cigar=S, tags=b"RGZli\0"),
        _rec(P, cigar=S, tags=b"RGZl\0"), _rec(P, cigar=S, tags=b"RGZnope\0"), _rec(P, cigar=S, tags=b"RGZ" + b"x" * 40 + b"\0"),
        _rec(P, cigar=S, tags=b"RGAx"), _rec(P, cigar=S, tags=b"XYi\1\0\0\0RGZlib1b\0MQC\x05"),                    # MQ 5 < q
        _rec(P, cigar=S, tags=b"MQC\x0a"), _rec(P, cigar=S, tags=b"MQf\0\0\x80\x3f"), _rec(P, cigar=S, tags=b"MQs\xff\xff"),
        _rec(0x1 | 0x4 | 0x40, cigar=(), tags=b"MQZ12\0"), _rec(0x1 | 0x4 | 0x40, cigar=(), tags=b"MQI\x3c\0\0\0"),
        _rec(0x1 | 0x4 | 0x20 | 0x80, cigar=(), mapq=5), _rec(0x1 | 0x8),                                          # mate forward / reverse; mapped with unmapped mate
        _rec(0x1 | 0x4 | 0x8, cigar=()),                                                                           # both unmapped
        _rec(P, cigar=((50, 0), (10, 3), (50, 0))), _rec(P, cigar=((10, 5), (90, 0))), _rec(P, cigar=((100, 9),)),
        _rec(P, cigar=((40, 0), (5, 4), (55, 0))), _rec(P | 0x10, cigar=((30, 4), (70, 0))), _rec(P | 0x10, cigar=((70, 0), (30, 4))),
        _rec(P, cigar=((70, 0), (30, 4))), _rec(P, cigar=((20, 0), (3, 1), (30, 0), (7, 2), (47, 0))),
        _rec(P, cigar=((5, 0), (3, 1), (92, 0))), _rec(P, cigar=((10, 0),) + ((1, 1), (10, 0)) * 6 + ((24, 0),)),  # > 4 CIGAR evidence
        _rec(0x1 | 0x20, isize=5000), _rec(0x1 | 0x10, isize=-5000), _rec(0x1 | 0x20, isize=650), _rec(0x1 | 0x20 | 0x10, isize=5000),
        _rec(0x1 | 0x20, isize=2000000), _rec(P, cigar=S, seq=bytes([0x13] * 50)), _rec(P, cigar=S, l_seq=33, seq=bytes([0x48] * 17)),
        _rec(P, cigar=((20, 4), (13, 0)), l_seq=33, seq=bytes([0x48] * 17)), _rec(P | 0x10 | 0x20, cigar=((20, 4), (13, 0)), l_seq=33, seq=bytes([0x84, 0x21] * 8 + [0xf0])),
        _rec(P, cigar=S)[:60],                                                                                     # truncated record
        _rec(P, cigar=S, qname=b"qq\0", tags=b"MQC\x0a", pad=b"RGZ"), _rec(P, cigar=S, qname=b"qqq\0", pad=b"MQ"),   # alignment padding is no aux field
        # new_readaln decodes EVERY proper pair op by op: a bad base code in a plain 100M read, behind / in front of a bad op,
        # in a clipped part, past the CIGAR's coverage (not read), and a CIGAR that runs past l_seq into the qualities
        _rec(P, seq=bytes([0x12] * 20 + [0x13] + [0x12] * 29)), _rec(P, cigar=((50, 0), (50, 3)), seq=bytes([0x12] * 10 + [0x31] + [0x12] * 39)),
        _rec(P, cigar=((10, 0), (5, 3), (85, 0)), seq=bytes([0x12] * 30 + [0x50] + [0x12] * 19)), _rec(P, cigar=((20, 4), (80, 0)), seq=bytes([0x02] + [0x12] * 49)),
        _rec(P, cigar=((60, 0),), seq=bytes([0x12] * 40 + [0x77] * 10)), _rec(P, cigar=((120, 0),)), _rec(P, cigar=((50, 0), (50, 9)), seq=bytes([0x12] * 49 + [0x10])),
        _rec(P, cigar=((120, 0),), qual=b"\x28" * 5 + b"\x30" + b"\x28" * 94), _rec(P, cigar=((106, 0),), qual=b"\x28" * 5 + b"\x30" + b"\x28" * 94),   # the bytes behind the bases, read as bases
        _rec(P, cigar=((300, 0),), l_seq=300, seq=bytes([0x12] * 145 + [0x13] + [0x12] * 4)), _rec(P, cigar=((200, 0),), l_seq=200, seq=bytes([0x12] * 90 + [0x31] + [0x12] * 9)),
        _rec(P, cigar=((290, 0), (10, 4)), l_seq=300, seq=bytes([0x12] * 150)), _rec(P, cigar=((129, 0),), l_seq=129, seq=bytes([0x12] * 64 + [0x70])),
    ]
    raw = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    off = np.zeros(len(recs) + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in recs], out=off[1:])
    pipe = capi.Pipeline(gpu_ctx, len(recs), len(raw), cap_cand=len(recs), maxpedelsize=1000000)
    tri, cand = _compare_triage(pipe, raw, off, names, ranges)
    seen = {t.cls for t, _ in tri}
    assert {0, 1, 2, 3, 4, 16, 17, 18, 19, 20, 21} <= seen, seen


def test_realign_keep_flush_groupby(gpu_ctx):
    """triage -> realign (CIGAR-derived evidence survives where realignment finds none) -> three flushes with
    paired-read entries in the cut -> group-by, against the oracle's realign + flush + a plain group-by"""
    refs, rd, raw, off = _synth(seed=11, ref_len=120_000, coverage=25)
    contig = refs[0].tobytes()
    gpu_ctx.set_reference([contig])
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    rng = np.random.default_rng(3)
    n_pe = 40
    pe_b1 = rng.integers(0, 120_000, n_pe).astype(np.int32)
    pe_b2 = (pe_b1 + rng.integers(200, 900, n_pe)).astype(np.int32)
    pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n // 4, n_pe=n_pe)
    tri, cand = _compare_triage(pipe, raw, off, ["generic"], [rd.range_max])
    pipe.set_pe(pe_b1, pe_b2)
    pipe.realign()
    m = len(cand)
    # oracle: realigned evidence replaces the CIGAR-derived evidence, else the latter stays (src/indelminer.c:494-512)
    P = ob.params()
    E = capi.MAX_EV
    o_cls = np.full(m * E + n_pe, -1, np.int32); o_b1 = np.zeros(m * E + n_pe, np.int32); o_b2 = np.zeros(m * E + n_pe, np.int32)
    kept = 0
    for j, i in enumerate(cand):
        t, b = tri[i]
        st, res = ob.realign(P, contig, len(contig), t.anchor, t.range_max, b)
        if st == 1 and res.n_ev > 0:
            ev = [(res.ev[k].cls, res.ev[k].b1, res.ev[k].b2) for k in range(res.n_ev)]
        else:
            ev = [(t.ev_cls[k], t.ev_b1[k], t.ev_b2[k]) for k in range(t.n_ev)]
            kept += bool(ev)
        for k, (c, x1, x2) in enumerate(ev):
            o_cls[j * E + k], o_b1[j * E + k], o_b2[j * E + k] = c, x1, x2
    assert kept > 0
    base = pipe.cap_cand * E
    nsl = m * E
    h_cls = pipe.d_cls.download(np.int32, pipe.n_slots); h_b1 = pipe.d_b1.download(np.int32, pipe.n_slots); h_b2 = pipe.d_b2.download(np.int32, pipe.n_slots)
    assert np.array_equal(h_cls[:nsl], o_cls[:nsl]) and np.array_equal(h_b1[:nsl], o_b1[:nsl]) and np.array_equal(h_b2[:nsl], o_b2[:nsl])
    o_cls[nsl:] = 2; o_b1[nsl:] = pe_b1; o_b2[nsl:] = pe_b2
    # flushes: (candidate prefix, PE prefix, marker); the last one takes everything
    flushes = [(m // 3, 10, 30_000), (2 * m // 3, 25, 70_000), (m, n_pe, 60_000), (m, n_pe, 2**31 - 1)]
    o_cons = np.zeros(nsl + n_pe, np.int32)
    for k, (ch, ph, marker) in enumerate(flushes):
        pipe.flush(k, ch, ph, marker)
        vis = np.full(nsl + n_pe, -1, np.int32)            # what this flush can see
        vis[:ch * E] = o_cls[:ch * E]; vis[nsl:nsl + ph] = 2
        ob.flush_cut(vis, o_b1, o_b2, o_cons, marker, k + 1)
    pipe.groupby()
    pipe.sync()
    h_cons = pipe.d_consumed.download(np.int32, pipe.n_slots)
    assert np.array_equal(h_cons[:nsl], o_cons[:nsl]) and np.array_equal(h_cons[base:base + n_pe], o_cons[nsl:])
    assert len(set(o_cons[:nsl][o_cls[:nsl] >= 0])) >= 3 and (o_cons[:nsl][o_cls[:nsl] >= 0] > 0).all()
    key, first, count, order = pipe.clusters()
    want = {}
    for s in range(nsl):
        if o_cls[s] >= 0 and o_cons[s] > 0:
            want.setdefault((int(o_cons[s]), int(o_cls[s]), int(o_b1[s]), int(o_b2[s])), []).append(s)
    got = {tuple(int(x) for x in key[c]): list(order[first[c]:first[c] + count[c]]) for c in range(len(key))}
    assert got == want
    # tie_desc reverses the members
    pipe.groupby(tie_desc=1)
    pipe.sync()
    key, first, count, order = pipe.clusters()
    got = {tuple(int(x) for x in key[c]): list(order[first[c]:first[c] + count[c]]) for c in range(len(key))}
    assert got == {k: v[::-1] for k, v in want.items()}


@pytest.mark.parametrize("mode", ["wide", "seq", "per-flush"])
def test_async_pass_matches_stepwise(gpu_ctx, mode):
    """the whole pass bound on one stream without host round trips (device-resident candidate count, record bounds,
    the flush list chip-wide without history / walked by one workgroup / one launch pair per flush) gives the stepwise results"""
    one_launch = mode != "per-flush"
    refs, rd, raw, off = _synth(seed=13, ref_len=150_000, coverage=25, big_every=5)
    contig = refs[0].tobytes()
    gpu_ctx.set_reference([contig])
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    import bench
    flushes, pe_b1, pe_b2 = bench.flush_schedule(rd)
    # finer flush points than READCHUNK, with the pair table's markers replaced by positions along the contig
    cuts = [rd.n // 4, rd.n // 2, 3 * rd.n // 4]
    flushes = [(0, c, int(np.searchsorted(np.arange(len(pe_b1)), len(pe_b1) * c // rd.n)), int(rd.pos[c - 1])) for c in cuts] + [flushes[-1]]
    ref_pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n // 4, n_pe=max(len(pe_b1), 1), n_flushes=len(flushes))
    ref_pipe.upload(raw, off); ref_pipe.set_pe(pe_b1, pe_b2)
    ref_pipe.triage(); ref_pipe.fetch_counts(); ref_pipe.realign()
    cand_rec = ref_pipe.d_cand_rec.download(np.int32, ref_pipe.n_cand)
    for k, (r0, r1, pe_hi, marker) in enumerate(flushes):
        ref_pipe.flush(k, int(np.searchsorted(cand_rec, r1)), pe_hi, marker, cand_lo=int(np.searchsorted(cand_rec, r0)))
    ref_pipe.groupby(); ref_pipe.sync()
    want_cons = ref_pipe.d_consumed.download(np.int32, ref_pipe.n_slots)
    key, first, count, order = ref_pipe.clusters()
    want = {tuple(int(x) for x in key[c]): list(order[first[c]:first[c] + count[c]]) for c in range(len(key))}
    assert len(want) > 10 and len({k[0] for k in want}) == len(flushes)

    pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n // 4, n_pe=max(len(pe_b1), 1), n_flushes=len(flushes), input_from=ref_pipe)
    pipe.set_pe(pe_b1, pe_b2)
    st = capi.new_stream(gpu_ctx)
    for fn, args in pipe.bind_async(flushes, st, grid_bound=rd.n // 8, one_launch_flushes=one_launch, wide=mode == "wide"):
        gpu_ctx._check(fn(*args))
    pipe.sync(st)
    pipe.fetch_counts()
    assert pipe.n_cand == ref_pipe.n_cand
    got_cons = pipe.d_consumed.download(np.int32, pipe.n_slots)
    live, base = pipe.n_cand * capi.MAX_EV, pipe.cap_cand * capi.MAX_EV        # the marks of unused slots are nobody's business
    assert np.array_equal(got_cons[:live], want_cons[:live]) and np.array_equal(got_cons[base:], want_cons[base:])
    key, first, count, order = pipe.clusters()
    got = {tuple(int(x) for x in key[c]): list(order[first[c]:first[c] + count[c]]) for c in range(len(key))}
    assert got == want
    a = pipe.d_res.download(capi.RESULT_DTYPE, pipe.n_cand); b = ref_pipe.d_res.download(capi.RESULT_DTYPE, pipe.n_cand)
    assert a.tobytes() == b.tobytes()


def test_groupby_large_support(gpu_ctx):
    """a breakpoint with more than 1024 supporting reads takes the global-memory ordering path"""
    n = 5000
    pipe = capi.Pipeline(gpu_ctx, 1, 64, cap_cand=n)
    E = capi.MAX_EV
    cls = np.full(n * E, -1, np.int32); b1 = np.zeros(n * E, np.int32); b2 = np.zeros(n * E, np.int32)
    cls[::E] = 1; b1[::E] = 1000; b2[::E] = 1010
    cls[1::E][:700] = 0; b1[1::E][:700] = 555; b2[1::E][:700] = 555
    pipe.d_cls.upload(cls); pipe.d_b1.upload(b1); pipe.d_b2.upload(b2)
    pipe.d_consumed.upload(np.where(cls >= 0, 1, 0).astype(np.int32))
    pipe.n_cand = n
    pipe.groupby()
    pipe.sync()
    key, first, count, order = pipe.clusters()
    got = {tuple(int(x) for x in key[c]): order[first[c]:first[c] + count[c]] for c in range(len(key))}
    assert set(got) == {(1, 1, 1000, 1010), (1, 0, 555, 555)}
    assert np.array_equal(got[(1, 1, 1000, 1010)], np.arange(0, n * E, E))
    assert np.array_equal(got[(1, 0, 555, 555)], np.arange(1, 700 * E, E))


def test_depth_genome_wide(gpu_ctx):
    refs, rd, raw, off = _synth(seed=5, ref_len=40_000, coverage=15, n_contigs=2)
    gpu_ctx.set_reference([r.tobytes() for r in refs])
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    gpu_ctx.depth_enable()
    pipe = capi.Pipeline(gpu_ctx, rd.n, len(raw), cap_cand=rd.n, want_depth=True)
    pipe.upload(raw, off)
    pipe.triage()
    pipe.sync()
    rng = np.random.default_rng(1)
    for tid in (0, 1):
        gpu_ctx.depth_scan(tid)
        want = ob.depth_of(raw, off, tid, 40_000)
        beg = rng.integers(0, 39_000, 200).astype(np.int32)
        end = (beg + rng.integers(1, 900, 200)).astype(np.int32)
        beg[0], end[0] = 0, 40_000
        got = gpu_ctx.depth_query_tid(tid, beg, end)
        cs = np.concatenate([[0], np.cumsum(want.astype(np.int64))])
        assert np.array_equal(got.astype(np.int64), cs[np.minimum(end, 40_000)] - cs[beg])


def test_groupby_scratch_serves_groups_of_any_size(gpu_ctx):
    """One pipeline's buffers serve group after group of very different sizes (the product: a few contigs per group, as they
    come).  The group-by scratch is laid out once, for its capacity: a layout per call put one call's counters inside the
    previous call's slot lists -- phantom clusters naming empty slots, found on the 24-contig input.  Each pass is checked
    against a plain group-by of the arrays the device itself holds."""
    refs, rd, raw, off = _synth(seed=17, ref_len=400_000, coverage=25, big_every=6)
    gpu_ctx.set_reference([refs[0].tobytes()])
    gpu_ctx.set_insert_ranges(["generic"], [rd.range_max])
    n = rd.n
    pipe = capi.Pipeline(gpu_ctx, n, len(raw), cap_cand=n // 4)
    sizes = [n, n // 40, n // 3, n // 200, n // 2, n // 11, n, 50, n // 5]
    seen = 0
    for m in sizes:
        sub_off = off[:m + 1]
        pipe.upload(raw[:int(sub_off[-1])], sub_off)
        pipe.recs.n = m
        pipe.triage()
        pipe.fetch_counts()
        pipe.realign()
        nc = pipe.n_cand
        pipe.flush(0, nc, 0, 2**31 - 1)
        pipe.groupby()
        pipe.sync()
        ns = nc * capi.MAX_EV
        cls = pipe.d_cls.download(np.int32, pipe.n_slots)[:ns]
        b1 = pipe.d_b1.download(np.int32, pipe.n_slots)[:ns]
        b2 = pipe.d_b2.download(np.int32, pipe.n_slots)[:ns]
        cons = pipe.d_consumed.download(np.int32, pipe.n_slots)[:ns]
        want = {}
        for s in np.nonzero((cls >= 0) & (cls < 2) & (cons > 0))[0]:
            want.setdefault((int(cons[s]), int(cls[s]), int(b1[s]), int(b2[s])), []).append(int(s))
        key, first, count, order = pipe.clusters()
        got = {tuple(int(x) for x in key[c]): [int(x) for x in order[first[c]:first[c] + count[c]]] for c in range(len(key))}
        assert got == want, (m, len(got), len(want))
        seen += len(want)
    assert seen > 500


def _wide_case(rng, n_cand, n_ctg, n_pe, pinned, dense):
    """slot arrays, candidate records, paired-read entries and a flush list of n_ctg contigs the way the product builds
    them: records ascend, a contig's flushes share rec0, markers never decrease inside a contig, INT_MAX at its end"""
    E = capi.MAX_EV
    rec = np.cumsum(rng.integers(1, 40, n_cand)).astype(np.int32)
    n_rec = int(rec[-1]) + 1
    ctg_of_rec = np.sort(rng.integers(0, n_ctg, n_rec))                  # records -> contigs, in order
    ctg_start = [int(np.searchsorted(ctg_of_rec, c)) for c in range(n_ctg)] + [n_rec]
    pos_of_rec = np.zeros(n_rec, np.int64)
    for c in range(n_ctg):
        a, b = ctg_start[c], ctg_start[c + 1]
        pos_of_rec[a:b] = np.sort(rng.integers(0, 2_000_000, b - a))
    cls = np.full(n_cand * E, -1, np.int32); b1 = np.zeros(n_cand * E, np.int32); b2 = np.zeros(n_cand * E, np.int32)
    pool = rng.integers(-600, 600, 64)                                   # few distinct offsets: clusters with several members
    for j in range(E):
        live = rng.random(n_cand) < (0.9 if j == 0 else 0.15)
        s = pos_of_rec[rec] // 200 * 200 + pool[rng.integers(0, 64, n_cand)]
        s = np.maximum(s, 0)
        cls[j::E] = np.where(live, rng.integers(0, 2, n_cand), -1)
        b1[j::E] = s
        b2[j::E] = s + np.where(rng.random(n_cand) < 0.6, rng.integers(1, 900, n_cand) // 50 * 50, 0)
    pe_rec = np.sort(rng.integers(0, n_rec, n_pe))
    pe_b1 = np.maximum(pos_of_rec[pe_rec] + rng.integers(-500, 500, n_pe), 0).astype(np.int32)
    pe_b2 = (pe_b1 + rng.integers(100, 5000, n_pe)).astype(np.int32)
    flushes = []
    for c in range(n_ctg):
        a, b = ctg_start[c], ctg_start[c + 1]
        nf = int(rng.integers(0, 40 if dense else 6))
        cuts = np.sort(rng.integers(a, b + 1, nf)) if b > a else []
        floor = int(rng.integers(0, 100_000)) if pinned else 2**31 - 1
        mk = -1
        for r1 in cuts:
            here = int(pos_of_rec[r1 - 1]) if r1 > a else 0
            mk = max(mk, min(floor, here - int(rng.integers(0, 3000))))
            flushes.append((a, int(r1), int(np.searchsorted(pe_rec, r1)), mk))
        flushes.append((a, b, int(np.searchsorted(pe_rec, b)), 2**31 - 1))
    return rec, cls, b1, b2, pe_b1, pe_b2, flushes


def test_flush_groupby_wide_against_the_stepwise_oracle(gpu_ctx):
    """im_dev_flush_groupby (three chip-wide launches, no history) on ONE pipeline's buffers, group after group: several
    contigs, dense and sparse flush lists, markers pinned low (every entry a cutting candidate up to its contig's end), paired-read
    entries in the cuts, both member orders -- against imo_flush_cut flush by flush and a plain group-by"""
    E = capi.MAX_EV
    cap, cap_pe = 60_000, 3000
    pipe = capi.Pipeline(gpu_ctx, 1, 64, cap_cand=cap, n_pe=cap_pe, n_flushes=400)
    rng = np.random.default_rng(77)
    cases = [(50_000, 3, 2500, False, False), (700, 1, 0, False, True), (20_000, 5, 900, True, True), (1, 1, 1, False, False),
             (33_333, 2, 3000, True, False), (5000, 8, 100, False, True), (60_000, 1, 10, False, False)]
    total = 0
    for ci, (n_cand, n_ctg, n_pe, pinned, dense) in enumerate(cases):
        rec, cls, b1, b2, pe_b1, pe_b2, flushes = _wide_case(rng, n_cand, n_ctg, n_pe, pinned, dense)
        assert len(flushes) <= 400
        pipe.d_cls.upload(cls); pipe.d_b1.upload(b1); pipe.d_b2.upload(b2)
        pipe.n_pe = n_pe
        pipe.set_pe(pe_b1, pe_b2)
        pipe.d_cand_rec.upload(rec)
        cnt = np.zeros(8, np.int32); cnt[0] = n_cand
        pipe.d_counters.upload(cnt)
        tie = ci & 1
        pipe.flush_groupby(flushes, tie_desc=tie, cand_bound=max(n_cand, 1) if ci % 3 else cap)
        pipe.sync()
        ns = n_cand * E
        a_cls = np.concatenate([cls, np.full(n_pe, 2, np.int32)]); a_b1 = np.concatenate([b1, pe_b1]); a_b2 = np.concatenate([b2, pe_b2])
        want = np.zeros(ns + n_pe, np.int32)
        for k, (r0, r1, pe_hi, marker) in enumerate(flushes):
            lo, hi = int(np.searchsorted(rec, r0)), int(np.searchsorted(rec, r1))
            vis = np.full(ns + n_pe, -1, np.int32)
            vis[lo * E:hi * E] = cls[lo * E:hi * E]; vis[ns:ns + pe_hi] = 2
            ob.flush_cut(vis, a_b1, a_b2, want, marker, k + 1)
        got = pipe.d_consumed.download(np.int32, pipe.n_slots)
        base = cap * E
        assert np.array_equal(got[:ns], want[:ns]), (ci, int((got[:ns] != want[:ns]).sum()))
        assert np.array_equal(got[base:base + n_pe], want[ns:]), ci
        groups = {}
        for s in np.nonzero((cls >= 0) & (cls < 2) & (want[:ns] > 0))[0]:
            groups.setdefault((int(want[s]), int(cls[s]), int(b1[s]), int(b2[s])), []).append(int(s))
        key, first, count, order = pipe.clusters()
        assert int(count.sum()) == len(order) == sum(len(v) for v in groups.values())
        seen = {tuple(int(x) for x in key[c]): [int(x) for x in order[first[c]:first[c] + count[c]]] for c in range(len(key))}
        assert seen == ({k: v[::-1] for k, v in groups.items()} if tie else groups), ci
        total += len(groups)
        if not pinned and n_cand >= 5000 and len(flushes) > 2 * n_ctg:
            assert len({k[0] for k in groups}) > n_ctg            # mid-contig flushes did consume
    assert total > 20_000
