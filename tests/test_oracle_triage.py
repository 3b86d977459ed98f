"""CPU checks of the triage oracle (oracle/im_oracle_triage.c) that need no device: the record
contract of include/indelminer_amd.h (im_dev_records)."""
import struct

import numpy as np

from tests.support import oraclebind as ob


def _rec(flag, cigar, tags=b"", qname=b"q\0", pad=b"\0\0\0", l_seq=100):
    cig = b"".join(struct.pack("<I", (l << 4) | o) for l, o in cigar)
    body = struct.pack("<iiBBHHHiiii", 0, 100, len(qname), 60, 0, len(cigar), flag, l_seq, 0, 300, 300) + qname + cig \
        + bytes([0x12] * ((l_seq + 1) // 2)) + b"\x28" * l_seq + tags
    return body + pad[:(-len(body)) % 4]


def test_alignment_padding_is_not_an_aux_field():
    """records sit at 4-byte aligned offsets, so up to three spare bytes follow the aux area; whatever they hold
    (here the start of an RG / MQ field) must not be read as a tag -- found on a 24-contig BAM where the pinned
    staging buffer's previous contents spelled 'RG' behind a record"""
    S = ((30, 4), (70, 0))
    recs = [_rec(0x3, S, qname=b"qq\0", tags=b"MQC\x0a", pad=b"RGZ"), _rec(0x3, S, qname=b"qq\0", tags=b"MQC\x0a"),
            _rec(0x3, S, qname=b"qqq\0", pad=b"MQ"), _rec(0x3, S, qname=b"qqq\0")]
    assert [len(r) % 4 for r in recs] == [0, 0, 0, 0]
    raw = np.frombuffer(b"".join(recs), dtype=np.uint8).copy()
    off = np.zeros(len(recs) + 1, dtype=np.uint32)
    np.cumsum([len(r) for r in recs], out=off[1:])
    tri = ob.triage_records(raw, off, ["generic"], [700])
    assert tri[0][0].cls == tri[1][0].cls == 3 and tri[2][0].cls == tri[3][0].cls == 3
    assert tri[0][0].range_max == 700


def _random_flush_case(rng, pinned):
    """a group of 1-4 contigs: entries in arrival order with (b1, b2) near their arrival position, flush points with
    non-decreasing markers inside a contig (src/indelminer.c:211-233, 622-623) and INT_MAX at its end (806)"""
    n_ctg = int(rng.integers(1, 5))
    marker, last, arr, cls, b1, b2 = [], [], [], [], [], []
    for _ in range(n_ctg):
        n = int(rng.integers(0, 400))
        pos = np.sort(rng.integers(0, 50_000, n))
        nf = int(rng.integers(0, 7))
        cuts = np.sort(rng.integers(0, n + 1, nf))                 # flush k sees entries [0, cuts[k]) of the contig
        floor = int(rng.integers(0, 3000)) if pinned else 2**31 - 1
        f0 = len(marker)
        mk = -1
        for c in cuts:
            here = int(pos[c - 1]) if c > 0 else 0
            m = min(floor, max(mk, here - int(rng.integers(0, 1500))))   # a pair-table entry further back holds it down
            mk = max(mk, m)
            marker.append(mk)
        marker.append(2**31 - 1)
        f1 = len(marker)
        last += [f1 - 1] * (f1 - f0)
        bounds = list(cuts) + [n]
        for i in range(n):
            arr.append(f0 + next(k for k, c in enumerate(bounds) if i < c))
            live = rng.random() < 0.8
            cls.append(int(rng.integers(0, 3)) if live else -1)
            s = int(pos[i]) + int(rng.integers(-800, 800))
            b1.append(max(s, 0))
            b2.append(max(s, 0) + (int(rng.integers(0, 1200)) if rng.random() < 0.7 else 0))
    return [np.array(x, np.int32) for x in (marker, last, arr, cls, b1, b2)]


def test_flush_marks_need_no_history():
    """imo_flush_nohistory (what im_dev_flush_groupby computes: every flush's cut from all entries at once) against the
    reference's own order of events -- imo_flush_cut flush by flush over what is still pending (src/indelminer.c:123-146)"""
    rng = np.random.default_rng(20261004)
    consumed_somewhere = 0
    for case in range(600):
        marker, last, arr, cls, b1, b2 = _random_flush_case(rng, pinned=case % 3 == 0)
        ids = np.arange(1, len(marker) + 1, dtype=np.int32)
        want = np.zeros(len(cls), np.int32)
        for f in range(len(marker)):
            first = int(np.searchsorted(last, last[f]))            # the contig's first flush
            vis = np.where((arr >= first) & (arr <= f), cls, -1).astype(np.int32)
            ob.flush_cut(vis, b1, b2, want, int(marker[f]), int(ids[f]))
        got = ob.flush_nohistory(marker, ids, last, cls, b1, b2, arr)
        assert got is not None and np.array_equal(got, want), case
        consumed_somewhere += int((want[cls >= 0] < ids[last[arr[cls >= 0]]]).sum()) if len(cls) else 0
    assert consumed_somewhere > 5000          # mid-contig flushes did consume entries: the comparison is not vacuous


def test_flush_nohistory_refuses_decreasing_markers():
    marker = np.array([500, 300, 2**31 - 1], np.int32)
    z = np.zeros(1, np.int32)
    assert ob.flush_nohistory(marker, [1, 2, 3], [2, 2, 2], z, z, z, z) is None


def test_oracle_reads_both_record_forms():
    """the delivered-record contract (include/indelminer_amd.h, im_dev_records): a record with bin = 0xFFFF travels without its
    base qualities.  The oracle's record rules and its pileup depth rule must give the same answers on both forms -- bench.py's
    DP= self-check fed the quality-less form to an oracle that could not parse it and compared against an all-zero depth."""
    from indelminer_amd import rawrec, synth
    refs, rd = synth.simulate(seed=3, ref_len=30_000, coverage=12, big_every=5)
    raw, off = rawrec.records(rd)
    raw2, off2 = rawrec.records(rd, qual=False)
    assert len(raw2) < len(raw) and len(off) == len(off2)
    clen = len(refs[0])
    d1 = ob.depth_of(raw, off, 0, clen)
    d2 = ob.depth_of(raw2, off2, 0, clen)
    assert d1.sum() > 0 and np.array_equal(d1, d2)
    t1 = ob.triage_records(raw, off, ["generic"], [rd.range_max])
    t2 = ob.triage_records(raw2, off2, ["generic"], [rd.range_max])
    assert sum(1 for t, _ in t1 if t.cls in (2, 3)) > 0
    for (a, ba), (b, bb) in zip(t1, t2):
        assert (a.cls, a.tid, a.anchor, a.range_max, a.n_ev, a.l_seq) == (b.cls, b.tid, b.anchor, b.range_max, b.n_ev, b.l_seq)
        assert ba == bb
