"""The general realign pass (indelminer_amd/csrc/im_realign_any.hip): what the reference takes and the laid-out kernels
do not -- reads beyond 255 bases with -g > 0, reads beyond 1020 bases, bands wider than a wave (-g > 60) -- against the
CPU oracle (itself pinned to the compiled reference, tests/test_oracle_vs_ref.py).  Bit-exact: integer / index work.
Reference: src/readaln.c:242-267 (any read length), src/indelminer.c:934,948 (numgaps unbounded),
src/alignment.c:393-447, src/localalign.c:15-196, src/globalalign.c:66-401."""
import random

import numpy as np
import pytest

from tests.support import gpucmp, leftedge, oraclebind as ob

pytestmark = pytest.mark.gpu


def _batch(seed, n, clen, lengths, Rm, max_indel=40, err=0.005):
    rng = random.Random(seed)
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    cases = []
    for _ in range(n):
        L = min(rng.choice(lengths), clen - 100)
        anchor = rng.randint(0, clen - 1)
        p = max(0, min(clen - L - max_indel - 2, anchor + rng.randint(-Rm + 50, Rm - 50)))
        typ = rng.random()
        cut = rng.randint(12, L - 12)
        d = rng.randint(1, max_indel)
        if typ < 0.45:
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.85:
            read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
        else:
            read = contig[p:p + L]
        read = "".join((rng.choice("ACGT") if rng.random() < err else ch) for ch in read)
        cases.append(dict(anchor=anchor, range_max=Rm, read=read))
    return contig.encode(), cases


def _compare(ctx, capi, kw, contig, cases, tag):
    reads = [c["read"].encode() for c in cases]
    n = len(cases)
    rc, out = ctx.realign_batch(capi.params(**kw), reads, np.zeros(n, np.int32),
                                np.array([c["anchor"] for c in cases], np.int32),
                                np.array([c["range_max"] for c in cases], np.int32),
                                allow=(capi.E_ABORT, capi.E_OVERFLOW))
    P = ob.params(**kw)
    bad = []
    for i, c in enumerate(cases):
        st, res = ob.realign(P, contig, len(contig), c["anchor"], c["range_max"], c["read"])
        msg = gpucmp.hip_vs_oracle(out[i], st, res)
        if msg:
            bad.append((i, msg, len(c["read"]), c["anchor"], c["range_max"]))
    assert not bad, "%s %r: %d of %d differ, first: %r" % (tag, kw, len(bad), n, bad[0])
    assert not (out["status"] == capi.ST_UNSUPPORTED).any()
    return out


@pytest.mark.parametrize("L", [256, 300, 1020])
@pytest.mark.parametrize("g", [1, 5, 12])
def test_long_reads_with_gaps(gpu_ctx, L, g):
    """-g > 0 on reads beyond 255 bases (a 2 x 300 library with -g 2 is a plausible run), mixed with reads the band kernel takes"""
    from indelminer_amd import capi
    contig, cases = _batch(7000 + L + g, 160, 60000, [L, L, L, 100, 250], 900, max_indel=max(12, 2 * g))
    gpu_ctx.set_reference([contig])
    out = _compare(gpu_ctx, capi, dict(klength=6, numgaps=g, maxdelsize=1000), contig, cases, "long+gaps")
    is_long = np.array([len(c["read"]) > capi.SHORT_READ for c in cases])
    assert int((out["status"][is_long] == 1).sum()) > 8 and int((out["status"][~is_long] == 1).sum()) > 5


@pytest.mark.parametrize("k,lengths,Rm,maxdel", [(6, [1021, 1500, 100, 300], 2000, 1000), (6, [2500, 4000], 5000, 3000),
                                                 (9, [1100, 2047, 2048, 2049], 2500, 1000), (13, [1300], 1500, 500)])
def test_reads_beyond_1020_bases(gpu_ctx, k, lengths, Rm, maxdel):
    """-g 0, reads beyond the long-read kernel's 1020 bases, mixed with reads of the two laid-out kernels"""
    from indelminer_amd import capi
    contig, cases = _batch(8000 + k + lengths[0], 120, 90000, lengths, Rm, max_indel=60, err=0.002)
    gpu_ctx.set_reference([contig])
    out = _compare(gpu_ctx, capi, dict(klength=k, numgaps=0, maxdelsize=maxdel), contig, cases, "beyond 1020")
    assert int((out["status"] == 1).sum()) > 4


@pytest.mark.parametrize("k,g,lengths", [(6, 61, [100, 150, 250]), (6, 100, [100, 250, 400]), (8, 200, [76, 150]), (4, 333, [150, 600])])
def test_bands_wider_than_a_wave(gpu_ctx, k, g, lengths):
    """-g above 60: every read takes the general pass (the band kernel holds a band in one wave); indels up to the band width"""
    from indelminer_amd import capi
    contig, cases = _batch(9000 + g, 90, 30000, lengths, 705, max_indel=g, err=0.01)
    gpu_ctx.set_reference([contig])
    out = _compare(gpu_ctx, capi, dict(klength=k, numgaps=g, maxdelsize=1000), contig, cases, "wide band")
    assert int((out["status"] == 1).sum()) > 4


def test_general_pass_fuzzed(gpu_ctx):
    """every k, odd -s / -n, short contigs (windows clipped at both ends), anchors on the contig's ends, lengths around every
    border between the kernels, bands on both sides of a wave's width"""
    from indelminer_amd import capi
    rng = random.Random(4242)
    for trial in range(24):
        g = rng.choice([0, 0, 1, 3, 12, 60, 61, 75, 130])
        kw = dict(klength=rng.choice([2, 4, 5, 6, 6, 7, 9, 12, 15]), numgaps=g,
                  maxdelsize=rng.choice([50, 300, 1000, 2500]), ethreshold=rng.choice([1, 5, 10, 25]))
        clen = rng.choice([1500, 3000, 20000])
        contig = "".join(rng.choice("ACGT") for _ in range(clen))
        cases = []
        for _ in range(36):
            L = min(rng.choice([4, 36, 100, 255, 256, 300, 1020, 1021, 1400]), clen - 10)
            Rm = rng.choice([200, 705, 1500, 3000])
            anchor = rng.choice([0, clen - 1, rng.randint(0, clen - 1)])
            p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
            cut = rng.randint(1, max(1, L - 1))
            d = rng.randint(1, 60)
            typ = rng.random()
            if typ < 0.45:
                read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
            elif typ < 0.8:
                read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
            else:
                read = contig[p:p + L]
            read = "".join((rng.choice("ACGTN") if rng.random() < 0.006 else ch) for ch in read)
            if len(read) < 4:
                read = contig[:4]
            cases.append(dict(anchor=anchor, range_max=Rm, read=read))
        gpu_ctx.set_reference([contig.encode()])
        _compare(gpu_ctx, capi, kw, contig.encode(), cases, "fuzz %d" % trial)


@pytest.mark.parametrize("k,g,seed", [(6, 61, 1), (6, 90, 2), (4, 64, 3)])
def test_general_pass_left_edge_bands(gpu_ctx, k, g, seed):
    """bands that hang off the window's left edge: local_align's reverse pass looks at the byte in front of the window
    (src/localalign.c:144-176 has no `ib > 0` guard); -g > 60 sends every read through the general pass"""
    from indelminer_amd import capi
    contig, cases = leftedge.cases(seed)
    gpu_ctx.set_reference([contig.encode()])
    _compare(gpu_ctx, capi, dict(klength=k, numgaps=g), contig.encode(), cases, "left edge")


@pytest.mark.parametrize("k,g", [(3, 0), (6, 0), (6, 2), (3, 70)])
def test_long_reads_hanging_off_the_contig_start(gpu_ctx, k, g):
    """A long read against a narrow window at the contig's start: the chosen band hangs thousands of bases off the window's left
    edge, the forward pass finds no positive cell, and local_align's reverse pass -- which has no `ib > 0` guard
    (src/localalign.c:144-176) -- walks all the way out in front of the contig.  The reference reads whatever lies there (and returns
    no alignment whatever it reads); the general pass must not fault (profiles/any_fuzz.py seed 2 did, in round 4)."""
    from indelminer_amd import capi
    rng = random.Random(500 + k + g)
    clen = 3000
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    cases = []
    for _ in range(48):
        L = rng.choice([1100, 2047, 2600]) if g == 0 else rng.choice([300, 1100, 2600])
        read = "".join(rng.choice("ACGT") for _ in range(L))
        if rng.random() < 0.3:
            read = read[:L - 20] + contig[:20]
        cases.append(dict(anchor=rng.choice([0, 0, 1, 7, 59]), range_max=rng.choice([60, 200]), read=read))
    gpu_ctx.set_reference([contig.encode()])
    _compare(gpu_ctx, capi, dict(klength=k, numgaps=g, maxdelsize=50), contig.encode(), cases, "off the contig's start")


def test_general_pass_on_the_device_entry(gpu_ctx):
    """im_dev_realign_keep with evidence slots: a read the general pass realigns replaces its CIGAR-derived slots, one it finds
    nothing for keeps them (src/indelminer.c:494-512)"""
    from indelminer_amd import capi
    contig, cases = _batch(5151, 64, 40000, [1200, 1500], 1500, max_indel=30, err=0.002)
    gpu_ctx.set_reference([contig])
    reads = [c["read"].encode() for c in cases]
    n = len(reads)
    gpu_ctx.expect_read_length(max(len(r) for r in reads))
    offs = np.zeros(n + 1, np.int64)
    np.cumsum([(len(r) + 3) & ~3 for r in reads], out=offs[1:])
    bases = np.zeros(int(offs[-1]) + 16, np.uint8)
    for i, r in enumerate(reads):
        bases[offs[i]:offs[i] + len(r)] = np.frombuffer(r, np.uint8)
    d = {}
    d["bases"] = capi.DevBuf(gpu_ctx, bases.nbytes); d["bases"].upload(bases)
    d["off"] = capi.DevBuf(gpu_ctx, 8 * n); d["off"].upload(offs[:n].copy())
    for name, arr in (("len", np.array([len(r) for r in reads], np.int32)), ("tid", np.zeros(n, np.int32)),
                      ("anchor", np.array([c["anchor"] for c in cases], np.int32)),
                      ("range", np.array([c["range_max"] for c in cases], np.int32))):
        d[name] = capi.DevBuf(gpu_ctx, 4 * n); d[name].upload(arr)
    d["res"] = capi.DevBuf(gpu_ctx, capi.RESULT_DTYPE.itemsize * n)
    seeded = np.full(n * capi.MAX_EV, 77, np.int32)
    for name in ("cls", "b1", "b2"):
        d[name] = capi.DevBuf(gpu_ctx, 4 * n * capi.MAX_EV); d[name].upload(seeded)
    batch = capi.DevBatch(n, d["bases"].ptr, d["off"].ptr, d["len"].ptr, d["tid"].ptr, d["anchor"].ptr, d["range"].ptr,
                          d["res"].ptr, d["cls"].ptr, d["b1"].ptr, d["b2"].ptr)
    import ctypes as C
    P_hip = capi.params()
    gpu_ctx._check(capi.lib().im_dev_realign_keep(gpu_ctx.h, C.byref(P_hip), C.byref(batch), gpu_ctx.stream))
    gpu_ctx._check(capi.lib().im_stream_sync(gpu_ctx.h, gpu_ctx.stream))
    out = d["res"].download(capi.RESULT_DTYPE, n)
    cls = d["cls"].download(np.int32, n * capi.MAX_EV).reshape(n, capi.MAX_EV)
    b1 = d["b1"].download(np.int32, n * capi.MAX_EV).reshape(n, capi.MAX_EV)
    P = ob.params()
    n_ev = 0
    for i, c in enumerate(cases):
        st, res = ob.realign(P, contig, len(contig), c["anchor"], c["range_max"], c["read"])
        assert gpucmp.hip_vs_oracle(out[i], st, res) is None, i
        if st == 1:
            n_ev += 1
            assert [int(x) for x in cls[i][:res.n_ev]] == [res.ev[k].cls for k in range(res.n_ev)]
            assert [int(x) for x in b1[i][:res.n_ev]] == [res.ev[k].b1 for k in range(res.n_ev)]
            assert (cls[i][res.n_ev:] == -1).all()
        else:
            assert (cls[i] == 77).all()
    assert n_ev > 4
