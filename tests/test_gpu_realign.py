"""Parity of the HIP realign kernel (through the C ABI) with the CPU oracle and the
reference's golden vectors.  Bit-exact: integer / index work."""
import os
import random

import numpy as np
import pytest

from tests.support import bamlite, golden, gpucmp, oraclebind as ob

pytestmark = pytest.mark.gpu


def _run_cases(ctx, capi, P_hip, P_or, contig, cases, dump=None):
    reads = [c["read"].encode() for c in cases]
    n = len(cases)
    rc, out = ctx.realign_batch(P_hip, reads, np.zeros(n, np.int32),
                                np.array([c["anchor"] for c in cases], np.int32),
                                np.array([c["range_max"] for c in cases], np.int32),
                                allow=(capi.E_ABORT,))
    bad = []
    for i, c in enumerate(cases):
        st, res = ob.realign(P_or, contig, len(contig), c["anchor"], c["range_max"], c["read"])
        msg = gpucmp.hip_vs_oracle(out[i], st, res)
        if msg:
            bad.append((i, msg, c["anchor"], c["range_max"], c["read"]))
    if bad and dump:
        os.makedirs(os.path.dirname(dump), exist_ok=True)
        with open(dump, "w") as fh:
            for b in bad:
                fh.write(repr(b) + "\n")
    return out, bad


def test_hip_matches_oracle_and_golden_on_test_data(gpu_ctx, golden_dir):
    from indelminer_amd import capi
    g = golden.load("realign_testdata.json")
    _, seqs = bamlite.read_fasta(os.path.join(golden_dir, "test_data", "reference.fa"))
    contig = seqs[0].encode()
    gpu_ctx.set_reference([contig])
    out, bad = _run_cases(gpu_ctx, capi, capi.params(**g["params"]), ob.params(**g["params"]), contig, g["cases"],
                          dump="gpurun_out/mismatch_testdata.txt")
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(g["cases"]), bad[0][:2])
    # and straight against what the reference itself returned
    n_ev = 0
    for rec, c in zip(out, g["cases"]):
        ops = [int(x) for x in rec["ops"][:int(rec["n_ops"])]] if rec["status"] == 1 else []
        evs = [(int(e["cls"]), int(e["b1"]), int(e["b2"]), int(e["seg"])) for e in rec["ev"][:int(rec["n_ev"])]] if rec["status"] == 1 else []
        msg = golden.golden_vs_segments(c["ref"], int(rec["status"]), int(rec["ref_start"]), ops, evs)
        assert msg is None, (c["qname"], msg)
        n_ev += len(evs)
    assert n_ev == 443


def test_hip_matches_oracle_on_synthetic_golden_groups(gpu_ctx):
    from indelminer_amd import capi
    groups = golden.load("realign_synth.json")        # -k 4..12, -g 0..5
    assert len(groups) >= 10
    for gi, grp in enumerate(groups):
        contig = grp["contig"].encode()
        gpu_ctx.set_reference([contig])
        cases = [c for c in grp["cases"] if len(c["read"]) <= (capi.MAX_READ if grp["params"].get("numgaps", 0) == 0 else capi.SHORT_READ)]
        out, bad = _run_cases(gpu_ctx, capi, capi.params(**grp["params"]), ob.params(**grp["params"]), contig, cases,
                              dump="gpurun_out/mismatch_synth_%d.txt" % gi)
        assert not bad, "%r: %d of %d differ, first: %r" % (grp["params"], len(bad), len(cases), bad[0][:2])
        for rec, c in zip(out, cases):
            ops = [int(x) for x in rec["ops"][:int(rec["n_ops"])]] if rec["status"] == 1 else []
            evs = [(int(e["cls"]), int(e["b1"]), int(e["b2"]), int(e["seg"])) for e in rec["ev"][:int(rec["n_ev"])]] if rec["status"] == 1 else []
            msg = golden.golden_vs_segments(c["ref"], int(rec["status"]), int(rec["ref_start"]), ops, evs)
            assert msg is None, (grp["params"], msg)


def _synthetic_batch(seed, n, clen=200000, L=100, Rm=700):
    rng = random.Random(seed)
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    cases = []
    for _ in range(n):
        anchor = rng.randint(0, clen - 1)
        p = max(0, min(clen - L - 80, anchor + rng.randint(-Rm + 50, Rm - 150)))
        typ = rng.random()
        cut = rng.randint(12, L - 12)
        if typ < 0.5:
            d = rng.randint(1, 50)
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.9:
            d = rng.randint(1, 30)
            ins = "".join(rng.choice("ACGT") for _ in range(d))
            read = (contig[p:p + cut] + ins + contig[p + cut:p + L])[:L]
        else:
            read = contig[p:p + L]
        read = "".join((rng.choice("ACGT") if rng.random() < 0.005 else ch) for ch in read)
        cases.append(dict(anchor=anchor, range_max=Rm, read=read))
    return contig.encode(), cases


@pytest.mark.parametrize("k,g,Rm,maxdel", [(6, 0, 1500, 1000), (6, 0, 700, 6000), (6, 0, 2600, 9000), (5, 0, 1200, 1000),
                                           (9, 0, 1500, 3000), (6, 2, 1500, 1000)])
def test_hip_matches_oracle_on_wide_windows(gpu_ctx, k, g, Rm, maxdel):
    """Windows wider than one histogram pass (1920 diagonals): the vote runs in chunks, the chunk's best band
    is carried over, window starts are re-based per chunk.  Insert-size range and -s chosen so that the first,
    the second, or both band searches of a read need 2 to 6 chunks; k = 6 (specialised), k = 5 (direct table),
    k = 9 (hash table) and the gapped kernel all share that loop."""
    from indelminer_amd import capi
    contig, cases = _synthetic_batch(300 + k + Rm, 1200, clen=120000, Rm=Rm)
    gpu_ctx.set_reference([contig])
    kw = dict(klength=k, numgaps=g, maxdelsize=maxdel)
    out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig, cases,
                          dump="gpurun_out/mismatch_wide_k%d_g%d_R%d.txt" % (k, g, Rm))
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    assert int((out["status"] == 1).sum()) > 300
    assert int(out["band"]["win_bytes"].max()) > 1920


@pytest.mark.parametrize("k,L,Rm,maxdel", [(6, 300, 900, 1000), (6, 256, 700, 1000), (6, 1020, 2500, 3000), (5, 511, 1500, 6000), (9, 300, 900, 1000),
                                           (13, 750, 2000, 1000), (6, 301, 5000, 9000)])
def test_hip_long_reads_match_oracle(gpu_ctx, k, L, Rm, maxdel):
    """Reads of 256 .. IM_MAX_READ bases: the second kernel (sixteen read positions per lane, 16-bit table offsets and vote
    counters; indelminer_amd/csrc/im_realign_long.hip), mixed into one batch with reads of 100 bases that the first
    kernel takes.  Windows of one to several histogram passes, k = 6 / 5 (direct table), 9 / 13 (hash)."""
    from indelminer_amd import capi
    contig, long_cases = _synthetic_batch(4000 + k + L, 500, clen=150000, L=L, Rm=Rm)
    contig_again, short_cases = _synthetic_batch(4000 + k + L, 300, clen=150000, L=100, Rm=Rm)      # same seed, same contig
    assert contig_again == contig
    cases = long_cases + short_cases
    random.Random(L).shuffle(cases)
    gpu_ctx.set_reference([contig])
    kw = dict(klength=k, numgaps=0, maxdelsize=maxdel)
    out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig, cases,
                          dump="gpurun_out/mismatch_long_k%d_L%d.txt" % (k, L))
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    is_long = np.array([len(c["read"]) > capi.SHORT_READ for c in cases])
    assert int((out["status"][is_long] == 1).sum()) > 50 and int((out["status"][~is_long] == 1).sum()) > 50


def test_hip_long_reads_fuzzed(gpu_ctx, seed=31):
    """lengths around the two kernels' border and the upper bound, short contigs (windows clipped at both ends), every k"""
    from indelminer_amd import capi
    rng = random.Random(seed)
    for trial in range(30):
        k = rng.choice([2, 3, 4, 5, 6, 6, 7, 8, 10, 12, 15])
        kw = dict(klength=k, numgaps=0, maxdelsize=rng.choice([50, 300, 1000, 2500]), ethreshold=rng.choice([1, 5, 10, 25]))
        clen = rng.choice([1100, 2500, 20000])
        contig = "".join(rng.choice("ACGT") for _ in range(clen))
        cases = []
        for _ in range(60):
            L = min(rng.choice([100, 255, 256, 257, 300, 400, 512, 777, 1019, 1020]), clen - 10)
            Rm = rng.choice([200, 705, 1500, 3000])
            anchor = rng.choice([0, clen - 1, rng.randint(0, clen - 1)])
            p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
            cut = rng.randint(1, max(1, L - 1))
            d = rng.randint(1, 60)
            typ = rng.random()
            if typ < 0.45:
                read = (contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)])
            elif typ < 0.8:
                read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
            else:
                read = contig[p:p + L]
            read = "".join((rng.choice("ACGT") if rng.random() < 0.004 else ch) for ch in read)
            if len(read) < 4:
                read = contig[:4]
            cases.append(dict(anchor=anchor, range_max=Rm, read=read))
        gpu_ctx.set_reference([contig.encode()])
        out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig.encode(), cases,
                              dump="gpurun_out/mismatch_longfuzz_%d.txt" % trial)
        assert not bad, "%r: %d of %d differ, first: %r" % (kw, len(bad), len(cases), bad[0][:2])


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_hip_matches_oracle_fuzzed_parameters(gpu_ctx, seed):
    """Every k the command line accepts (2..15), odd -n / -s values, short contigs (windows clipped at both
    ends), reads of 4..255 bases, anchors on the contig's first and last bases: 40 parameter sets x 60 reads."""
    from indelminer_amd import capi
    rng = random.Random(seed)
    for trial in range(40):
        k = rng.choice([2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14, 15])
        kw = dict(klength=k, numgaps=0, maxdelsize=rng.choice([50, 300, 1000, 2500]), ethreshold=rng.choice([1, 5, 10, 25]))
        clen = rng.choice([300, 900, 2500, 20000])
        contig = "".join(rng.choice("ACGT") for _ in range(clen))
        cases = []
        for _ in range(60):
            L = rng.choice([4, 17, 36, 76, 100, 101, 150, 250, 255])
            L = min(L, clen - 10)
            Rm = rng.choice([60, 200, 705, 1500])
            anchor = rng.choice([0, clen - 1, rng.randint(0, clen - 1)])
            p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
            cut = rng.randint(1, max(1, L - 1))
            d = rng.randint(1, 60)
            typ = rng.random()
            if typ < 0.45:
                read = (contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)])
            elif typ < 0.8:
                read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
            else:
                read = contig[p:p + L]
            if len(read) < 4:
                read = contig[:4]
            cases.append(dict(anchor=anchor, range_max=Rm, read=read))
        gpu_ctx.set_reference([contig.encode()])
        out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig.encode(), cases,
                              dump="gpurun_out/mismatch_fuzz_%d_%d.txt" % (seed, trial))
        assert not bad, "%r: %d of %d differ, first: %r" % (kw, len(bad), len(cases), bad[0][:2])


def test_hip_gapped_matches_oracle_fuzzed_parameters(gpu_ctx):
    """The banded affine-gap kernel over -g 1..12 and k 4..12 on short contigs and odd read lengths: every
    comparison of the reference's traceback keeps its strictness, so co-optimal paths must come out the same."""
    from indelminer_amd import capi
    rng = random.Random(99)
    for trial in range(14):
        kw = dict(klength=rng.choice([4, 6, 6, 8, 12]), numgaps=rng.choice([1, 2, 3, 5, 8, 12]),
                  maxdelsize=rng.choice([300, 1000]), ethreshold=rng.choice([5, 10]))
        clen = rng.choice([900, 4000, 20000])
        contig = "".join(rng.choice("ACGT") for _ in range(clen))
        cases = []
        for _ in range(40):
            L = min(rng.choice([36, 76, 100, 150, 250]), clen - 10)
            Rm = rng.choice([200, 705])
            anchor = rng.randint(0, clen - 1)
            p = max(0, min(clen - L, anchor + rng.randint(-Rm, Rm)))
            cut = rng.randint(1, max(1, L - 1))
            d = rng.randint(1, 12)
            typ = rng.random()
            if typ < 0.45:
                read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
            elif typ < 0.85:
                read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
            else:
                read = contig[p:p + L]
            read = "".join((rng.choice("ACGT") if rng.random() < 0.01 else ch) for ch in read)
            if len(read) < 4:
                read = contig[:4]
            cases.append(dict(anchor=anchor, range_max=Rm, read=read))
        gpu_ctx.set_reference([contig.encode()])
        out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig.encode(), cases,
                              dump="gpurun_out/mismatch_gapfuzz_%d.txt" % trial)
        assert not bad, "%r: %d of %d differ, first: %r" % (kw, len(bad), len(cases), bad[0][:2])


@pytest.mark.parametrize("k", [6, 7])
def test_hip_matches_oracle_with_ambiguity_codes(gpu_ctx, k):
    """N / IUPAC bytes in the contig and N in the reads: the k-mer code maps them to 0 (src/alignment.c:11-24)
    while the alignment compares raw bytes (N matches N only, src/localalign.c:61-67) -- the packed and the
    ASCII copy of the reference must disagree in exactly that way."""
    from indelminer_amd import capi
    contig, cases = _synthetic_batch(900 + k, 1500, clen=60000)
    rng = random.Random(77)
    cb = bytearray(contig)
    for _ in range(600):
        cb[rng.randrange(len(cb))] = ord(rng.choice("NRYKM"))
    for start in (5000, 20000, 41000):
        cb[start:start + rng.randint(30, 400)] = b"N" * len(cb[start:start + rng.randint(30, 400)])
    contig = bytes(cb)
    for c in cases[::3]:
        r = list(c["read"])
        r[rng.randrange(len(r))] = "N"
        c["read"] = "".join(r)
    gpu_ctx.set_reference([contig])
    out, bad = _run_cases(gpu_ctx, capi, capi.params(klength=k), ob.params(klength=k), contig, cases,
                          dump="gpurun_out/mismatch_iupac_k%d.txt" % k)
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    assert int((out["status"] == 1).sum()) > 150


@pytest.mark.parametrize("k", [6, 8])
def test_hip_matches_oracle_on_seeded_batch(gpu_ctx, k):
    """5000 seeded reads with planted 1-50 bp indels (BASELINE config-2 shape, smaller)."""
    from indelminer_amd import capi
    contig, cases = _synthetic_batch(100 + k, 5000)
    gpu_ctx.set_reference([contig])
    out, bad = _run_cases(gpu_ctx, capi, capi.params(klength=k), ob.params(klength=k), contig, cases,
                          dump="gpurun_out/mismatch_seeded_k%d.txt" % k)
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    assert int((out["status"] == 1).sum()) > 2000


def test_edge_cases(gpu_ctx):
    """Windows clipped at both contig ends, reads shorter than k, all-N reads, empty batch."""
    from indelminer_amd import capi
    rng = random.Random(7)
    contig = "".join(rng.choice("ACGT") for _ in range(1500))
    cb = contig.encode()
    gpu_ctx.set_reference([cb])
    cases = []
    for anchor in (0, 1, 5, 700, 1400, 1499, 1500):
        for p in (0, 3, 600, 1380, 1400):
            cases.append(dict(anchor=anchor, range_max=705, read=contig[p:p + 40] + contig[p + 60:p + 120]))
            cases.append(dict(anchor=anchor, range_max=50, read=contig[p:p + 100][:max(4, 100 - p % 7)]))
    cases.append(dict(anchor=700, range_max=705, read="ACG"))
    cases.append(dict(anchor=700, range_max=705, read="N" * 100))
    cases.append(dict(anchor=700, range_max=705, read="A" * 100))
    cases = [c for c in cases if len(c["read"]) > 0]
    out, bad = _run_cases(gpu_ctx, capi, capi.params(), ob.params(), cb, cases, dump="gpurun_out/mismatch_edge.txt")
    assert not bad, bad[0][:2]
    rc, out = gpu_ctx.realign_batch(capi.params(), [], [], [], [])
    assert rc == 0 and len(out) == 0


@pytest.mark.parametrize("k,g", [(6, 1), (6, 3), (8, 2)])
def test_hip_gapped_matches_oracle_on_seeded_batch(gpu_ctx, k, g):
    """-g > 0: banded affine-gap path (local_align + ALIGN traceback on the device)."""
    from indelminer_amd import capi
    contig, cases = _synthetic_batch(300 + 10 * k + g, 1500)
    gpu_ctx.set_reference([contig])
    out, bad = _run_cases(gpu_ctx, capi, capi.params(klength=k, numgaps=g), ob.params(klength=k, numgaps=g), contig, cases,
                          dump="gpurun_out/mismatch_gapped_k%d_g%d.txt" % (k, g))
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    assert int((out["status"] == 1).sum()) > 500


@pytest.mark.parametrize("k,g,seed", [(6, 0, 1), (6, 0, 2), (6, 1, 3), (6, 2, 4), (6, 5, 5), (8, 3, 6), (4, 0, 7), (6, 12, 8)])
def test_hip_left_edge_bands(gpu_ctx, k, g, seed):
    """The constructed left-edge cases of tests/support/leftedge.py (bands without a k-mer vote that hang off the window's
    left edge, -g 0 and -g > 0): the oracle is pinned on them against the reference compiled from its own sources
    (tests/test_oracle_vs_ref.py::test_left_edge_bands_against_reference); the kernels must agree with the oracle."""
    from indelminer_amd import capi
    from tests.support import leftedge
    contig, cases = leftedge.cases(seed)
    gpu_ctx.set_reference([contig.encode()])
    kw = dict(klength=k, numgaps=g, maxdelsize=1000, ethreshold=10)
    out, bad = _run_cases(gpu_ctx, capi, capi.params(**kw), ob.params(**kw), contig.encode(), cases,
                          dump="gpurun_out/mismatch_leftedge_%d.txt" % seed)
    assert not bad, "%r: %d of %d differ, first: %r" % (kw, len(bad), len(cases), bad[0][:2])


@pytest.mark.parametrize("k,g", [(6, 20), (6, 40), (8, 60), (4, 33)])
def test_hip_gapped_wide_bands(gpu_ctx, k, g):
    """the widest bands the library accepts (-g up to 60: 61 diagonals across one wave), indels up to the band width"""
    from indelminer_amd import capi
    rng = random.Random(1000 + g)
    clen = 6000
    contig = "".join(rng.choice("ACGT") for _ in range(clen))
    cases = []
    for _ in range(60):
        L = rng.choice([100, 150, 250])
        anchor = rng.randint(0, clen - 1)
        p = max(0, min(clen - L - 80, anchor + rng.randint(-600, 450)))
        cut = rng.randint(20, L - 20)
        d = rng.randint(1, g)
        typ = rng.random()
        if typ < 0.45:
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        elif typ < 0.9:
            read = (contig[p:p + cut] + "".join(rng.choice("ACGT") for _ in range(d)) + contig[p + cut:p + L])[:L]
        else:
            read = contig[p:p + L]
        read = "".join((rng.choice("ACGT") if rng.random() < 0.01 else ch) for ch in read)
        cases.append(dict(anchor=anchor, range_max=705, read=read))
    gpu_ctx.set_reference([contig.encode()])
    out, bad = _run_cases(gpu_ctx, capi, capi.params(klength=k, numgaps=g), ob.params(klength=k, numgaps=g), contig.encode(), cases,
                          dump="gpurun_out/mismatch_wideband_g%d.txt" % g)
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    assert int((out["status"] == 1).sum()) > 10


def test_hip_matches_oracle_on_config2_at_full_size(gpu_ctx):
    """BASELINE configs[1] at its full size -- the batch bench.py times: every candidate read of the seeded
    1 Mb / 30x / 100 bp data set (about 12 000 of 300 000 reads, chosen by the reference's own candidate rule)
    against the oracle, record by record; and the size-independent properties on top: the call is idempotent,
    and nearly every planted indel of the seeded donor is found at its exact breakpoint by some read."""
    from indelminer_amd import capi, synth
    refs, rd = synth.simulate(seed=1, ref_len=1_000_000, coverage=30)
    cand = synth.candidates(rd)
    contig = refs[0].tobytes()
    gpu_ctx.set_reference([contig])
    cases = [dict(anchor=int(a), range_max=int(r), read=bytes(b).decode())
             for a, r, b in zip(cand["anchor"], cand["range_max"], cand["bases"])]
    assert 10000 < len(cases) < 15000 and rd.n == 300000
    out, bad = _run_cases(gpu_ctx, capi, capi.params(), ob.params(), contig, cases, dump="gpurun_out/mismatch_config2.txt")
    assert not bad, "%d of %d differ, first: %r" % (len(bad), len(cases), bad[0][:2])
    rc, again = gpu_ctx.realign_batch(capi.params(), [c["read"].encode() for c in cases], np.zeros(len(cases), np.int32),
                                      cand["anchor"].astype(np.int32), cand["range_max"].astype(np.int32))
    ok = out["status"] == 1
    assert np.array_equal(out["status"], again["status"]) and np.array_equal(out["ev"][:, 0][ok], again["ev"][:, 0][ok])
    assert int(ok.sum()) > 6000
    # deletions found = distinct (b1, b2) pairs with b2 > b1 supported by >= 3 reads: about one per 4 kb was planted
    ev = out["ev"][:, 0][ok]
    dels = ev[ev["cls"] == 1]
    keys, support = np.unique(np.stack([dels["b1"], dels["b2"]], axis=1), axis=0, return_counts=True)
    assert int((support >= 3).sum()) > 150


def test_host_entry_pipelines_large_batches(gpu_ctx):
    """im_realign_batch cuts a batch into 32768-read chunks that overlap packing, PCIe and the kernel (pinned
    staging, three streams).  70 000 reads = the same 5 000 reads fourteen times: every copy must come back
    identical to the first, and the first 5 000 identical to a one-chunk call."""
    from indelminer_amd import capi
    contig, cases = _synthetic_batch(4242, 5000)
    gpu_ctx.set_reference([contig])
    reads = [c["read"].encode() for c in cases]
    anchor = [c["anchor"] for c in cases]
    rng = [c["range_max"] for c in cases]
    rc, small = gpu_ctx.realign_batch(capi.params(), reads, [0] * 5000, anchor, rng)
    rc, big = gpu_ctx.realign_batch(capi.params(), reads * 14, [0] * 70000, anchor * 14, rng * 14)
    assert len(big) == 70000

    def canon(r):       # the defined part of a record: ops / ev past their counts are not written by the kernel
        ok = r["status"] == 1
        ops = np.where(ok[:, None] & (np.arange(r["ops"].shape[1])[None, :] < r["n_ops"][:, None]), r["ops"], 0)
        ev = r["ev"][:, 0]
        evs = [np.where(ok & (r["n_ev"] > 0), ev[f], 0) for f in ("cls", "b1", "b2", "seg", "read_off", "lflank", "rflank", "nd_print", "nd_filter")]
        return [r["status"], np.where(ok, r["ref_start"], 0), np.where(ok, r["n_ops"], 0), np.where(ok, r["n_ev"], 0),
                r["n_band"], ops] + evs + [r["band"][f] for f in r["band"].dtype.names]

    want = canon(small)
    assert int((small["status"] == 1).sum()) > 2000
    for rep in range(14):
        got = canon(big[rep * 5000:(rep + 1) * 5000])
        for a, b in zip(want, got):
            assert np.array_equal(a, b), "copy %d differs" % rep


def test_what_used_to_be_unsupported_runs(gpu_ctx):
    """Until round 4 the library turned away -g above 60, reads beyond IM_MAX_READ, and reads beyond 255 bases with -g > 0; the
    reference has no such bounds (src/readaln.c:242-267, src/indelminer.c:934,948).  They take the general pass now
    (tests/test_gpu_realign_any.py holds the parity tests); nothing comes back IM_ST_UNSUPPORTED."""
    from indelminer_amd import capi
    contig = b"ACGT" * 500
    gpu_ctx.set_reference([contig])
    for kw, read in ((dict(numgaps=200), b"ACGTACGTACGTACGTACGT"), (dict(), b"A" * (capi.MAX_READ + 1)), (dict(numgaps=2), b"A" * 300)):
        rc, out = gpu_ctx.realign_batch(capi.params(**kw), [read], [0], [100], [300], allow=(capi.E_ABORT,))
        st, res = ob.realign(ob.params(**kw), contig, len(contig), 100, 300, read.decode())
        assert gpucmp.hip_vs_oracle(out[0], st, res) is None, kw
    gpu_ctx.expect_read_length(capi.MAX_READ + 1)
