"""Parity of the HIP split-read cluster kernels with the CPU oracle (bit-exact)."""
import os

import numpy as np
import pytest

from tests.test_oracle_golden import _cluster_oracle

pytestmark = pytest.mark.gpu


def _random_evidence(seed, n, span):
    rng = np.random.default_rng(seed)
    nsite = max(1, n // 12)
    site_b1 = rng.integers(0, span, nsite)
    site_len = rng.integers(0, 60, nsite)          # 0 -> insertion (b1 == b2)
    pick = rng.integers(0, nsite, n)
    b1 = site_b1[pick].astype(np.int32)
    b2 = (site_b1[pick] + site_len[pick]).astype(np.int32)
    noise = rng.random(n) < 0.2
    b1[noise] = rng.integers(0, span, noise.sum())
    b2[noise] = b1[noise] + rng.integers(0, 60, noise.sum())
    cls = (b2 > b1).astype(np.int32)
    return cls, b1, b2


@pytest.mark.parametrize("n,span,marker,tie", [
    (1, 100, 2**31 - 1, 0), (7, 50, 2**31 - 1, 1), (300, 2000, 2**31 - 1, 0), (5000, 100000, 60000, 0),
    (5000, 100000, 60000, 1), (70000, 5000000, 2**31 - 1, 0), (70000, 5000000, 2500000, 1), (2049, 300, 200, 0),
])
def test_cluster_matches_oracle(gpu_ctx, n, span, marker, tie):
    cls, b1, b2 = _random_evidence(n * 31 + tie, n, span)
    o_order, o_first, o_count, o_used, o_k = _cluster_oracle(cls, b1, b2, marker, tie)
    h_order, h_first, h_count, h_used, h_k = gpu_ctx.cluster_sr(cls, b1, b2, marker, tie)
    assert h_k == o_k
    assert np.array_equal(h_used, o_used)
    assert np.array_equal(h_first, o_first)
    assert np.array_equal(h_count, o_count)
    m = int(o_used.sum())
    assert np.array_equal(h_order[:m], o_order[:m])


def test_cluster_empty(gpu_ctx):
    order, first, count, used, k = gpu_ctx.cluster_sr([], [], [])
    assert k == 0 and len(order) == 0


def test_cluster_sortedness_and_grouping_large(gpu_ctx):
    """Size-independent properties at 2M records: clusters ascend in (b1,b2), members share
    a key and ascend in arrival, every record is placed exactly once."""
    n = 2_000_000
    cls, b1, b2 = _random_evidence(5, n, 50_000_000)
    order, first, count, used, k = gpu_ctx.cluster_sr(cls, b1, b2)
    assert used.all() and int(count.sum()) == n
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.int32))
    kb1, kb2 = b1[order], b2[order]
    key = kb1.astype(np.int64) << 32 | kb2.astype(np.int64)
    assert (np.diff(key) >= 0).all()
    same = np.diff(key) == 0
    assert (np.diff(order)[same] > 0).all()
    heads = np.zeros(n, bool); heads[first] = True
    assert np.array_equal(heads[1:], ~same)


@pytest.mark.parametrize("path", ["slots", "hist"])
def test_cluster_slots_matches_oracle(gpu_ctx, path):
    """Slot form (empty slots = cls -1) through im_dev_cluster_slots (one workgroup) and
    im_dev_cluster_hist (breakpoint histogram): same clusters as the oracle on the compacted
    records, order[] in slot indices; the histogram scratch must come back clean (3 rounds)."""
    import ctypes as C
    from indelminer_amd import capi
    L = capi.lib()
    for seed, n_slots, live_frac, marker, tie in [(1, 4000, 0.5, 2**31 - 1, 0), (2, 30000, 0.25, 2**31 - 1, 1),
                                                  (3, 9000, 0.9, 40000, 0), (4, 50, 0.5, 2**31 - 1, 0),
                                                  (5, 40000, 0.5, 2**31 - 1, 0)]:
        rng = np.random.default_rng(seed)
        cls, b1, b2 = _random_evidence(seed, n_slots, 80000)
        live = rng.random(n_slots) < live_frac
        cls_s = np.where(live, cls, -1).astype(np.int32)
        idx = np.nonzero(live)[0]
        o_order, o_first, o_count, o_used, o_k = _cluster_oracle(cls[idx], b1[idx], b2[idx], marker, tie)
        bufs = {}
        for name, arr in (("cls", cls_s), ("b1", b1), ("b2", b2)):
            bufs[name] = capi.DevBuf(gpu_ctx, 4 * n_slots).upload(arr)
        d_order = capi.DevBuf(gpu_ctx, 4 * n_slots); d_first = capi.DevBuf(gpu_ctx, 4 * n_slots)
        d_count = capi.DevBuf(gpu_ctx, 4 * n_slots); d_used = capi.DevBuf(gpu_ctx, n_slots); d_counts = capi.DevBuf(gpu_ctx, 64)
        if path == "slots":
            gpu_ctx._check(L.im_dev_cluster_slots(gpu_ctx.h, n_slots, bufs["cls"].ptr, bufs["b1"].ptr, bufs["b2"].ptr, marker, tie,
                                                  d_order.ptr, d_first.ptr, d_count.ptr, d_used.ptr, d_counts.ptr, gpu_ctx.stream))
            gpu_ctx._check(L.im_stream_sync(gpu_ctx.h, gpu_ctx.stream))
            counts = d_counts.download(np.int32, 2)
            if len(idx) > L.im_dev_cluster_slots_max():
                assert counts[0] == -1 and counts[1] == len(idx)
                continue
        else:
            hb = L.im_dev_cluster_hist_scratch_bytes(n_slots)
            d_hs = capi.DevBuf(gpu_ctx, hb)
            gpu_ctx._check(L.im_dev_cluster_hist_init(gpu_ctx.h, n_slots, d_hs.ptr, hb, gpu_ctx.stream))
            for _round in range(3):     # the table cleans itself: identical answers call after call
                gpu_ctx._check(L.im_dev_cluster_hist(gpu_ctx.h, n_slots, bufs["cls"].ptr, bufs["b1"].ptr, bufs["b2"].ptr, marker, tie,
                                                     d_order.ptr, d_first.ptr, d_count.ptr, d_used.ptr, d_counts.ptr,
                                                     d_hs.ptr, hb, gpu_ctx.stream))
            # ... and once more replayed from a captured launch graph (im_capture_* / im_graph_launch)
            with capi.Graph.capture(gpu_ctx) as g:
                gpu_ctx._check(L.im_dev_cluster_hist(gpu_ctx.h, n_slots, bufs["cls"].ptr, bufs["b1"].ptr, bufs["b2"].ptr, marker, tie,
                                                     d_order.ptr, d_first.ptr, d_count.ptr, d_used.ptr, d_counts.ptr,
                                                     d_hs.ptr, hb, gpu_ctx.stream))
            g.launch()
            gpu_ctx._check(L.im_stream_sync(gpu_ctx.h, gpu_ctx.stream))
            g.close()
            counts = d_counts.download(np.int32, 2)
            n_keys = len(np.unique(np.stack([cls[idx], b1[idx], b2[idx]]), axis=1).T)
            if n_keys > 8192:           # more distinct breakpoints than the table lists: refused, caller takes the radix path
                assert counts[0] == -1
                continue
        assert counts[1] == len(idx) and counts[0] == o_k
        m = int(o_used.sum())
        assert np.array_equal(d_order.download(np.int32, n_slots)[:m], idx[o_order[:m]])
        assert np.array_equal(d_first.download(np.int32, n_slots)[:o_k], o_first)
        assert np.array_equal(d_count.download(np.int32, n_slots)[:o_k], o_count)
        used = d_used.download(np.uint8, n_slots)
        want = np.zeros(n_slots, np.uint8); want[idx[o_used.astype(bool)]] = 1
        assert np.array_equal(used, want)


def test_cluster_records_and_rccl_allgather_world1(gpu_ctx):
    """The gathered unit (16 B cluster records) and the RCCL all-gather entry point with one rank."""
    import ctypes as C
    from indelminer_amd import capi
    from tests.support import gathered as shard
    L = capi.lib()
    n = 3000
    cls, b1, b2 = _random_evidence(77, n, 40000)
    o_order, o_first, o_count, o_used, o_k = _cluster_oracle(cls, b1, b2, 2**31 - 1, 0)
    d = {k: capi.DevBuf(gpu_ctx, 4 * n).upload(v) for k, v in (("cls", cls), ("b1", b1), ("b2", b2))}
    d_order = capi.DevBuf(gpu_ctx, 4 * n); d_first = capi.DevBuf(gpu_ctx, 4 * n); d_count = capi.DevBuf(gpu_ctx, 4 * n)
    d_counts = capi.DevBuf(gpu_ctx, 64)
    cap = 4096
    d_recs = capi.DevBuf(gpu_ctx, 16 * cap); d_all = capi.DevBuf(gpu_ctx, 16 * cap)
    st = gpu_ctx.stream
    gpu_ctx._check(L.im_dev_cluster_slots(gpu_ctx.h, n, d["cls"].ptr, d["b1"].ptr, d["b2"].ptr, 2**31 - 1, 0,
                                          d_order.ptr, d_first.ptr, d_count.ptr, None, d_counts.ptr, st))
    gpu_ctx._check(L.im_dev_cluster_records(gpu_ctx.h, 3, d_counts.ptr, d_order.ptr, d_first.ptr, d_count.ptr,
                                            d["cls"].ptr, d["b1"].ptr, d["b2"].ptr, d_recs.ptr, cap, st))
    comm = capi.Comm(gpu_ctx, capi.comm_unique_id(), 0, 1)
    comm.allgather(d_recs.ptr, d_all.ptr, 16 * cap, st)
    gpu_ctx._check(L.im_stream_sync(gpu_ctx.h, st))
    got, trunc = shard.merge_gathered(d_all.download(np.int32, 4 * cap), cap)
    heads = o_order[o_first]
    want = shard.pack_records(3, b1[heads], b2[heads], cls[heads], o_count, n, cap)
    want, _ = shard.merge_gathered(want.reshape(-1), cap)
    assert not trunc and np.array_equal(got, want)
    comm.close()


def test_depth_matches_numpy(gpu_ctx):
    """Region depth (DP=): difference array + device scan + range sums vs numpy, incl. segments
    clipped at both contig ends and a contig longer than one scan tile."""
    rng = np.random.default_rng(11)
    for clen, nseg in ((5000, 800), (100_000, 60_000), (3_000_000, 900_000)):
        start = rng.integers(-50, clen + 20, nseg).astype(np.int32)
        ln = rng.integers(1, 101, nseg).astype(np.int32)
        gpu_ctx.depth_build(clen, start, ln)
        diff = np.zeros(clen + 1, np.int64)
        a = np.clip(start.astype(np.int64), 0, clen); b = np.clip(start.astype(np.int64) + ln, 0, clen)
        ok = a < b
        np.add.at(diff, a[ok], 1); np.add.at(diff, b[ok], -1)
        depth = np.cumsum(diff)[:clen]
        csum = np.concatenate([[0], np.cumsum(depth)])
        qb = rng.integers(0, clen - 1, 500).astype(np.int32)
        qe = np.minimum(qb + rng.integers(1, 1200, 500), clen).astype(np.int32)
        got = gpu_ctx.depth_query(qb, qe)
        want = (csum[qe] - csum[qb]).astype(np.uint32)
        assert np.array_equal(got, want)


def test_support_sw_matches_oracle(gpu_ctx):
    """Annotate-mode SW (K7): forward-carried path statistics vs the oracle's full-matrix DP +
    traceback on random target/query pairs (planted indels, substitutions, lengths up to 255 x 1500)."""
    import ctypes as C
    from tests.support import oraclebind as ob
    L = ob.lib()
    rng = np.random.default_rng(5)
    targets, queries = [], []
    for it in range(460):
        len1 = int(rng.choice([40, 120, 200, 260, 700, 1500]))
        len2 = int(rng.choice([20, 64, 65, 100, 100, 128, 150, 255]))
        if it >= 400:                  # reads of up to IM_MAX_READ bases: the packed path statistics (11 + 11 + 10 bits) at their widest
            len1, len2 = int(rng.choice([1200, 2500, 4000])), int(rng.choice([256, 300, 600, 1020, 1020]))
        t = rng.choice(list(b"ACGT"), size=len1).astype(np.uint8)
        p = int(rng.integers(0, max(1, len1 - len2 // 2)))
        q = t[p:p + len2].copy()
        if len(q) < len2:
            q = np.concatenate([q, rng.choice(list(b"ACGT"), size=len2 - len(q)).astype(np.uint8)])
        typ = rng.random()
        if typ < 0.35 and len2 > 20:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 12))
            q = np.concatenate([q[:cut], q[cut + d:], rng.choice(list(b"ACGT"), size=d).astype(np.uint8)])
        elif typ < 0.7 and len2 > 20:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 12))
            q = np.concatenate([q[:cut], rng.choice(list(b"ACGT"), size=d).astype(np.uint8), q[cut:]])[:len2]
        elif typ < 0.8:
            q = rng.choice(list(b"ACGT"), size=len2).astype(np.uint8)
        sub = rng.random(len(q)) < rng.choice([0, 0.02, 0.1, 0.3] if it >= 400 else [0, 0.02, 0.1])
        q[sub] = rng.choice(list(b"ACGT"), size=int(sub.sum())).astype(np.uint8)
        if it % 17 == 0:
            q[len(q) // 2] = ord("N")
        if it % 23 == 0:
            t[len(t) // 3] = ord("N")
        targets.append(t.tobytes()); queries.append(q.tobytes())
    targets.append(b"ACGTACGT"); queries.append(b"TTTT")        # nothing positive... (a T matches)
    targets.append(b"AAAA"); queries.append(b"CCCC")            # all-nonpositive matrix
    got = gpu_ctx.support_batch(targets, queries)
    for k, (t, q) in enumerate(zip(targets, queries)):
        s, i, a = C.c_int32(), C.c_int32(), C.c_int32()
        L.imo_sw_indel(t, len(t), q, len(q), C.byref(s), C.byref(i), C.byref(a))
        assert tuple(got[k][:3]) == (s.value, i.value, a.value), (k, len(t), len(q), tuple(got[k]), (s.value, i.value, a.value))


def test_support_sw_long_windows_and_queries(gpu_ctx):
    """Windows beyond IM_MAX_SW_TARGET and queries beyond IM_MAX_READ (the reference takes any, src/variant.c:1246-1424): the
    support kernel's second form -- boundary row in device memory, 64-bit packed statistics -- in one batch with tasks of the LDS form"""
    import ctypes as C
    from indelminer_amd import capi
    from tests.support import oraclebind as ob
    L = ob.lib()
    rng = np.random.default_rng(77)
    targets, queries = [], []
    shapes = [(4096, 100), (5100, 100), (13000, 150), (9000, 300), (3000, 1021), (6000, 1500), (2000, 2600), (200, 100), (1500, 255), (4095, 1020), (20000, 64)]
    for it in range(40):
        len1, len2 = shapes[it % len(shapes)]
        t = rng.choice(list(b"ACGT"), size=len1).astype(np.uint8)
        p = int(rng.integers(0, max(1, len1 - len2)))
        q = t[p:p + len2].copy()
        if len(q) < len2:
            q = np.concatenate([q, rng.choice(list(b"ACGT"), size=len2 - len(q)).astype(np.uint8)])
        if it % 3 == 0:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 30))
            q = np.concatenate([q[:cut], q[cut + d:], rng.choice(list(b"ACGT"), size=d).astype(np.uint8)])
        elif it % 3 == 1:
            cut = int(rng.integers(5, len2 - 5)); d = int(rng.integers(1, 30))
            q = np.concatenate([q[:cut], rng.choice(list(b"ACGT"), size=d).astype(np.uint8), q[cut:]])[:len2]
        sub = rng.random(len(q)) < rng.choice([0, 0.02, 0.1])
        q[sub] = rng.choice(list(b"ACGT"), size=int(sub.sum())).astype(np.uint8)
        targets.append(t.tobytes()); queries.append(q.tobytes())
    got = gpu_ctx.support_batch(targets, queries)
    n_big = 0
    for k, (t, q) in enumerate(zip(targets, queries)):
        s, i, a = C.c_int32(), C.c_int32(), C.c_int32()
        L.imo_sw_indel(t, len(t), q, len(q), C.byref(s), C.byref(i), C.byref(a))
        assert tuple(got[k][:3]) == (s.value, i.value, a.value) and got[k][3] == capi.ST_EVIDENCE, (k, len(t), len(q), tuple(got[k]), (s.value, i.value, a.value))
        n_big += len(t) > 4095 or len(q) > capi.MAX_READ
    assert n_big >= 20


def test_cluster_kernels_match_reference_process_evidence(gpu_ctx):
    """the device cluster path against vectors the reference's own process_evidence produced (tests/golden/units_cluster.json)"""
    import json
    from tests.test_oracle_units import reference_clusters
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "units_cluster.json")))["cases"]
    for case in cases:
        cls, b1, b2 = case["cls"], case["b1"], case["b2"]
        for tie in (0, 1):
            order, first, count, used, k = gpu_ctx.cluster_sr(cls, b1, b2, case["marker"], tie)
            got = []
            for f, c in zip(first, count):
                m = [int(x) for x in order[f:f + c]]
                got.append(((cls[m[0]], b1[m[0]], b2[m[0]]), m))
            want = reference_clusters(case)
            assert got == (want if tie == 0 else [(key, m[::-1]) for key, m in want]), case["seed"]
            assert [int(x) for x in used] == case["used"]


def test_support_kernel_matches_reference_realign_with_indel(gpu_ctx):
    """the annotate-mode Smith-Waterman against vectors the reference's own realign_with_indel produced"""
    import json
    from tests.test_oracle_units import variant_window
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "units_sw.json")))["cases"]
    targets = [variant_window(c) for c in cases]
    queries = [c["read"][c["qstart"]:c["qstop"]].encode() for c in cases]
    got = gpu_ctx.support_batch(targets, queries)
    for k, c in enumerate(cases):
        assert [int(x) for x in got[k][:3]] == c["expect"], (k, c["expect"], tuple(got[k]))


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 50001])
def test_compact_results_packs_exactly_the_records_with_evidence(gpu_ctx, n):
    """im_dev_compact_results: status and place of every read, the records with status == IM_ST_EVIDENCE and n_ev > 0 packed whole
    (any order), nothing else; also with the batch size read from device memory"""
    import ctypes as C
    from indelminer_amd import capi
    L = capi.lib()
    rng = np.random.default_rng(n)
    res = np.frombuffer(rng.integers(0, 256, size=512 * max(n, 1), dtype=np.uint8).tobytes(), dtype=capi.RESULT_DTYPE).copy()
    res["status"] = rng.choice([0, 1, 1, -1, -2, -3], size=len(res))
    res["n_ev"] = rng.choice([0, 1, 2, 4], size=len(res))
    for n_live in ([n] if n < 100 else [n, n - 37]):
        d_res = capi.DevBuf(gpu_ctx, 512 * max(n, 1)).upload(res)
        d_stat, d_slot = capi.DevBuf(gpu_ctx, 4 * max(n, 1)), capi.DevBuf(gpu_ctx, 4 * max(n, 1))
        d_comp, d_cnt = capi.DevBuf(gpu_ctx, 512 * max(n, 1)), capi.DevBuf(gpu_ctx, 256)
        d_n = capi.DevBuf(gpu_ctx, 256).upload(np.array([n_live], np.int32))
        gpu_ctx._check(L.im_dev_compact_results(gpu_ctx.h, d_res.ptr, n, d_n.ptr if n_live != n else None, d_stat.ptr, d_slot.ptr, d_comp.ptr, d_cnt.ptr, gpu_ctx.stream))
        gpu_ctx._check(L.im_stream_sync(gpu_ctx.h, gpu_ctx.stream))
        cnt = int(d_cnt.download(np.int32, 1)[0])
        want = (res["status"][:n_live] == 1) & (res["n_ev"][:n_live] > 0)
        assert cnt == int(want.sum())
        if n_live:
            stat, slot = d_stat.download(np.int32, n_live), d_slot.download(np.int32, n_live)
            assert np.array_equal(stat, res["status"][:n_live])
            assert np.array_equal(slot >= 0, want)
            assert sorted(slot[want].tolist()) == list(range(cnt))
            if cnt:
                comp = d_comp.download(capi.RESULT_DTYPE, cnt)
                assert comp[slot[want]].tobytes() == res[:n_live][want].tobytes()
        for b in (d_res, d_stat, d_slot, d_comp, d_cnt, d_n):
            b.free()
