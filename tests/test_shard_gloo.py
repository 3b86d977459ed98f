"""N > 1 host logic on CPU: two gloo ranks own disjoint contigs, cluster their own evidence
(CPU oracle stands in for the device here -- this test is about sharding and the merge of
gathered cluster lists, not about the kernels), all-gather fixed-capacity record buffers and
must reproduce the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _evidence_for_contig(tid):
    rng = np.random.default_rng(100 + tid)
    n = 400 + 50 * tid
    nsite = 40
    site = rng.integers(0, 50000, nsite)
    ln = rng.integers(0, 40, nsite)
    pick = rng.integers(0, nsite, n)
    b1 = site[pick].astype(np.int32)
    b2 = (site[pick] + ln[pick]).astype(np.int32)
    cls = (b2 > b1).astype(np.int32)
    return cls, b1, b2


def _clusters(tid, cap):
    from indelminer_amd import shard
    from tests.test_oracle_golden import _cluster_oracle
    cls, b1, b2 = _evidence_for_contig(tid)
    order, first, count, used, k = _cluster_oracle(cls, b1, b2, 2**31 - 1, 0)
    heads = order[first]
    return shard.pack_records(tid, b1[heads], b2[heads], cls[heads], count, len(cls), cap)


def _worker(rank, world, port, n_contigs, cap, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from indelminer_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.contigs_of_rank(n_contigs, rank, world)
    per_rank = (n_contigs + world - 1) // world
    bufs = np.zeros((per_rank, cap, 4), np.int32)
    for j, t in enumerate(mine):
        bufs[j] = _clusters(t, cap)
    send = torch.from_numpy(bufs.reshape(-1))
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    gathered = np.concatenate([r.numpy() for r in recv])
    recs, trunc = shard.merge_gathered(gathered, cap)
    if rank == 0:
        q.put((recs, trunc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_merge_equals_single_process():
    import torch.multiprocessing as mp
    from indelminer_amd import shard
    n_contigs, cap, world = 5, 256, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_contigs, cap, q)) for r in range(world)]
    for p in procs:
        p.start()
    recs, trunc = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = np.concatenate([_clusters(t, cap)[None] for t in range(n_contigs)])
    want, _ = shard.merge_gathered(single.reshape(-1, 4), cap)
    assert not trunc
    assert np.array_equal(recs, want)
    # ordered by (tid, b1, b2), every contig present, supports add up to the evidence counts
    assert (np.diff(recs[:, 0]) >= 0).all()
    for t in range(n_contigs):
        sup = recs[recs[:, 0] == t][:, 3] & 0xFFFFFF
        assert int(sup.sum()) == 400 + 50 * t


def test_contig_ownership_is_a_partition():
    from indelminer_amd import shard
    for world in (1, 2, 4, 8):
        owned = [t for r in range(world) for t in shard.contigs_of_rank(24, r, world)]
        assert sorted(owned) == list(range(24))
        assert all(shard.owner_of_contig(t, world) == t % world for t in range(24))
