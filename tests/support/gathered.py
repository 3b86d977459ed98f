"""Host-side twin of im_dev_cluster_records and of the merge of all-gathered record buffers (test helper for
tests/test_gpu_cluster.py::test_cluster_records_and_rccl_allgather_world1; the product's multi-rank path is
indelminer_amd/host/host_multirank.c, covered by tests/test_multi_rank_driver.py)."""
import numpy as np


def pack_records(tid, keys_b1, keys_b2, cls, support, n_live, cap):
    """Host-side twin of im_dev_cluster_records: int32 [cap,4] buffer."""
    recs = np.zeros((cap, 4), dtype=np.int32)
    n = len(keys_b1)
    lim = min(n, cap - 1)
    recs[0] = (n, n_live, tid, 1 if n > cap - 1 else 0)
    recs[1:1 + lim, 0] = tid
    recs[1:1 + lim, 1] = keys_b1[:lim]
    recs[1:1 + lim, 2] = keys_b2[:lim]
    recs[1:1 + lim, 3] = (np.asarray(cls[:lim], dtype=np.int32) << 24) | (np.asarray(support[:lim], dtype=np.int32) & 0xFFFFFF)
    return recs


def merge_gathered(gathered, cap):
    """gathered: int32 array [world * n_bufs_per_rank * cap, 4] as the all-gather leaves it.
    Returns (records [m,4] ordered by (tid,b1,b2), truncated flag)."""
    g = np.asarray(gathered, dtype=np.int32).reshape(-1, cap, 4)
    out = []
    truncated = False
    for buf in g:
        n = int(buf[0, 0])
        if n < 0:
            raise ValueError("a shard reported a cluster overflow")
        truncated |= bool(buf[0, 3])
        out.append(buf[1:1 + min(n, cap - 1)])
    recs = np.concatenate(out) if out else np.zeros((0, 4), np.int32)
    order = np.lexsort((recs[:, 2], recs[:, 1], recs[:, 0]))
    return recs[order], truncated
