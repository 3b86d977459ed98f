"""Parity of the HIP split-read cluster kernels with the CPU oracle (bit-exact)."""
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
