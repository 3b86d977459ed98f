"""Live check of the CPU oracle against the real reference compiled in place
(oracle/_ref/libimref.so).  Only runs where that library exists (the build
container, or a GPU box that received the prebuilt file); the committed golden
vectors cover the same ground everywhere else."""
import ctypes as C
import random

import pytest

from tests.support import compare, oraclebind as ob, refbind

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref not built (make -C oracle ref)")


@pytest.mark.parametrize("k,g,seed", [(6, 0, 11), (6, 1, 12), (8, 2, 13), (5, 0, 14),
                                      (2, 0, 21), (3, 0, 22), (4, 0, 23), (7, 0, 24), (10, 0, 25), (11, 0, 26),
                                      (6, 3, 31), (6, 5, 32), (6, 8, 33), (6, 12, 34), (11, 4, 35)])
# k stops at 11 here: the reference allocates and clears two 4^k-entry tables per band search
# (src/alignment.c:38-39), 128 MB per call at k = 12 and 8 GB at k = 15
def test_fuzz_against_reference(k, g, seed):
    rng = random.Random(seed)
    R = refbind.Ref()
    maxdel = rng.choice([1000, 200])
    eth = max(k, 10)
    R.set_params(k, g, maxdel, eth)
    P = ob.params(k, g, maxdel, eth)
    contig = "".join(rng.choice("ACGT") for _ in range(6000))
    cb = contig.encode()
    buf = C.create_string_buffer(cb)
    n_ev = 0
    for _ in range(150):
        L = rng.choice([100, 100, 76, 150])
        anchor = rng.randint(0, len(contig) - 1)
        p = max(0, min(len(contig) - L - 60, anchor + rng.randint(-700, 700)))
        d = rng.choice([1, 3, 10, 50, 300])
        cut = rng.randint(5, L - 5)
        if rng.random() < 0.6:
            read = contig[p:p + cut] + contig[p + cut + d:p + cut + d + (L - cut)]
        else:
            ins = "".join(rng.choice("ACGT") for _ in range(min(d, 30)))
            read = (contig[p:p + cut] + ins + contig[p + cut:p + L])[:L]
        if rng.random() < 0.3:
            i = rng.randrange(len(read))
            read = read[:i] + rng.choice("ACGT") + read[i + 1:]
        st, res = ob.realign(P, cb, len(cb), anchor, 500, read)
        if st == -1:             # the reference would exit(1) inside this process (forceassert): nothing to compare
            continue
        ro = R.realign(buf, anchor, 500, read)
        msg = compare.ref_vs_oracle(ro, st, res, read)
        assert msg is None, msg
        n_ev += 0 if ro is None else len(ro)
    assert n_ev > 5 or k < 4          # two- and three-base seeds are never unique in a 100-base read: no band, no evidence


@pytest.mark.parametrize("k,g,seed", [(6, 0, 1), (6, 0, 2), (6, 1, 3), (6, 2, 4), (6, 5, 5), (8, 3, 6), (4, 0, 7), (6, 12, 8)])
def test_left_edge_bands_against_reference(k, g, seed):
    """Pins the left-edge behaviour of local_align's reverse pass (tests/support/leftedge.py): bands without a single
    k-mer vote that hang off the window's left edge, for -g 0 and -g > 0 (where endj - startj = -g != 0 lets the
    result reach ALIGN).  The oracle -- and through tests/test_gpu_realign.py the HIP kernels -- must give exactly
    what the reference compiled from its own sources gives, byte in front of the window included."""
    from tests.support import leftedge
    R = refbind.Ref()
    R.set_params(k, g, 1000, 10)
    P = ob.params(k, g, 1000, 10)
    contig, cases = leftedge.cases(seed)
    cb = contig.encode()
    # one byte in front of the contig that never equals a base: what the reference reads for a window at the contig's start
    raw = C.create_string_buffer(b"#" + cb)
    buf = C.cast(C.addressof(raw) + 1, C.POINTER(C.c_char * (len(cb) + 1))).contents
    seen = 0
    for c in cases:
        st, res = ob.realign(P, cb, len(cb), c["anchor"], c["range_max"], c["read"])
        if st == -1:
            continue
        ro = R.realign(buf, c["anchor"], c["range_max"], c["read"])
        msg = compare.ref_vs_oracle(ro, st, res, c["read"])
        assert msg is None, (c, msg)
        seen += 1
    assert seen > 60
