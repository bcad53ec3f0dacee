"""Pins the CPU oracle (oracle/zopfli_oracle.c) to the REAL reference: the functions of
oracle/_ref/libzopfli_ref.so (compiled from /root/reference by oracle/Makefile) are called
directly on the same inputs.  CPU only."""
import numpy as np
import pytest

import oracle_lib as ol
from zopfli_amd import generate

pytestmark = pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built (needs /root/reference)")

CASES = [("T", 12000, 0), ("X", 9000, 0), ("Z", 20000, 0), ("B", 8000, 0), ("R", 6000, 0), ("P", 9000, 0),
         ("M", 50000, 36000), ("Z", 45000, 33500)]


@pytest.mark.parametrize("cls,n,instart", CASES)
def test_match_table(cls, n, instart):
    """zo_find_longest_match == ZopfliFindLongestMatch(limit 258, sublen) at every position (lz77.c:407)."""
    data = generate(cls, n)
    t = ol.OracleTable(data, instart, n)
    for i, (l, d, sub) in enumerate(ol.ref_match_table(data, instart, n)):
        ol_, od, osub = t.find_longest_match(instart + i)
        if l >= 3:
            assert (ol_, od) == (l, d), f"pos {instart + i}"
            assert list(osub[3:l + 1]) == sub, f"sublen at pos {instart + i}"
        else:
            assert ol_ < 3, f"pos {instart + i}"


@pytest.mark.parametrize("cls,n,instart", CASES)
def test_greedy_and_fixed(cls, n, instart):
    """zo_greedy == ZopfliLZ77Greedy (lz77.c:544); DP+trace+follow with the fixed-tree costs ==
    ZopfliLZ77OptimalFixed (squeeze.c:528)."""
    data = generate(cls, n)
    t = ol.OracleTable(data, instart, n)
    ll, dd = t.greedy()
    rl, rd = ol.ref_greedy(data, instart, n)
    assert np.array_equal(ll, rl) and np.array_equal(dd, rd)
    fll = np.array([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, dtype=np.float64)
    fd = np.full(32, 5.0)
    _, sl, sd = t.squeeze_run(fll, fd, ol.model_min_cost(fll, fd))
    rl, rd = ol.ref_optimal_fixed(data, instart, n)
    assert np.array_equal(sl, rl) and np.array_equal(sd, rd)
