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


def test_host_code_lengths_vs_reference():
    """The product's length-limited code lengths (host/huffman.cc) == ZopfliLengthLimitedCodeLengths
    (katajainen.c:172) on random histograms of the three alphabets: ties, long tails that hit the length
    limit, sparse and dense counts."""
    import ctypes
    host = ol.hosttest_library()
    host.zamd_test_code_lengths.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_uint)]
    rng = np.random.default_rng(5)
    for it in range(1500):
        n, maxbits = ((288, 15), (32, 15), (19, 7))[it % 3]
        style = (it // 3) % 5
        if style == 0:
            f = rng.integers(0, 1000, n)
        elif style == 1:
            f = rng.integers(0, 3, n)
        elif style == 2:
            # (below 2^22: the reference's comparator returns a difference of keys = weight << 9 as an int,
            # katajainen.c:146-148, so its order is only defined while counts differ by less than that —
            # always, inside the 1 MB master blocks of ZopfliDeflate)
            f = 1 << rng.integers(0, 22, n)
        elif style == 3:
            f = np.where(rng.random(n) < 0.7, 0, rng.integers(1, 100000, n))
        else:
            f = np.where(np.arange(n) < rng.integers(0, n + 1), rng.integers(1, 8, n), 0)
        rc, want = ol.ref_code_lengths(f, maxbits)
        arr = (ctypes.c_size_t * n)(*[int(x) for x in f])
        out = (ctypes.c_uint * n)()
        got_rc = host.zamd_test_code_lengths(arr, n, maxbits, out)
        assert (got_rc != 0) == (rc != 0), (it, f)
        if rc == 0:
            assert list(out) == want, (it, list(f))


@pytest.mark.parametrize("cls,n", [("T", 60000), ("X", 40000), ("M", 90000), ("Z", 30000), ("R", 8000), ("B", 20000)])
def test_host_block_size_and_split_vs_reference(cls, n):
    """CalculateBlockSizeAutoType (deflate.c:610; host/block_cost.cc) over random symbol ranges of a greedy
    parse, and the split points of ZopfliBlockSplitLZ77 (blocksplitter.c:215; host/block_split.cc)."""
    import ctypes
    host = ol.hosttest_library()
    u16p, sz = ctypes.POINTER(ctypes.c_uint16), ctypes.c_size_t
    host.zamd_test_block_size_auto.argtypes = [u16p, u16p, sz, sz, sz]
    host.zamd_test_block_size_auto.restype = ctypes.c_double
    host.zamd_test_block_split.argtypes = [u16p, u16p, sz, sz, ctypes.POINTER(sz), sz]
    host.zamd_test_block_split.restype = sz
    data = generate(cls, n)
    ll, dd = ol.OracleTable(data, 0, n).greedy()
    ll, dd = np.ascontiguousarray(ll, dtype=np.uint16), np.ascontiguousarray(dd, dtype=np.uint16)
    pl, pd = ll.ctypes.data_as(u16p), dd.ctypes.data_as(u16p)
    r = ol.RefSymbols(ll, dd)
    try:
        rng = np.random.default_rng(11)
        m = len(ll)
        ranges = [(0, m), (0, 1), (m - 1, m), (0, min(m, 300)), (0, min(m, 700))]
        ranges += [tuple(sorted(rng.integers(0, m + 1, 2).tolist())) for _ in range(60)]
        for a, b in ranges:
            if a == b:
                continue
            assert host.zamd_test_block_size_auto(pl, pd, m, a, b) == r.block_size_auto(a, b), (a, b)
        for maxblocks in (15, 4, 0):
            pts = (sz * 64)()
            k = host.zamd_test_block_split(pl, pd, m, maxblocks, pts, 64)
            assert [pts[i] for i in range(k)] == r.block_split(maxblocks), maxblocks
        # the round-by-round search of several sequences at once (BlockSplitLz77Batch: blocks searched before their
        # turn, decisions in the reference's order): the sequence cut at three lengths, each against the reference
        host.zamd_test_block_split_batch.argtypes = [u16p, u16p, sz, sz, sz, ctypes.POINTER(sz), ctypes.POINTER(sz), sz]
        host.zamd_test_block_split_batch.restype = None
        copies = 3
        refs = []
        for v in range(copies):
            mv = m - v * (m // 7)
            rv = ol.RefSymbols(ll[:mv], dd[:mv])
            refs.append([rv.block_split(mb) for mb in (15, 4, 0)])
            rv.close()
        for j, maxblocks in enumerate((15, 4, 0)):
            pts = (sz * (64 * copies))()
            cnt = (sz * copies)()
            host.zamd_test_block_split_batch(pl, pd, m, copies, maxblocks, pts, cnt, 64)
            for v in range(copies):
                assert [pts[v * 64 + i] for i in range(cnt[v])] == refs[v][j], (maxblocks, v)
    finally:
        r.close()


def test_exported_block_size_functions_vs_reference():
    """ZopfliCalculateBlockSize / ZopfliCalculateBlockSizeAutoType as libzopfli_amd.so exports them (deflate.h:79,85), on
    the REFERENCE's own ZopfliLZ77Store (filled by the reference's ZopfliStoreLitLenDist): the same doubles as the
    reference's functions for every block type, on ranges below and above the 1000-symbol switch of deflate.c:614."""
    import ctypes
    from zopfli_amd import api
    mine = api.library()
    r = ol.ref()
    sz = ctypes.c_size_t
    for lib in (mine, r):
        lib.ZopfliCalculateBlockSize.argtypes = [ctypes.c_void_p, sz, sz, ctypes.c_int]
        lib.ZopfliCalculateBlockSize.restype = ctypes.c_double
        lib.ZopfliCalculateBlockSizeAutoType.argtypes = [ctypes.c_void_p, sz, sz]
        lib.ZopfliCalculateBlockSizeAutoType.restype = ctypes.c_double
    rng = np.random.default_rng(3)
    for cls, n in (("T", 30000), ("R", 3000), ("Z", 20000), ("T", 2500)):
        data = generate(cls, n)
        ll, dd = ol.OracleTable(data, 0, n).greedy()
        rs = ol.RefSymbols(np.ascontiguousarray(ll, dtype=np.uint16), np.ascontiguousarray(dd, dtype=np.uint16))
        try:
            m = len(ll)
            p = ctypes.addressof(rs.store)
            ranges = [(0, m), (0, 1), (m - 1, m), (0, min(m, 900)), (5, 5)] + [tuple(sorted(rng.integers(0, m + 1, 2).tolist())) for _ in range(40)]
            for a, b in ranges:
                for btype in (0, 1, 2):
                    assert mine.ZopfliCalculateBlockSize(p, a, b, btype) == r.ZopfliCalculateBlockSize(p, a, b, btype), (cls, a, b, btype)
                if a != b:
                    assert mine.ZopfliCalculateBlockSizeAutoType(p, a, b) == r.ZopfliCalculateBlockSizeAutoType(p, a, b), (cls, a, b)
        finally:
            rs.close()
