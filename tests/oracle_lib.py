"""Test-side bindings: the CPU oracle (oracle/libzopfli_oracle.so), the real
reference (oracle/_ref/libzopfli_ref.so) and the CPU-only host test library.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libzopfli_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libzopfli_ref.so")
HOSTTEST_SO = os.path.join(ROOT, "tests", "_build", "libzopfli_hosttest.so")

_u16p = ctypes.POINTER(ctypes.c_uint16)
_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libzopfli_oracle.so"])
        lib = ctypes.CDLL(ORACLE_SO)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        lib.zo_table_build.argtypes = [ctypes.c_char_p, sz, sz]
        lib.zo_table_build.restype = vp
        lib.zo_table_free.argtypes = [vp]
        lib.zo_table_free.restype = None
        lib.zo_find_longest_match.argtypes = [vp, sz, _u16p, _u16p, _u16p]
        lib.zo_find_longest_match.restype = None
        for n in ("zo_same", "zo_prev1", "zo_prev2"):
            getattr(lib, n).argtypes = [vp, sz]
            getattr(lib, n).restype = ctypes.c_uint16
        lib.zo_greedy.argtypes = [vp, _u16p, _u16p]
        lib.zo_greedy.restype = sz
        lib.zo_get_best_lengths.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                            ctypes.c_double, _u16p]
        lib.zo_get_best_lengths.restype = ctypes.c_double
        lib.zo_trace_follow.argtypes = [vp, _u16p, _u16p, _u16p]
        lib.zo_trace_follow.restype = sz
        lib.zo_histogram.argtypes = [_u16p, _u16p, sz, ctypes.POINTER(ctypes.c_uint32)]
        lib.zo_histogram.restype = None
        _oracle = lib
    return _oracle


class OracleTable:
    """zo_table for block [instart, inend) of `data`."""

    def __init__(self, data, instart, inend):
        self.lib = oracle()
        self.data = data  # keep alive
        self.instart, self.inend = instart, inend
        self.h = self.lib.zo_table_build(data, instart, inend)

    def close(self):
        if self.h:
            self.lib.zo_table_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def find_longest_match(self, pos):
        sub = np.zeros(259, dtype=np.uint16)
        d, l = ctypes.c_uint16(0), ctypes.c_uint16(0)
        self.lib.zo_find_longest_match(self.h, pos, sub.ctypes.data_as(_u16p), ctypes.byref(d), ctypes.byref(l))
        return l.value, d.value, sub

    def hash_links(self):
        """same[], prev1[], prev2[] for positions windowstart .. inend - 1 (zo_same / zo_prev1 / zo_prev2)."""
        ws = max(0, self.instart - 32768)
        out = [np.zeros(self.inend - ws, dtype=np.uint16) for _ in range(3)]
        for a, f in zip(out, (self.lib.zo_same, self.lib.zo_prev1, self.lib.zo_prev2)):
            for i, p in enumerate(range(ws, self.inend)):
                a[i] = f(self.h, p)
        return out

    def greedy(self):
        B = self.inend - self.instart
        ll = np.zeros(B + 1, dtype=np.uint16)
        dd = np.zeros(B + 1, dtype=np.uint16)
        n = self.lib.zo_greedy(self.h, ll.ctypes.data_as(_u16p), dd.ctypes.data_as(_u16p))
        return ll[:n].copy(), dd[:n].copy()

    def squeeze_run(self, ll_cost, d_cost, mincost):
        B = self.inend - self.instart
        ll_cost = np.ascontiguousarray(ll_cost, dtype=np.float64)
        d_cost = np.ascontiguousarray(d_cost, dtype=np.float64)
        la = np.zeros(B + 1, dtype=np.uint16)
        self.lib.zo_get_best_lengths(self.h, ll_cost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                     d_cost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), float(mincost),
                                     la.ctypes.data_as(_u16p))
        ll = np.zeros(B + 1, dtype=np.uint16)
        dd = np.zeros(B + 1, dtype=np.uint16)
        n = self.lib.zo_trace_follow(self.h, la.ctypes.data_as(_u16p), ll.ctypes.data_as(_u16p),
                                     dd.ctypes.data_as(_u16p))
        return la, ll[:n].copy(), dd[:n].copy()


def histogram(litlens, dists):
    lib = oracle()
    litlens = np.ascontiguousarray(litlens, dtype=np.uint16)
    dists = np.ascontiguousarray(dists, dtype=np.uint16)
    h = np.zeros(320, dtype=np.uint32)
    lib.zo_histogram(litlens.ctypes.data_as(_u16p), dists.ctypes.data_as(_u16p), len(litlens),
                     h.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
    return h


# ---- cost-model helpers restating squeeze.c (test side only) -----------------
_LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195,
             227, 258]


def length_symbol(l):
    s = 28
    while _LEN_BASE[s] > l:
        s -= 1
    return 257 + s


def length_extra_bits(l):
    s = length_symbol(l) - 257
    return 0 if (s < 8 or s == 28) else (s - 4) // 4


def dist_symbol(d):
    if d < 5:
        return d - 1
    l = (d - 1).bit_length() - 1
    return 2 * l + (((d - 1) >> (l - 1)) & 1)


def dist_extra_bits(d):
    return 0 if d < 5 else (d - 1).bit_length() - 2


def entropy_costs(hist320):
    """CalculateStatistics (squeeze.c:392) on a 320-bin histogram with the end symbol set."""
    import math
    kInvLog2 = 1.4426950408889

    def ent(counts):
        s = int(sum(int(c) for c in counts)) & 0xffffffff
        log2sum = (math.log(len(counts)) if s == 0 else math.log(s)) * kInvLog2
        out = []
        for c in counts:
            b = log2sum if c == 0 else log2sum - math.log(int(c)) * kInvLog2
            if -1e-5 < b < 0:
                b = 0.0
            out.append(b)
        return np.array(out, dtype=np.float64)

    ll = [int(x) for x in hist320[:288]]
    ll[256] = 1
    return ent(ll), ent([int(x) for x in hist320[288:]])


def model_min_cost(ll, d):
    """GetCostModelMinCost (squeeze.c:163)."""
    first = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]

    def cost(l, dist):
        return (length_extra_bits(l) + dist_extra_bits(dist)) + ll[length_symbol(l)] + d[dist_symbol(dist)]

    best_l, m = 0, 1e30
    for i in range(3, 259):
        c = cost(i, 1)
        if c < m:
            best_l, m = i, c
    best_d, m = 0, 1e30
    for dist in first:
        c = cost(3, dist)
        if c < m:
            best_d, m = dist, c
    return cost(best_l, best_d)


# ---- the real reference ------------------------------------------------------
class RefOptions(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("verbose", "verbose_more", "numiterations", "blocksplitting",
                                            "blocksplittinglast", "blocksplittingmax")]


class RefHash(ctypes.Structure):  # hash.h:29-47
    _fields_ = [("head", ctypes.POINTER(ctypes.c_int)), ("prev", _u16p), ("hashval", ctypes.POINTER(ctypes.c_int)),
                ("val", ctypes.c_int), ("head2", ctypes.POINTER(ctypes.c_int)), ("prev2", _u16p),
                ("hashval2", ctypes.POINTER(ctypes.c_int)), ("val2", ctypes.c_int), ("same", _u16p)]


class RefBlockState(ctypes.Structure):  # lz77.h:86-97
    _fields_ = [("options", ctypes.POINTER(RefOptions)), ("lmc", ctypes.c_void_p), ("blockstart", ctypes.c_size_t),
                ("blockend", ctypes.c_size_t)]


class RefStore(ctypes.Structure):  # lz77.h:44-62
    _fields_ = [("litlens", _u16p), ("dists", _u16p), ("size", ctypes.c_size_t), ("data", ctypes.c_char_p),
                ("pos", ctypes.POINTER(ctypes.c_size_t)), ("ll_symbol", _u16p), ("d_symbol", _u16p),
                ("ll_counts", ctypes.POINTER(ctypes.c_size_t)), ("d_counts", ctypes.POINTER(ctypes.c_size_t))]


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = ctypes.CDLL(REF_SO)
        sz = ctypes.c_size_t
        lib.ZopfliCompress.argtypes = [ctypes.POINTER(RefOptions), ctypes.c_int, ctypes.c_char_p, sz,
                                       ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte)), ctypes.POINTER(sz)]
        lib.ZopfliCompress.restype = None
        lib.ZopfliDeflatePart.argtypes = [ctypes.POINTER(RefOptions), ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz,
                                          sz, ctypes.POINTER(ctypes.c_ubyte),
                                          ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte)), ctypes.POINTER(sz)]
        lib.ZopfliDeflatePart.restype = None
        lib.ZopfliDeflate.argtypes = [ctypes.POINTER(RefOptions), ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz,
                                      ctypes.POINTER(ctypes.c_ubyte),
                                      ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte)), ctypes.POINTER(sz)]
        lib.ZopfliDeflate.restype = None
        lib.ZopfliAllocHash.argtypes = [sz, ctypes.POINTER(RefHash)]
        lib.ZopfliResetHash.argtypes = [sz, ctypes.POINTER(RefHash)]
        lib.ZopfliCleanHash.argtypes = [ctypes.POINTER(RefHash)]
        lib.ZopfliWarmupHash.argtypes = [ctypes.c_char_p, sz, sz, ctypes.POINTER(RefHash)]
        lib.ZopfliUpdateHash.argtypes = [ctypes.c_char_p, sz, sz, ctypes.POINTER(RefHash)]
        lib.ZopfliFindLongestMatch.argtypes = [ctypes.POINTER(RefBlockState), ctypes.POINTER(RefHash), ctypes.c_char_p,
                                               sz, sz, sz, _u16p, _u16p, _u16p]
        lib.ZopfliFindLongestMatch.restype = None
        lib.ZopfliInitLZ77Store.argtypes = [ctypes.c_char_p, ctypes.POINTER(RefStore)]
        lib.ZopfliCleanLZ77Store.argtypes = [ctypes.POINTER(RefStore)]
        lib.ZopfliInitBlockState.argtypes = [ctypes.POINTER(RefOptions), sz, sz, ctypes.c_int,
                                             ctypes.POINTER(RefBlockState)]
        lib.ZopfliCleanBlockState.argtypes = [ctypes.POINTER(RefBlockState)]
        lib.ZopfliLZ77Greedy.argtypes = [ctypes.POINTER(RefBlockState), ctypes.c_char_p, sz, sz,
                                         ctypes.POINTER(RefStore), ctypes.POINTER(RefHash)]
        lib.ZopfliLZ77OptimalFixed.argtypes = [ctypes.POINTER(RefBlockState), ctypes.c_char_p, sz, sz,
                                               ctypes.POINTER(RefStore)]
        lib.ZopfliStoreLitLenDist.argtypes = [ctypes.c_ushort, ctypes.c_ushort, sz, ctypes.POINTER(RefStore)]
        lib.ZopfliStoreLitLenDist.restype = None
        lib.ZopfliLengthLimitedCodeLengths.argtypes = [ctypes.POINTER(sz), ctypes.c_int, ctypes.c_int,
                                                       ctypes.POINTER(ctypes.c_uint)]
        lib.ZopfliCalculateBlockSizeAutoType.argtypes = [ctypes.POINTER(RefStore), sz, sz]
        lib.ZopfliCalculateBlockSizeAutoType.restype = ctypes.c_double
        lib.ZopfliBlockSplitLZ77.argtypes = [ctypes.POINTER(RefOptions), ctypes.POINTER(RefStore), sz,
                                             ctypes.POINTER(ctypes.POINTER(sz)), ctypes.POINTER(sz)]
        lib.ZopfliBlockSplitLZ77.restype = None
        for n in ("ZopfliAllocHash", "ZopfliResetHash", "ZopfliCleanHash", "ZopfliWarmupHash", "ZopfliUpdateHash",
                  "ZopfliInitLZ77Store", "ZopfliCleanLZ77Store", "ZopfliInitBlockState", "ZopfliCleanBlockState",
                  "ZopfliLZ77Greedy", "ZopfliLZ77OptimalFixed"):
            getattr(lib, n).restype = None
        _ref = lib
    return _ref


_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def ref_compress(data, fmt=0, numiterations=15, blocksplitting=1, blocksplittingmax=15, verbose=0, verbose_more=0):
    lib = ref()
    o = RefOptions(verbose, verbose_more, numiterations, blocksplitting, 0, blocksplittingmax)
    out, size = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0)
    lib.ZopfliCompress(ctypes.byref(o), fmt, data, len(data), ctypes.byref(out), ctypes.byref(size))
    r = ctypes.string_at(out, size.value)
    _libc.free(out)
    return r


def ref_deflate_part(data, instart, inend, btype=2, final=1, numiterations=15, blocksplitting=1):
    lib = ref()
    o = RefOptions(0, 0, numiterations, blocksplitting, 0, 15)
    out, size, bp = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
    lib.ZopfliDeflatePart(ctypes.byref(o), btype, final, data, instart, inend, ctypes.byref(bp), ctypes.byref(out),
                          ctypes.byref(size))
    r = ctypes.string_at(out, size.value)
    _libc.free(out)
    return r, bp.value


def ref_match_table(data, instart, inend):
    """ZopfliFindLongestMatch (limit 258, sublen, no cache) at every position of the block, by
    driving the reference's own hash exactly as ZopfliLZ77Greedy / GetBestLengths do."""
    lib = ref()
    h = RefHash()
    lib.ZopfliAllocHash(32768, ctypes.byref(h))
    lib.ZopfliResetHash(32768, ctypes.byref(h))
    ws = instart - 32768 if instart > 32768 else 0
    lib.ZopfliWarmupHash(data, ws, inend, ctypes.byref(h))
    for i in range(ws, instart):
        lib.ZopfliUpdateHash(data, i, inend, ctypes.byref(h))
    o = RefOptions(0, 0, 15, 1, 0, 15)
    s = RefBlockState(ctypes.pointer(o), None, instart, inend)
    out = []
    sub = (ctypes.c_uint16 * 259)()
    d, l = ctypes.c_uint16(0), ctypes.c_uint16(0)
    for i in range(instart, inend):
        lib.ZopfliUpdateHash(data, i, inend, ctypes.byref(h))
        lib.ZopfliFindLongestMatch(ctypes.byref(s), ctypes.byref(h), data, i, inend, 258, sub, ctypes.byref(d),
                                   ctypes.byref(l))
        out.append((l.value, d.value, list(sub[3:l.value + 1]) if l.value >= 3 else []))
    lib.ZopfliCleanHash(ctypes.byref(h))
    return out


def ref_greedy(data, instart, inend):
    lib = ref()
    h = RefHash()
    lib.ZopfliAllocHash(32768, ctypes.byref(h))
    o = RefOptions(0, 0, 15, 1, 0, 15)
    s = RefBlockState()
    lib.ZopfliInitBlockState(ctypes.byref(o), instart, inend, 0, ctypes.byref(s))
    st = RefStore()
    lib.ZopfliInitLZ77Store(data, ctypes.byref(st))
    lib.ZopfliLZ77Greedy(ctypes.byref(s), data, instart, inend, ctypes.byref(st), ctypes.byref(h))
    ll = np.array(st.litlens[:st.size], dtype=np.uint16)
    dd = np.array(st.dists[:st.size], dtype=np.uint16)
    lib.ZopfliCleanLZ77Store(ctypes.byref(st))
    lib.ZopfliCleanBlockState(ctypes.byref(s))
    lib.ZopfliCleanHash(ctypes.byref(h))
    return ll, dd


class RefSymbols:
    """A ZopfliLZ77Store of the real reference filled from (litlen, dist) arrays (lz77.c:98)."""

    def __init__(self, litlens, dists):
        self.lib = ref()
        self.store = RefStore()
        self.lib.ZopfliInitLZ77Store(b"", ctypes.byref(self.store))
        pos = 0
        for l, d in zip(litlens.tolist(), dists.tolist()):
            self.lib.ZopfliStoreLitLenDist(l, d, pos, ctypes.byref(self.store))
            pos += 1 if d == 0 else l

    def block_size_auto(self, lstart, lend):
        return self.lib.ZopfliCalculateBlockSizeAutoType(ctypes.byref(self.store), lstart, lend)

    def block_split(self, maxblocks):
        o = RefOptions(0, 0, 15, 1, 0, 15)
        pts, n = ctypes.POINTER(ctypes.c_size_t)(), ctypes.c_size_t(0)
        self.lib.ZopfliBlockSplitLZ77(ctypes.byref(o), ctypes.byref(self.store), maxblocks, ctypes.byref(pts), ctypes.byref(n))
        out = [pts[i] for i in range(n.value)]
        _libc.free(ctypes.cast(pts, ctypes.c_void_p))
        return out

    def close(self):
        self.lib.ZopfliCleanLZ77Store(ctypes.byref(self.store))


def ref_code_lengths(freq, maxbits):
    n = len(freq)
    f = (ctypes.c_size_t * n)(*[int(x) for x in freq])
    out = (ctypes.c_uint * n)()
    rc = ref().ZopfliLengthLimitedCodeLengths(f, n, maxbits, out)
    return rc, list(out)


def ref_optimal_fixed(data, instart, inend):
    lib = ref()
    o = RefOptions(0, 0, 15, 1, 0, 15)
    s = RefBlockState()
    lib.ZopfliInitBlockState(ctypes.byref(o), instart, inend, 1, ctypes.byref(s))
    st = RefStore()
    lib.ZopfliInitLZ77Store(data, ctypes.byref(st))
    lib.ZopfliLZ77OptimalFixed(ctypes.byref(s), data, instart, inend, ctypes.byref(st))
    ll = np.array(st.litlens[:st.size], dtype=np.uint16)
    dd = np.array(st.dists[:st.size], dtype=np.uint16)
    lib.ZopfliCleanLZ77Store(ctypes.byref(st))
    lib.ZopfliCleanBlockState(ctypes.byref(s))
    return ll, dd


def hosttest_library():
    """Product host code + oracle-backed zmx layer (CPU only)."""
    if not os.path.exists(HOSTTEST_SO):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "hostlib")])
    import sys
    sys.path.insert(0, ROOT)
    from zopfli_amd import api
    return api.library(HOSTTEST_SO)
