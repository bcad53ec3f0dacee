"""The C-ABI library loads and exports every symbol include/zopfli_amd.h declares (no compute calls)."""
import ctypes
import os
import re

from zopfli_amd._build import LIB, ROOT


def test_exports_match_header():
    assert os.path.exists(LIB), "libzopfli_amd.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(LIB)
    with open(os.path.join(ROOT, "include", "zopfli_amd.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b((?:Zopfli|zmx_)\w+)\s*\(", text))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in include/zopfli_amd.h but not exported"


def test_png_library_exports_the_reference_abi():
    """libzopflipng_amd.so has the C entry points of the reference's zopflipng_lib.h (:76-85) and its defaults
    (zopflipng_lib.cc:33-43); the C++ ones are checked by linking the reference's own command line against it
    (_build.build_png_tests).  No device call."""
    import pytest
    from zopfli_amd._build import PNG_LIB
    if not os.path.exists(PNG_LIB):
        pytest.skip("libzopflipng_amd.so not built (LodePNG's sources were not at hand)")
    lib = ctypes.CDLL(PNG_LIB)

    class COpts(ctypes.Structure):    # zopflipng_lib.h:51-72
        _fields_ = [("lossy_transparent", ctypes.c_int), ("lossy_8bit", ctypes.c_int), ("filter_strategies", ctypes.c_void_p),
                    ("num_filter_strategies", ctypes.c_int), ("auto_filter_strategy", ctypes.c_int), ("keepchunks", ctypes.c_void_p),
                    ("num_keepchunks", ctypes.c_int), ("use_zopfli", ctypes.c_int), ("num_iterations", ctypes.c_int),
                    ("num_iterations_large", ctypes.c_int), ("block_split_strategy", ctypes.c_int)]
    o = COpts()
    ctypes.memset(ctypes.byref(o), 0xff, ctypes.sizeof(o))
    lib.CZopfliPNGSetDefaults(ctypes.byref(o))
    assert (o.lossy_transparent, o.lossy_8bit, o.num_filter_strategies, o.auto_filter_strategy, o.num_keepchunks, o.use_zopfli,
            o.num_iterations, o.num_iterations_large, o.block_split_strategy) == (0, 0, 0, 1, 0, 1, 15, 5, 1)
    assert o.filter_strategies is None and o.keepchunks is None
    assert hasattr(lib, "CZopfliPNGOptimize")
    with open(os.path.join(ROOT, "include", "zopflipng_amd.h")) as f:
        text = f.read()
    for name in ("CZopfliPNGSetDefaults", "CZopfliPNGOptimize", "ZopfliPNGOptimize", "kStrategyBruteForce"):
        assert name in text


def test_options_layout():
    from zopfli_amd import ZopfliOptions, api
    lib = api.library()
    o = ZopfliOptions(1, 0, 1)
    lib.ZopfliInitOptions(ctypes.byref(o))
    assert (o.verbose, o.verbose_more, o.numiterations, o.blocksplitting, o.blocksplittinglast,
            o.blocksplittingmax) == (0, 0, 15, 1, 0, 15)
    assert ctypes.sizeof(ZopfliOptions) == 24


def _ref_run_info(cost):
    """The tie mask and the weight bound of a cost model in exact rational arithmetic: a float c of
    binade e plus an edge weight w is computed as dbl(w + c) = c + RNE(w / g) g with g = 2^(e-52),
    and the float rounding of that sum ties iff the remainder modulo 2^(e-23) is half of it."""
    from fractions import Fraction
    ll, d = cost[:288], cost[288:]

    def lbits(s):
        return 0 if s < 265 or s == 285 else (s - 261) // 4

    def dbits(s):
        return 0 if s < 4 else s // 2 - 1

    ws = [float(x) for x in ll[:256]]
    for ls in range(257, 286):
        for ds in range(30):
            ws.append((float(lbits(ls) + dbits(ds)) + float(ll[ls])) + float(d[ds]))
    mask = 0
    for w in ws:
        if w <= 0:
            continue
        fw = Fraction(w)
        for e in range(4, 32):
            q = fw / Fraction(2) ** (e - 52)
            fl = q.numerator // q.denominator
            rem = q - fl
            r = fl + (1 if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1) else 0)
            if r % (1 << 29) == (1 << 28):
                mask |= 1 << e
    return max(ws), mask


def test_chain_acceptance_facts():
    """zmx_hip.hip RunInfo (what k_dp4_fix's acceptance test knows about a cost model) against exact
    arithmetic: random entropy-like tables never tie; weights built to tie in a chosen binade do."""
    import numpy as np
    from zopfli_amd import api
    lib = api.library()
    lib.zmx_internal_run_info.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float),
                                          ctypes.POINTER(ctypes.c_uint32)]
    lib.zmx_internal_run_info.restype = None
    rng = np.random.default_rng(5)

    def run(cost):
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        wmax, mask = ctypes.c_float(0), ctypes.c_uint32(0)
        lib.zmx_internal_run_info(cost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(wmax),
                                  ctypes.byref(mask))
        return wmax.value, mask.value

    for _ in range(3):
        cost = np.log2(1.0 / rng.dirichlet(np.ones(320) * 0.3)).clip(0, 40)
        wmax, mask = run(cost)
        rmax, rmask = _ref_run_info(cost)
        assert mask == rmask
        assert wmax >= rmax
    # a literal cost of 5.0625 = 40.5 float ulps of binade 20 (ulp 1/8): ties there, and in the
    # binades whose ulp it is also an odd multiple of a half of
    cost = np.full(320, 3.0)
    cost[65] = 5.0625
    wmax, mask = run(cost)
    rmax, rmask = _ref_run_info(cost)
    assert mask == rmask and (mask >> 20) & 1
    # a match cost that only ties after the double rounding of the sum: 2^-30 above a half ulp of binade 24
    cost = np.full(320, 2.0)
    cost[288] = 1.0 + 2.0 ** -29
    assert run(cost)[1] == _ref_run_info(cost)[1]
