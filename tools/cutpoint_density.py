"""How often does the DP offer an exact cut point?  (CPU only; test infrastructure: uses the oracle.)

A position q that no DP edge crosses — p + kend(p) <= q for every p < q, kend(p) = the longest match
at p (ZopfliFindLongestMatch, lz77.c:407) or 1 for the literal — splits GetBestLengths (squeeze.c:217)
exactly; k_cutpoints (zopfli_amd/csrc/device/zmx_dp5.h) starts the chain's tasks there.  This prints,
per class of tests' synthetic data, the density of such positions: the numbers quoted in DESIGN.md
section 4 ("Cut points").

    python tools/cutpoint_density.py TXZ 150000
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from zopfli_amd import generate  # noqa: E402


def density(cls, n):
    data = generate(cls, n)
    t = ol.OracleTable(data, 0, n)
    sub = np.zeros(259, dtype=np.uint16)
    d, l = ctypes.c_uint16(0), ctypes.c_uint16(0)
    kend = np.ones(n, dtype=np.int64)
    for p in range(n):
        t.lib.zo_find_longest_match(t.h, p, sub.ctypes.data_as(ol._u16p), ctypes.byref(d), ctypes.byref(l))
        if l.value >= 3:
            kend[p] = min(l.value, n - p)        # squeeze.c:286
    reach = np.maximum.accumulate(np.arange(n) + kend)      # reach[q - 1] = furthest cell an edge from below q gets to
    cuts = np.flatnonzero(reach <= np.arange(1, n + 1)) + 1
    gaps = np.diff(np.concatenate(([0], cuts)))
    return len(cuts), gaps


if __name__ == "__main__":
    classes = sys.argv[1] if len(sys.argv) > 1 else "TXZ"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 150000
    for cls in classes:
        k, gaps = density(cls, n)
        if k:
            print(f"class {cls}: {k} cut points in {n} positions, mean gap {gaps.mean():.1f}, "
                  f"p90 {int(np.percentile(gaps, 90))}, p99 {int(np.percentile(gaps, 99))}, max {gaps.max()}")
        else:
            print(f"class {cls}: no cut point in {n} positions")
