"""round 4: the block-split search on the host (BlockSplitLz77 / BlockSplitLz77Batch of the product's host sources, through
the CPU test build): time per search on this box's cores, sequential and round by round on the pool."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from zopfli_amd import generate  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "libzopfli_hosttest.so"))
lib.zamd_test_block_split.restype = ctypes.c_size_t
lib.zamd_test_block_split.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
lib.zamd_test_block_split_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
for cls, size in (("T", 65536), ("T", 1000000), ("P", 1000000)):
    data = generate(cls, size)
    ll, dd = ol.OracleTable(data, 0, size).greedy()
    ll = np.ascontiguousarray(ll, dtype=np.uint16)
    dd = np.ascontiguousarray(dd, dtype=np.uint16)
    n = len(ll)
    pts = (ctypes.c_size_t * 256)()
    cnt = (ctypes.c_size_t * 8)()
    res = {}
    for name, fn in (("sequential", lambda: lib.zamd_test_block_split(ll.ctypes.data, dd.ctypes.data, n, 15, pts, 64)),
                     ("batch x1", lambda: lib.zamd_test_block_split_batch(ll.ctypes.data, dd.ctypes.data, n, 1, 15, pts, cnt, 32)),
                     ("batch x4", lambda: lib.zamd_test_block_split_batch(ll.ctypes.data, dd.ctypes.data, n, 4, 15, pts, cnt, 32))):
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        res[name] = round(min(ts) * 1e3, 2)
    print(cls, size, "symbols", n, res, flush=True)
