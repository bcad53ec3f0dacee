"""Small-call latency of the C entry points on a warm context: wall ms of ZopfliCompress (gzip, default
options unless stated) for the sizes of BASELINE config 0 (64 KiB) and of zopflipng's IDATs (<= 1 MB)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zopfli_amd import ZopfliOptions, api, generate  # noqa: E402

lib = api.library()
res = []
for cls, size, n in (("T", 65536, 1), ("T", 65536, 15), ("T", 1000000, 15), ("P", 1000000, 15), ("T", 4000000, 15)):
    data = generate(cls, size)
    opt = ZopfliOptions(n)
    for _ in range(8 if size > 3000000 else 1):  # warm: context, table pool, kernels loaded (a call of 4 master blocks
        api.compress(data, 0, opt, lib=lib)      # or more is dealt over three contexts from the process's eighth such call on)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out = api.compress(data, 0, opt, lib=lib)
        ts.append((time.perf_counter() - t0) * 1e3)
    t = api.last_timing(lib)
    res.append({"cls": cls, "size": size, "numiterations": n, "ms_min": round(min(ts), 2), "ms_median": round(sorted(ts)[2], 2),
                "out": len(out), "breakdown_ms": {k: round(v * 1e3, 2) for k, v in t.items()
                                                  if k in ("tables", "greedy", "squeeze", "cost_model", "split", "encode",
                                                           "download", "dp_kernel", "wtab_kernel", "trace_kernel")}})
    print(json.dumps(res[-1]), flush=True)
