"""Run by tests/test_gpu_parity.py::test_chain_task_paths in a subprocess (the task geometry of the chain
kernels is read from the environment once per process): three chained squeeze runs per case against the
CPU oracle, then one JSON line with how the chain's tasks fared (zmx_last_seg_stats)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import oracle_lib as ol  # noqa: E402
from zopfli_amd import Context, api, generate  # noqa: E402

CASES = [
    ("T", 200000, [(0, 120000), (120000, 200000)]),
    ("X", 100000, [(30000, 100000)]),
    ("M", 400000, [(0, 400000)]),
    ("Z", 150000, [(0, 150000)]),
    ("B", 60000, [(0, 60000)]),
]


def main():
    global CASES
    if os.environ.get("SEG_PROBE_CASES"):
        CASES = [c for c in CASES if c[0] in os.environ["SEG_PROBE_CASES"]]
    lib = api.library()
    ctx = Context(0, lib)
    for cls, n, blocks in CASES:
        data = generate(cls, n)
        ctx.set_input(data)
        t = ctx.build_tables(blocks)
        nb = len(blocks)
        nsym, hist = t.greedy(0)
        tables = [ol.OracleTable(data, s, e) for (s, e) in blocks]
        for it in range(3):
            cost = np.zeros((nb, 320))
            mincost = np.zeros(nb)
            for b in range(nb):
                ll, d = ol.entropy_costs(hist[b])
                cost[b, :288], cost[b, 288:] = ll, d
                mincost[b] = ol.model_min_cost(ll, d)
            nsym, hist = t.squeeze_run(cost, mincost, np.full(nb, it & 1, dtype=np.int32))
            for b, (s, e) in enumerate(blocks):
                la, oll, odd = tables[b].squeeze_run(cost[b, :288], cost[b, 288:], mincost[b])
                gla = t.length_array(b)
                if not np.array_equal(gla[1:], la[1:]):
                    bad = int(np.nonzero(gla[1:] != la[1:])[0][0]) + 1
                    print(f"MISMATCH {cls} iter {it} block {b}: length_array differs first at {bad}", flush=True)
                    sys.exit(1)
                gll, gdd = t.store(b, it & 1, nsym[b])
                if not (np.array_equal(gll, oll) and np.array_equal(gdd, odd)):
                    print(f"MISMATCH {cls} iter {it} block {b}: store", flush=True)
                    sys.exit(1)
        t.free()
    ctx.close()
    print(json.dumps(api.last_seg_stats(lib)), flush=True)


if __name__ == "__main__":
    main()
