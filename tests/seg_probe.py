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

def dyadic_costs(hist):
    """A cost model whose weights are dyadic rationals (SEG_PROBE_COSTS=dyadic): every cost is a small integer
    plus an odd multiple of 2^-k, k = 4..11 by symbol.  w mod ulp(e) = ulp(e)/2 then holds in binade e = k + 16
    (float ulp 2^(e-23)), so the sum of such a weight and a float of that binade lands exactly half-way between
    two floats (squeeze.c:281-299 rounds it to even): the tie rule of the chain's acceptance test
    (zmx_dp4.h d4_accept -> 2, the integer table refused) has to re-run every task whose values lie in a
    masked binade, 2^13 .. 2^21 bits here — entropy costs never get there."""
    ll = np.zeros(288)
    d = np.zeros(32)
    total = max(int(hist[:288].sum()), 1)
    for i in range(288):
        base = 3 + (int(np.log2(total / max(int(hist[i]), 1))) if hist[i] else 9)
        ll[i] = base + 2.0 ** -(4 + i % 8)
    for i in range(32):
        d[i] = 2 + i % 5 + 2.0 ** -(4 + (i * 3) % 8)
    return ll, d


CASES = [
    ("T", 200000, [(0, 120000), (120000, 200000)]),
    ("X", 100000, [(30000, 100000)]),
    ("M", 400000, [(0, 400000)]),
    ("Z", 150000, [(0, 150000)]),
    ("B", 60000, [(0, 60000)]),
]


def main():
    global CASES
    dyadic = os.environ.get("SEG_PROBE_COSTS") == "dyadic"
    if dyadic:   # one long block: its costs pass 2^20 bits, through every binade the weights can tie in
        CASES = [("T", 400000, [(0, 400000)]), ("X", 300000, [(10000, 300000)])]
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
                ll, d = dyadic_costs(hist[b]) if dyadic else ol.entropy_costs(hist[b])
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
