import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from zopfli_amd import Context, api, generate
lib = api.library(); ctx = Context(0, lib)
data = generate("Z", 90000); blocks = [(0, 45001)]
ctx.set_input(data); t = ctx.build_tables(blocks)
nsym, hist = t.greedy(0)
o = ol.OracleTable(data, 0, 45001)
ll = np.array([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, dtype=np.float64); d = np.full(32, 5.0)
cost = np.zeros((1, 320)); cost[0, :288], cost[0, 288:] = ll, d
mc = np.array([ol.model_min_cost(ll, d)])
print("mincost", mc)
try:
    t.squeeze_run(cost, mc, np.zeros(1, dtype=np.int32))
except Exception as e:
    print("ERR", e)
la, _, _ = o.squeeze_run(ll, d, mc[0])
g = t.length_array(0)
bad = np.nonzero(g[1:] != la[1:])[0] + 1
print("nbad", len(bad), bad[:40])
for a in (640, 700):
    print(a, "gpu", g[a:a+24].tolist()); print(a, "ref", la[a:a+24].tolist())
print(api.last_seg_stats(lib))
