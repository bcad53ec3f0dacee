import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from zopfli_amd import Context, api, generate
lib = api.library(); ctx = Context(0, lib)
n = int(os.environ.get("DBG_N", "1000000"))
data = generate(os.environ.get("DBG_CLS", "Z"), n); blocks = [(0, n)]
ctx.set_input(data); t = ctx.build_tables(blocks)
nsym, hist = t.greedy(0)
for it in range(2):
    ll, d = ol.entropy_costs(hist[0]); cost = np.zeros((1, 320)); cost[0, :288], cost[0, 288:] = ll, d
    mc = np.array([ol.model_min_cost(ll, d)])
    nsym, hist = t.squeeze_run(cost, mc, np.zeros(1, dtype=np.int32))
    print("RUN", it, api.last_seg_stats(lib), flush=True)
