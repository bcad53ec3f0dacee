"""Stress of the cooperative run-task job (ZOPFLI_AMD_COOP=1, zmx_dp6.h): the intermittent mismatch.  Class Z, 100 master
blocks: squeeze runs of the whole batch with the same cost model, repeated; every block's length_array is compared with
the first repetition's (one-wave job, ZOPFLI_AMD_COOP=0, in a sibling process writes the reference digests).  Prints which
blocks / cells differ and how they sit relative to the 32-cell windows."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def run(reps, n, dump):
    import oracle_lib as ol
    from zopfli_amd import Context, api, generate
    lib = api.library()
    ctx = Context(0, lib)
    data = generate("Z", n)
    blocks = [(s, min(s + 1000000, n)) for s in range(0, n, 1000000)]
    ctx.set_input(data)
    t = ctx.build_tables(blocks)
    nb = len(blocks)
    nsym, hist = t.greedy(0)
    cost = np.zeros((nb, 320)); mincost = np.zeros(nb)
    for b in range(nb):
        ll, d = ol.entropy_costs(hist[b]); cost[b, :288], cost[b, 288:] = ll, d; mincost[b] = ol.model_min_cost(ll, d)
    out = []
    for r in range(reps):
        t.squeeze_run(cost, mincost, np.zeros(nb, dtype=np.int32))
        out.append([t.length_array(b).copy() for b in range(nb)])
    if dump:
        np.save(dump, np.concatenate(out[0]))
    return out


if __name__ == "__main__":
    n = int(os.environ.get("N", "100000000"))
    if sys.argv[1:] == ["ref"]:
        run(1, n, "/tmp/coop_ref.npy")
        sys.exit(0)
    subprocess.check_call([sys.executable, __file__, "ref"], env=dict(os.environ, ZOPFLI_AMD_COOP="0"))
    ref = np.load("/tmp/coop_ref.npy")
    os.environ["ZOPFLI_AMD_COOP"] = "1"
    reps = int(os.environ.get("REPS", "12"))
    out = run(reps, n, None)
    off = np.cumsum([0] + [len(a) for a in out[0]])
    bad = 0
    for r in range(reps):
        for b, a in enumerate(out[r]):
            want = ref[off[b]:off[b + 1]]
            if not np.array_equal(a, want):
                bad += 1
                d = np.nonzero(a != want)[0]
                print(f"rep {r} block {b}: {len(d)} cells differ, first {d[:8].tolist()} (mod 32: {(d[:8] % 32).tolist()}), gpu {a[d[:8]].tolist()} ref {want[d[:8]].tolist()}", flush=True)
    print(f"{bad} block runs of {reps * len(out[0])} differ", flush=True)
