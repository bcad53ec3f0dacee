import sys, ctypes, time
sys.path.insert(0, "/root/repo")
from zopfli_amd import Context, api, generate
lib = api.library()
ctx = Context(0, lib)
prev = [0.0, 0.0, 0.0]
for cls, n in (("P", 20000000), ("B", 20000000), ("T", 20000000)):
    data = generate(cls, n)
    ctx.set_input(data)
    blocks = [(s, min(s + 1000000, n)) for s in range(0, n, 1000000)]
    dig = {}
    for kern in (2, 5):
        lib.zmx_set_match_kernel(kern)
        t0 = time.time()
        t = ctx.build_tables(blocks, matches_only=True)
        dig[kern] = t.match_digest()
        t.free()
        w = (ctypes.c_double * 3)()
        lib.zmx_last_match_walk(w)
        d = [w[i] - prev[i] for i in range(3)]
        prev = list(w)
        print(cls, "kernel", kern, "build %.1f ms" % ((time.time() - t0) * 1e3), "walk stats", d,
              ("entries/pos %.2f lanes/iter %.1f" % (d[0] / d[2], d[0] / max(d[1], 1))) if d[2] else "", flush=True)
    print(cls, "digests equal", dig[2] == dig[5], flush=True)
lib.zmx_set_match_kernel(0)
