"""k_match5 (exact skip-walk) against k_match2 on the GPU box, without torch: record digests of whole tables for every
class (zmx_match_digest), then the time of the match kernel and of the hash kernels per class.

    python tools/r04_match5.py parity            # digests, kernels 2 and 5, every class, master-block style blocks
    python tools/r04_match5.py time [T:100000000 X:100000000 ...]
    ZOPFLI_AMD_PROF=1 python tools/r04_match5.py time T:20000000      # entries touched per position on stderr
"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zopfli_amd import Context, api, generate  # noqa: E402

MB = 1000000


def blocks_of(n, size=MB):
    return [(s, min(s + size, n)) for s in range(0, n, size)]


def match_timing(lib):
    m = (ctypes.c_double * 4)()
    lib.zmx_last_match_timing(m)
    return list(m)


def parity(lib, ctx):
    bad = 0
    cases = [(cls, 4 * MB + 12345, None) for cls in "TXRZBPM"]
    cases += [("B", 300000, [(0, 100000), (100000, 100002), (100002, 300000)]), ("Z", 2100000, None),
              ("P", 3 * MB, [(500000, 1500000), (1500000, 3 * MB)]), ("M", 20 * MB, None), ("T", 70000, [(0, 3), (3, 5), (5, 70000)])]
    for cls, n, blocks in cases:
        data = generate(cls, n)
        ctx.set_input(data)
        blocks = blocks or blocks_of(n)
        dig = {}
        for kern in (2, 5, 0):
            lib.zmx_set_match_kernel(kern)
            t = ctx.build_tables(blocks, matches_only=True)
            dig[kern] = t.match_digest()
            t.free()
        same = dig[2] == dig[5] == dig[0]
        bad += not same
        print("parity", cls, n, len(blocks), "blocks:", "identical" if same else "DIFFERENT %r" % (dig,), flush=True)
        if not same:
            # the first differing position, through the per-position probe
            for kern in (2, 5):
                lib.zmx_set_match_kernel(kern)
                dig[kern] = ctx.build_tables(blocks, matches_only=True)
            found = 0
            for b, (s, e) in enumerate(blocks):
                for pos in range(s, e):
                    a = dig[2].find_longest_match(b, pos)
                    c = dig[5].find_longest_match(b, pos)
                    if (a[0], a[1]) != (c[0], c[1]) or (a[0] >= 3 and not (a[2][3:a[0] + 1] == c[2][3:a[0] + 1]).all()):
                        print("   first difference: block", b, "pos", pos, "k_match2", a[0], a[1], "k_match5", c[0], c[1],
                              list(a[2][3:a[0] + 1][:40]), list(c[2][3:c[0] + 1][:40]), flush=True)
                        found += 1
                        if found >= 3:
                            break
                if found >= 3:
                    break
            dig[2].free()
            dig[5].free()
    lib.zmx_set_match_kernel(2)
    return bad


def timing(lib, ctx, specs):
    out = []
    for spec in specs:
        cls, n = spec.split(":")
        n = int(n)
        data = generate(cls, n)
        ctx.set_input(data)
        blocks = blocks_of(n)
        row = {"cls": cls, "size": n}
        for kern in (2, 5, 0):
            lib.zmx_set_match_kernel(kern)
            ctx.build_tables(blocks, matches_only=True).free()       # warm: the pool holds the arrays
            best = None
            for _ in range(2):
                m0 = match_timing(lib)
                t0 = time.perf_counter()
                t = ctx.build_tables(blocks, matches_only=True)
                dt = time.perf_counter() - t0
                m1 = match_timing(lib)
                d = t.match_digest()
                t.free()
                cur = {"wall_ms": round(dt * 1e3, 2), "match_ms": round((m1[0] - m0[0]) * 1e3, 2),
                       "hash_ms": round((m1[1] - m0[1]) * 1e3, 2)}
                if best is None or cur["wall_ms"] < best["wall_ms"]:
                    best = cur
            best["digest"] = "%016x" % d[0]
            row["k%d" % kern] = best
        row["identical"] = row["k2"]["digest"] == row["k5"]["digest"] == row["k0"]["digest"]
        print(json.dumps(row), flush=True)
        out.append(row)
    lib.zmx_set_match_kernel(2)
    return out


def main():
    lib = api.library()
    ctx = Context(0, lib)
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    rc = 0
    if what == "parity":
        rc = parity(lib, ctx)
    else:
        specs = sys.argv[2:] or ["T:100000000", "X:100000000", "P:20000000", "B:20000000", "Z:20000000", "M:20000000", "R:20000000"]
        rows = timing(lib, ctx, specs)
        rc = sum(not r["identical"] for r in rows)
    ctx.close()
    sys.exit(1 if rc else 0)


if __name__ == "__main__":
    main()
