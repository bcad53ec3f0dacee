"""Parity of the HIP kernels against the CPU oracle / the real reference, through the C ABI of
libzopfli_amd.so on a real MI355X.  Bit-exact: everything on this path is integer work or
IEEE double/float arithmetic in a fixed order (no tolerance)."""
import gzip
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from zopfli_amd import ZopfliOptions, api, generate

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vectors.json")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (class, total size, blocks [(instart, inend), ...])
TABLE_CASES = [
    ("T", 70000, [(0, 70000)]),
    ("X", 50000, [(0, 20000), (20000, 50000)]),
    ("Z", 90000, [(0, 45001), (45001, 90000)]),        # runs crossing a block end
    ("B", 40000, [(0, 40000)]),                        # chain cap + hash switch
    ("R", 30000, [(0, 30000)]),
    ("P", 66000, [(33000, 66000)]),                    # window before instart
    ("M", 150000, [(0, 3), (3, 5), (5, 5), (5, 100000), (100000, 150000)]),  # tiny / empty blocks
]


def _ids(c):
    return f"{c[0]}{c[1]}x{len(c[2])}"


def _first_diff(a, b):
    n = min(len(a), len(b))
    idx = np.nonzero(np.asarray(a[:n]) != np.asarray(b[:n]))[0]
    return int(idx[0]) if len(idx) else n


@pytest.mark.parametrize("case", TABLE_CASES, ids=_ids)
def test_match_table(gpu_ctx, case):
    """k_same + k_chain + k_match == ZopfliFindLongestMatch at every position (lz77.c:407)."""
    cls, n, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables(blocks)
    try:
        for b, (s, e) in enumerate(blocks):
            o = ol.OracleTable(data, s, e)
            bad = []
            for pos in range(s, e):
                gl, gd, gsub = t.find_longest_match(b, pos)
                ol_, od, osub = o.find_longest_match(pos)
                same = (gl == ol_ and gd == od) if ol_ >= 3 else (gl < 3 and ol_ < 3)
                if same and ol_ >= 3:
                    same = np.array_equal(gsub[3:ol_ + 1], osub[3:ol_ + 1])
                if not same:
                    bad.append((pos, (gl, gd), (ol_, od)))
                    if len(bad) >= 5:
                        break
            assert not bad, f"block {b} [{s},{e}): first mismatches (pos, gpu, oracle) {bad}"
    finally:
        t.free()


@pytest.mark.parametrize("case", TABLE_CASES, ids=_ids)
def test_match_table_skip_walk(gpu_ctx, case):
    """k_levels + k_rank2 + k_match5 (the exact skip-walk: level links, hits counted from ranks, zmx_match5.h) ==
    ZopfliFindLongestMatch at every position, on every class — forced for every block (kernel 5), whatever k_hits
    would choose."""
    cls, n, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    gpu_ctx.lib.zmx_set_match_kernel(5)
    try:
        t = gpu_ctx.build_tables(blocks)
    finally:
        gpu_ctx.lib.zmx_set_match_kernel(0)
    try:
        for b, (s, e) in enumerate(blocks):
            o = ol.OracleTable(data, s, e)
            bad = []
            for pos in range(s, e):
                gl, gd, gsub = t.find_longest_match(b, pos)
                ol_, od, osub = o.find_longest_match(pos)
                same = (gl == ol_ and gd == od) if ol_ >= 3 else (gl < 3 and ol_ < 3)
                if same and ol_ >= 3:
                    same = np.array_equal(gsub[3:ol_ + 1], osub[3:ol_ + 1])
                if not same:
                    bad.append((pos, (gl, gd), (ol_, od)))
                    if len(bad) >= 5:
                        break
            assert not bad, f"block {b} [{s},{e}): first mismatches (pos, gpu, oracle) {bad}"
    finally:
        t.free()


@pytest.mark.parametrize("cls", list("TXRZBPM"))
def test_match_digests_at_size(gpu_ctx, cls):
    """Every match record of 4 MB of every class (5 master blocks: windows, block ends, the hit cap and the hash
    switch on B / Z / P) is the same from k_match2, from k_match5 and from the default per-block choice between
    them (zmx_match_digest: a hash of the logical content of all records)."""
    n = 4012345
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    blocks = [(s, min(s + 1000000, n)) for s in range(0, n, 1000000)]
    dig = {}
    try:
        kernels = (2, 5, 0) + ((3, 4) if _has_experiments() else ())
        for kern in kernels:
            assert gpu_ctx.lib.zmx_set_match_kernel(kern) == 0
            t = gpu_ctx.build_tables(blocks, matches_only=True)
            dig[kern] = t.match_digest()
            t.free()
        if not _has_experiments():      # the shipped library refuses the kernels it does not contain
            assert gpu_ctx.lib.zmx_set_match_kernel(3) != 0 and gpu_ctx.lib.zmx_set_match_kernel(4) != 0
    finally:
        gpu_ctx.lib.zmx_set_match_kernel(0)
    assert len(set(dig.values())) == 1, dig


@pytest.mark.parametrize("case", TABLE_CASES, ids=_ids)
def test_hash_links(gpu_ctx, case):
    """k_same + k_chain == the reference's hash state as static arrays (hash.c:100-137): same[] and the
    links to the previous position of the same hash value, for both hashes, at every position of the
    block and of the window before it."""
    cls, n, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables(blocks)
    try:
        for b, (s, e) in enumerate(blocks):
            if e == s:
                continue
            got = t.hash_links(b)
            want = ol.OracleTable(data, s, e).hash_links()
            for name, g, w in zip(("same", "prev1", "prev2"), got, want):
                assert np.array_equal(g, w), f"block {b} {name}: first diff at window position {_first_diff(g, w)}"
    finally:
        t.free()


def test_change_point_pool_overflow_and_retry():
    """The change-point pool (records with more than 8 sublen change points) starting far too small
    (ZOPFLI_AMD_POOL_ENTRIES): the build overflows and retries with a larger pool, a table built from a
    parent whose pool is full falls through to a build of its own — the match records must not change."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import Context, api, generate\n"
        "lib = api.library()\n"
        "ctx = Context(0, lib)\n"
        "h = hashlib.sha256()\n"
        "for cls, n, parents, blocks in (('B', 60000, [(0, 60000)], [(0, 30000), (30000, 60000)]),\n"
        "                                ('Z', 200000, [(0, 200000)], [(0, 40001), (40001, 200000)])):\n"
        "    data = generate(cls, n)\n"
        "    ctx.set_input(data)\n"
        "    pt = ctx.build_tables(parents)\n"
        "    t = ctx.build_tables(blocks, parent=pt)\n"
        "    pt.free()\n"
        "    for b, (s, e) in enumerate(blocks):\n"
        "        for pos in range(s, e, 3):\n"
        "            l, d, sub = t.find_longest_match(b, pos)\n"
        "            h.update(np.array([l, d], dtype=np.uint16).tobytes() + sub[3:l + 1].tobytes())\n"
        "    t.free()\n"
        "print(h.hexdigest())\n" % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for entries in ("", "1000"):
        env = dict(os.environ)
        if entries:
            env["ZOPFLI_AMD_POOL_ENTRIES"] = entries
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[entries] = r.stdout.split()[-1]
    assert res[""] == res["1000"]


def test_match_kernels_agree():
    """The match-table kernels (ZOPFLI_AMD_MATCH: 2 = k_chain + k_match2 on prev links, 3 = k_bucket + k_match3,
    a wave per position on sorted candidate slices, 4 = k_bucket + k_match4, the slices streamed by a lane per position,
    5 = k_match5, the exact skip-walk on level links with counted hits, 0 = the default: k_match5 or k_match2 per block)
    produce the same records — (length, distance, sublen) at every position of every class, blocks with a window in
    front, tables built from a parent, and a change-point pool that overflows — and the same hash arrays
    (zmx_hash_links_download reads k_bucket's sorted / rank / bucket arrays back as prev links).  Kernel 2 is the one
    test_match_table checks against the oracle position by position."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "import numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import Context, api, generate\n"
        "lib = api.library()\n"
        "ctx = Context(0, lib)\n"
        "h = hashlib.sha256()\n"
        "cases = [('T', 70000, None, [(0, 70000)]), ('X', 50000, None, [(0, 20000), (20000, 50000)]),\n"
        "         ('Z', 90000, None, [(0, 45001), (45001, 90000)]), ('B', 40000, None, [(0, 40000)]),\n"
        "         ('R', 30000, None, [(0, 30000)]), ('P', 66000, None, [(33000, 66000)]),\n"
        "         ('M', 150000, None, [(0, 3), (3, 5), (5, 5), (5, 100000), (100000, 150000)]),\n"
        "         ('Z', 200000, [(0, 100000), (100000, 200000)], [(0, 40001), (40001, 100000), (100000, 100300), (100300, 200000)]),\n"
        "         ('B', 60000, [(0, 60000)], [(0, 30000), (30000, 60000)])]\n"
        "for cls, n, parents, blocks in cases:\n"
        "    data = generate(cls, n)\n"
        "    ctx.set_input(data)\n"
        "    pt = ctx.build_tables(parents) if parents else None\n"
        "    t = ctx.build_tables(blocks, parent=pt) if pt else ctx.build_tables(blocks)\n"
        "    if pt: pt.free()\n"
        "    for b, (s, e) in enumerate(blocks):\n"
        "        for pos in range(s, e):\n"
        "            l, d, sub = t.find_longest_match(b, pos)\n"
        "            h.update(np.array([l if l >= 3 else 0, d if l >= 3 else 0], dtype=np.uint16).tobytes() + sub[3:l + 1].tobytes())\n"
        "        if e > s and not parents:\n"
        "            for a in t.hash_links(b): h.update(np.ascontiguousarray(a).tobytes())\n"
        "    t.free()\n"
        "print(h.hexdigest())\n" % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for kern in ("2", "3", "4", "5", "0"):
        for entries in ("", "2000"):
            env = dict(os.environ, ZOPFLI_AMD_MATCH=kern)
            if entries:
                env["ZOPFLI_AMD_POOL_ENTRIES"] = entries
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
            assert r.returncode == 0, (kern, entries, r.stderr[-2000:])
            res[kern + entries] = r.stdout.split()[-1]
    assert len(set(res.values())) == 1, res


# (class, total size, parent blocks, sub-blocks)
REUSE_CASES = [
    ("T", 120000, [(0, 120000)], [(0, 50000), (50000, 50100), (50100, 120000)]),
    ("Z", 200000, [(0, 100000), (100000, 200000)], [(0, 40001), (40001, 100000), (100000, 100300), (100300, 200000)]),
    ("B", 60000, [(0, 60000)], [(0, 30000), (30000, 60000)]),          # hash switch + chain cap near the new end
    ("M", 150000, [(0, 150000)], [(0, 3), (3, 5), (5, 5), (5, 70000), (70000, 150000)]),
    ("P", 90000, [(20000, 90000)], [(20000, 55000), (55000, 90000)]),  # window before the parent's start
]


@pytest.mark.parametrize("matches_only", [True, False], ids=["parent_matches_only", "parent_full"])
@pytest.mark.parametrize("case", REUSE_CASES, ids=lambda c: f"{c[0]}{c[1]}x{len(c[3])}")
def test_match_table_reuse(gpu_ctx, case, matches_only):
    """zmx_tables_build_from (records copied from the enclosing blocks' tables, tiles near the new
    block ends recomputed) == ZopfliFindLongestMatch at every position of every sub-block.  The parent as the
    library builds it for the greedy pass over master blocks (zmx_tables_build_matches), and as full tables."""
    cls, n, parents, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    pt = gpu_ctx.build_tables(parents, matches_only=matches_only)
    t = gpu_ctx.build_tables(blocks, parent=pt)
    pt.free()
    try:
        for b, (s, e) in enumerate(blocks):
            o = ol.OracleTable(data, s, e)
            bad = []
            for pos in range(s, e):
                gl, gd, gsub = t.find_longest_match(b, pos)
                ol_, od, osub = o.find_longest_match(pos)
                same = (gl == ol_ and gd == od) if ol_ >= 3 else (gl < 3 and ol_ < 3)
                if same and ol_ >= 3:
                    same = np.array_equal(gsub[3:ol_ + 1], osub[3:ol_ + 1])
                if not same:
                    bad.append((pos, (gl, gd), (ol_, od)))
                    if len(bad) >= 5:
                        break
            assert not bad, f"block {b} [{s},{e}): first mismatches (pos, gpu, oracle) {bad}"
    finally:
        t.free()


def test_matches_only_tables_refuse_the_squeeze(gpu_ctx):
    """zmx_tables_build_matches: greedy works, zmx_squeeze_run fails with a message (no DP rows were built)."""
    data = generate("T", 40000)
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables([(0, len(data))], matches_only=True)
    try:
        nsym, _ = t.greedy(0)
        assert nsym[0] == len(ol.OracleTable(data, 0, len(data)).greedy()[0])
        with pytest.raises(Exception, match="matches only"):
            t.squeeze_run(np.ones((1, 320)) * 8.0, np.array([8.0]), [1])
    finally:
        t.free()


def test_trimmed_tables_keep_their_stores(gpu_ctx):
    """zmx_tables_trim: the stores can still be downloaded, everything else fails with a message."""
    data = generate("X", 60000)
    blocks = [(0, 30000), (30000, 60000)]
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables(blocks)
    try:
        nsym, _ = t.greedy(1)
        before = [t.store(b, 1, nsym[b]) for b in range(2)]
        t.trim()
        for b in range(2):
            ll, dd = t.store(b, 1, nsym[b])
            assert np.array_equal(ll, before[b][0]) and np.array_equal(dd, before[b][1])
            oll, odd = ol.OracleTable(data, *blocks[b]).greedy()
            assert np.array_equal(ll, oll) and np.array_equal(dd, odd)
        with pytest.raises(Exception, match="trimmed"):
            t.greedy(0)
        with pytest.raises(Exception, match="trimmed"):
            t.squeeze_run(np.ones((2, 320)) * 8.0, np.array([8.0, 8.0]), [0, 0])
    finally:
        t.free()


@pytest.mark.parametrize("matches_only", [False, True], ids=["full", "matches_only"])
@pytest.mark.parametrize("case", TABLE_CASES, ids=_ids)
def test_greedy(gpu_ctx, case, matches_only):
    """k_greedy == ZopfliLZ77Greedy (lz77.c:544) + its histogram."""
    cls, n, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables(blocks, matches_only=matches_only)
    try:
        nsym, hist = t.greedy(0)
        for b, (s, e) in enumerate(blocks):
            oll, odd = ol.OracleTable(data, s, e).greedy()
            gll, gdd = t.store(b, 0, nsym[b])
            assert nsym[b] == len(oll), f"block {b}: nsym {nsym[b]} vs {len(oll)}"
            assert np.array_equal(gll, oll) and np.array_equal(gdd, odd), \
                f"block {b}: first diff at symbol {min(_first_diff(gll, oll), _first_diff(gdd, odd))}"
            assert np.array_equal(hist[b], ol.histogram(oll, odd)), f"block {b}: histogram"
    finally:
        t.free()


@pytest.mark.parametrize("case", TABLE_CASES, ids=_ids)
def test_squeeze_runs(gpu_ctx, case):
    """k_squeeze == GetBestLengths + TraceBackwards + FollowPath (squeeze.c:217,317,338): three
    chained runs (greedy statistics, then each run's own statistics) and one fixed-tree run."""
    cls, n, blocks = case
    data = generate(cls, n)
    gpu_ctx.set_input(data)
    t = gpu_ctx.build_tables(blocks)
    try:
        nb = len(blocks)
        nsym, hist = t.greedy(0)
        tables = [ol.OracleTable(data, s, e) for (s, e) in blocks]
        fixed_ll = np.array([8] * 144 + [9] * 112 + [7] * 24 + [8] * 8, dtype=np.float64)
        fixed_d = np.full(32, 5.0)
        for it in range(4):
            cost = np.zeros((nb, 320))
            mincost = np.zeros(nb)
            for b in range(nb):
                if it == 3:
                    ll, d = fixed_ll, fixed_d
                else:
                    ll, d = ol.entropy_costs(hist[b])
                cost[b, :288], cost[b, 288:] = ll, d
                mincost[b] = ol.model_min_cost(ll, d)
            slot = np.full(nb, it & 1, dtype=np.int32)
            nsym, hist = t.squeeze_run(cost, mincost, slot)
            for b, (s, e) in enumerate(blocks):
                la, oll, odd = tables[b].squeeze_run(cost[b, :288], cost[b, 288:], mincost[b])
                if e > s:
                    gla = t.length_array(b)
                    assert np.array_equal(gla[1:], la[1:]), \
                        f"iter {it} block {b}: length_array first diff at {1 + _first_diff(gla[1:], la[1:])}"
                gll, gdd = t.store(b, it & 1, nsym[b])
                assert nsym[b] == len(oll), f"iter {it} block {b}: nsym {nsym[b]} vs {len(oll)}"
                assert np.array_equal(gll, oll) and np.array_equal(gdd, odd), f"iter {it} block {b}: store"
                assert np.array_equal(hist[b], ol.histogram(oll, odd)), f"iter {it} block {b}: histogram"
    finally:
        t.free()


# (environment, what the task statistics must show)
CHAIN_ENVS = [
    ({}, lambda st: st["tasks"] > 100 and st["accepted"] > 0),
    ({"ZOPFLI_AMD_SEG_L": "0"}, lambda st: st["tasks"] == 0),                                   # the serial chain
    ({"ZOPFLI_AMD_SEG_WARM": "64", "ZOPFLI_AMD_SEG_HEAD": "0", "ZOPFLI_AMD_SEG_CUTS": "0"}, lambda st: st["rerun_state"] > 0),                           # warm-up too short: states differ
    ({"ZOPFLI_AMD_SEG_SCALE": "1.9"}, lambda st: st["rerun_level"] + st["rerun_values"] > 0),                         # wrong binade guessed
    ({"ZOPFLI_AMD_SEG_L": "1024", "ZOPFLI_AMD_SEG_WARM": "256", "ZOPFLI_AMD_SEG_HEAD": "4096"}, lambda st: st["tasks"] > 400),
    ({"ZOPFLI_AMD_INT_PATH": "0"}, lambda st: st["accepted"] > 0),                              # every window in the reference's doubles
    ({"ZOPFLI_AMD_FIX_LEAN": "0"}, lambda st: st["rerun_state"] + st["rerun_level"] + st["rerun_values"] > 0),   # serial re-runs by the lean one-wave job
    ({"ZOPFLI_AMD_SEG_REDO": "0"}, lambda st: st["rerun_level"] > 0),                           # no second speculative pass
    ({"ZOPFLI_AMD_SEG_CUTS": "0"}, lambda st: st["accepted"] > 0),                              # every task warms up over 512 positions (no cut points)
    ({"ZOPFLI_AMD_SEG_CUTS": "64", "ZOPFLI_AMD_SEG_L": "512", "ZOPFLI_AMD_SEG_HEAD": "2048"}, lambda st: st["tasks"] > 1000 and st["accepted"] > 0),   # short tasks, cut points sought close by
    ({"ZOPFLI_AMD_MATCH_FILTER": "0"}, lambda st: st["accepted"] > 0),                          # k_match2 with the one-byte candidate test
    ({"ZOPFLI_AMD_SEG_MID": "0"}, lambda st: st["accepted"] > 0),
    ({"ZOPFLI_AMD_SHORTCUT_CHAIN": "0"}, lambda st: st["accepted"] > 0),
    ({"ZOPFLI_AMD_RUN_CODES": "1"}, lambda st: st["accepted"] > 0),                             # codes for wide run rows too (round 5's layout)                        # long-run shortcuts window by window (no chain in a fixed frame)                               # no mid snapshots: a task that leaves its binade is re-run whole
]


def _has_experiments():
    """-DZMX_EXPERIMENTS builds (tools/build_variant.py exp -DZMX_EXPERIMENTS, ZOPFLI_AMD_LIB=...) carry the kernels that
    lost their measurement; the shipped library does not (zmx_has_experiments)."""
    try:
        return bool(api.library().zmx_has_experiments())
    except Exception:
        return False


if _has_experiments():
    CHAIN_ENVS.append(({"ZOPFLI_AMD_COOP": "1"}, lambda st: st["accepted"] > 0))   # run tasks by four waves each (zmx_dp6.h: exact, 10 % slower)


@pytest.mark.parametrize("env,expect", CHAIN_ENVS, ids=lambda v: "-".join(f"{k[15:]}{x}" for k, x in v.items()) if isinstance(v, dict) else "")
def test_chain_task_paths(env, expect):
    """GetBestLengths cut into verified tasks (zmx_dp4.h): whatever the task geometry and however wrong
    the guessed levels, length_array and the stores equal the oracle's on every class — and the
    statistics show that the path under test (accept, re-run from the true state, serial) really ran."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(__file__), "seg_probe.py")
    r = subprocess.run([sys.executable, probe], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert expect(st), st


@pytest.mark.parametrize("int_path", ["1", "0"])
def test_chain_tie_rule(int_path):
    """The exactness branch entropy costs never reach: a cost model of dyadic weights (seg_probe.dyadic_costs)
    ties in the float rounding of the binades 2^13 .. 2^21 (squeeze.c:281-299: float cells, double sums), so a
    task there must not be accepted on a shifted entry state (zmx_dp4.h d4_accept -> 2) nor take the integer
    chain step (zmx_dp5.h: the workgroup's table is refused when the tie mask has its binade).  length_array
    and the stores equal the oracle's, and the statistics show tasks re-run for the tie rule."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(__file__), "seg_probe.py")
    env = dict(os.environ, SEG_PROBE_COSTS="dyadic", ZOPFLI_AMD_INT_PATH=int_path)
    r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["rerun_tie"] > 0 and st["tasks"] > 100, st


def test_guard_mode():
    """ZOPFLI_AMD_GUARD=1 (red zones around every device allocation, poisoned bodies, a check after every kernel
    launch: zmx_hip.hip): the chained squeeze runs of seg_probe.py on every class and a block-split stream stay
    bit-exact and touch no red zone; a byte broken on purpose (ZOPFLI_AMD_GUARD_SELFTEST) is reported with the
    allocation it lies behind."""
    import subprocess
    import sys
    probe = os.path.join(os.path.dirname(__file__), "seg_probe.py")
    env = dict(os.environ, ZOPFLI_AMD_GUARD="1")
    r = subprocess.run([sys.executable, probe], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    code = ("import sys, gzip; sys.path.insert(0, %r)\n"
            "from zopfli_amd import ZopfliOptions, api, generate\n"
            "d = generate('M', 2500000)\n"
            "o = api.compress(d, 0, ZopfliOptions(5, 1, 15))\n"
            "assert gzip.decompress(o) == d\n"
            "import hashlib; print(hashlib.sha256(o).hexdigest())\n" % os.path.dirname(os.path.dirname(__file__)))
    sha = {}
    for guard in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZOPFLI_AMD_GUARD=guard), capture_output=True,
                           text=True, timeout=1800)
        assert r.returncode == 0, (guard, r.stdout[-2000:], r.stderr[-2000:])
        sha[guard] = r.stdout.split()[-1]
    assert sha["0"] == sha["1"]
    r = subprocess.run([sys.executable, probe], env=dict(env, ZOPFLI_AMD_GUARD_SELFTEST="7", SEG_PROBE_CASES="X"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "red zone" in (r.stdout + r.stderr), (r.stdout[-1000:], r.stderr[-1000:])


def _golden(lo, hi):
    out = []
    for name in ("vectors.json", "vectors_extra.json"):
        path = os.path.join(os.path.dirname(GOLDEN), name)
        if os.path.exists(path):
            with open(path) as f:
                out += [c for c in json.load(f) if lo <= c["insize"] <= hi]
    return out


def _input(spec):
    if spec["kind"] == "literal":
        from golden.make_golden import LITERALS
        return LITERALS[spec["name"]]
    return generate(spec["cls"], spec["size"], spec.get("seed"))


def _gid(c):
    return (f"{c['input'].get('name', c['input'].get('cls'))}-{c['insize']}-f{c['format']}-n{c['numiterations']}"
            f"-s{c['blocksplitting']}-m{c['blocksplittingmax']}")


@pytest.mark.parametrize("case", _golden(0, 4000000), ids=_gid)
def test_stream_golden(gpu_lib, case):
    """ZopfliCompress through libzopfli_amd.so == the reference's bytes (SHA-256 of the output of
    oracle/_ref, committed in tests/golden/vectors.json)."""
    data = _input(case["input"])
    opt = ZopfliOptions(case["numiterations"], case["blocksplitting"], case["blocksplittingmax"])
    out = api.compress(data, case["format"], opt, lib=gpu_lib)
    if case["format"] == 0:
        assert gzip.decompress(out) == data
    assert len(out) == case["outsize"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


@pytest.mark.parametrize("btype", [0, 1, 2])
def test_deflate_part_vs_reference(gpu_lib, btype):
    """ZopfliDeflatePart with a dictionary, forced block types (deflate.c:811-842)."""
    if not ol.have_ref():
        pytest.skip("oracle/_ref not shipped")
    data = generate("M", 90000)
    a = api.deflate_part(data, 40000, 90000, btype, 1, ZopfliOptions(5), lib=gpu_lib)
    b = ol.ref_deflate_part(data, 40000, 90000, btype, 1, 5)
    assert a == b


def _big_cases():
    out = []
    for name in ("vectors_big.json", "vectors_big2.json", "vectors_big3.json"):
        path = os.path.join(os.path.dirname(GOLDEN), name)
        if os.path.exists(path):
            with open(path) as f:
                out += [c for c in json.load(f) if c["insize"] == 20000000]
    return out


@pytest.mark.parametrize("case", _big_cases(), ids=_gid)
def test_full_size_round_trip(gpu_lib, case):
    """The bench shapes at 20 MB (20 master blocks, numiterations 15) on every class of data — text-like, markup-like,
    mixed, long runs, two-symbol, PNG-like, random — with and without block splitting (BASELINE configs 2 and 3),
    and the mixed corpus at numiterations 50 (configs[3]'s shape): round trip through zlib and the reference's
    SHA-256 (tests/golden/vectors_big*.json, written by make_golden.py --big / --big2 / --big3)."""
    data = _input(case["input"])
    opt = ZopfliOptions(case["numiterations"], case["blocksplitting"], case["blocksplittingmax"])
    out = api.compress(data, case["format"], opt, lib=gpu_lib)
    assert gzip.decompress(out) == data
    assert len(out) == case["outsize"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


def _at_size_cases():
    """BASELINE's own sizes: every class at 100 MB (configs[1] / [2] shapes, blocksplitting 0 and 1) and the mixed corpus
    at 200 MB with numiterations 50 (configs[3]); goldens from the real reference (make_golden.py --big / --big2 /
    --big3, make_golden_parallel.py)."""
    out = []
    for name in ("vectors_big.json", "vectors_big2.json", "vectors_big3.json", "vectors_big4.json"):
        path = os.path.join(os.path.dirname(GOLDEN), name)
        if os.path.exists(path):
            with open(path) as f:
                out += [c for c in json.load(f) if c["insize"] >= 100000000]
    return out


@pytest.mark.parametrize("case", _at_size_cases(), ids=_gid)
def test_at_size_streams(gpu_lib, case):
    """The workloads BASELINE.json names at the sizes it names — 100 MB of every class with and without block splitting,
    200 MB of the mixed corpus at numiterations 50 — are the reference's streams: SHA-256 and length of the whole
    gzip file (the round trip through zlib is implied by the CRC-32 / ISIZE trailer the digest covers, and checked
    at 20 MB by test_full_size_round_trip)."""
    data = _input(case["input"])
    opt = ZopfliOptions(case["numiterations"], case["blocksplitting"], case["blocksplittingmax"])
    out = api.compress(data, case["format"], opt, lib=gpu_lib)
    assert len(out) == case["outsize"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


def _fixed_codes():
    """The fixed tree (deflate.c:343-349) as zmx_encode_blocks wants it: bit-reversed canonical code | length << 16."""
    ll_len = [8] * 144 + [9] * 112 + [7] * 24 + [8] * 8
    d_len = [5] * 32

    def canon(lens):
        maxl = max(lens)
        cnt = [0] * (maxl + 1)
        for n in lens:
            cnt[n] += 1
        cnt[0] = 0
        nxt, code = [0] * (maxl + 1), 0
        for b in range(1, maxl + 1):
            code = (code + cnt[b - 1]) << 1
            nxt[b] = code
        out = []
        for n in lens:
            c = nxt[n]
            nxt[n] += 1
            out.append(int(format(c, "0%db" % n)[::-1], 2) | (n << 16))
        return out
    return np.array(canon(ll_len) + canon(d_len), dtype=np.uint32), ll_len, d_len


def _py_symbol_bits(ll, dd, codes, bit_start):
    """AddLZ77Data + end symbol (deflate.c:297-333) in plain Python: (bytes, nbits)."""
    acc, pos = 0, bit_start
    codes = [int(c) for c in codes]

    def put(v, n):
        nonlocal acc, pos
        acc |= (int(v) & ((1 << n) - 1)) << pos
        pos += n
    for litlen, dist in zip(ll.tolist(), dd.tolist()):
        if dist == 0:
            put(codes[litlen] & 0xffff, codes[litlen] >> 16)
            continue
        if litlen < 11:
            ls, le = 254 + litlen, 0
        elif litlen == 258:
            ls, le = 285, 0
        else:
            le = (litlen - 3).bit_length() - 1 - 2
            ls = 261 + 4 * le + (((litlen - 3) >> le) & 3)
        if dist < 5:
            ds, de = dist - 1, 0
        else:
            lg = (dist - 1).bit_length() - 1
            ds, de = 2 * lg + (((dist - 1) >> (lg - 1)) & 1), lg - 1
        put(codes[ls] & 0xffff, codes[ls] >> 16)
        put((litlen - 3) & ((1 << le) - 1), le)
        put(codes[288 + ds] & 0xffff, codes[288 + ds] >> 16)
        put((dist - 1) & ((1 << de) - 1), de)
    put(codes[256] & 0xffff, codes[256] >> 16)
    return acc.to_bytes((pos + 7) // 8, "little"), pos - bit_start


@pytest.mark.parametrize("cls", ["T", "M", "Z"])
def test_device_bit_writer(gpu_ctx, cls):
    """zmx_encode_blocks (deflate.c:297-333 + the end symbol on the device) against a bit-by-bit Python writer:
    greedy stores of blocks of ragged sizes (symbol counts around the 2048-symbol tiles), header offsets 0 / 3 / 77,
    both with one code table per job.  The host-side use of it is covered by every stream test."""
    data = generate(cls, 300000)
    gpu_ctx.set_input(data)
    blocks = [(0, 1), (1, 2500), (2500, 100000), (100000, 100007), (100007, 300000)]
    t = gpu_ctx.build_tables(blocks)
    nsym, _ = t.greedy(0)
    codes, _, _ = _fixed_codes()
    starts = [0, 3, 77, 31, 64]
    want, jobs = [], []
    for b in range(len(blocks)):
        ll, dd = t.store(b, 0, nsym[b])
        by, nb = _py_symbol_bits(ll, dd, codes, starts[b])
        want.append(by)
        jobs.append((b, 0, int(nsym[b]), starts[b], nb))
    got = t.encode_blocks(jobs, np.tile(codes, (len(jobs), 1)))
    for b in range(len(blocks)):
        assert got[b] == want[b], f"block {b}: {len(got[b])} bytes against {len(want[b])}"
    # a wrong expectation is reported, not written
    bad = list(jobs[2])
    bad[4] += 1
    with pytest.raises(RuntimeError):
        t.encode_blocks([tuple(bad)], codes)
    t.free()


def test_device_checksums(gpu_ctx, gpu_lib):
    """k_checksum: CRC-32 and Adler-32 of ranges of the resident input against zlib's, around
    every boundary of the kernel's geometry (4-byte body loads, 1 KiB lanes, 256 KiB pieces), at the bench's
    size, and through the containers (gzip_container.c:75,107-110; zlib_container.c:29,71-74)."""
    from test_cpu_host_stream import _checksum_cases
    data = generate("X", 300000) + generate("R", 300000) + bytes(200000) + b"\xff" * 70000
    gpu_ctx.set_input(data)
    for a, b in _checksum_cases(len(data)):
        assert gpu_ctx.checksum(api.CRC32, a, b) == zlib.crc32(data[a:b]), (a, b)
        assert gpu_ctx.checksum(api.ADLER32, a, b) == zlib.adler32(data[a:b]), (a, b)
    with pytest.raises(RuntimeError):
        gpu_ctx.checksum(api.CRC32, 0, len(data) + 1)
    big = generate("T", 100000000)
    gpu_ctx.set_input(big)
    assert gpu_ctx.checksum(api.CRC32, 0, len(big)) == zlib.crc32(big)
    assert gpu_ctx.checksum(api.ADLER32, 0, len(big)) == zlib.adler32(big)
    assert gpu_ctx.checksum(api.CRC32, 32768, len(big) - 5) == zlib.crc32(big[32768:-5])
    small = generate("M", 70001)
    z = api.compress(small, 1, ZopfliOptions(1), lib=gpu_lib)
    assert z[-4:] == zlib.adler32(small).to_bytes(4, "big") and zlib.decompress(z) == small
    g = api.compress(small, 0, ZopfliOptions(1), lib=gpu_lib)
    assert g[-8:-4] == zlib.crc32(small).to_bytes(4, "little") and gzip.decompress(g) == small
    for empty_fmt, tail in ((0, bytes(8)), (1, (1).to_bytes(4, "big"))):
        e = api.compress(b"", empty_fmt, ZopfliOptions(1), lib=gpu_lib)
        assert e.endswith(tail)


def test_verify_pass(gpu_ctx):
    """ZopfliVerifyLenDist on the device (lz77.c:270-295): a greedy and an optimal parse pass; the same symbols read
    against another input, or one symbol short, do not."""
    data = generate("M", 200000)
    gpu_ctx.set_input(data)
    blocks = [(0, 70000), (70000, 200000)]
    t = gpu_ctx.build_tables(blocks)
    nsym, hist = t.greedy(0)
    t.verify_stores([0, 1], [0, 0], nsym)
    with pytest.raises(RuntimeError, match="add up"):
        t.verify_stores([1], [0], [int(nsym[1]) - 1])
    cost = np.zeros((2, 320))
    mincost = np.zeros(2)
    for b in range(2):
        ll, d = ol.entropy_costs(hist[b])
        cost[b, :288], cost[b, 288:] = ll, d
        mincost[b] = ol.model_min_cost(ll, d)
    nsym2, _ = t.squeeze_run(cost, mincost, np.ones(2, dtype=np.int32))
    t.verify_stores([0, 1], [1, 1], nsym2)
    t.free()
    # the library's own use of it: same bytes with the pass switched on
    import subprocess
    import sys
    code = ("import sys, hashlib\nsys.path.insert(0, %r)\nfrom zopfli_amd import ZopfliOptions, api, generate\n"
            "print(hashlib.sha256(api.compress(generate('M', 1200000), 0, ZopfliOptions(3))).hexdigest())\n"
            % os.path.dirname(os.path.dirname(__file__)))
    outs = []
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZOPFLI_AMD_VERIFY=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1]


def _part_cases():
    path = os.path.join(os.path.dirname(GOLDEN), "vectors_part.json")
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _part_cases(), ids=lambda c: "part-%d" % (c["inend"] - c["instart"]))
def test_deflate_part_one_20mb_block(gpu_lib, case):
    """ZopfliDeflatePart with blocksplitting off over 20.1 MB = ONE deflate block with a dictionary before it
    (deflate.c:811-842): 32-bit DP row offsets stopped round 1 at 16 MB per block.  Against the reference's
    SHA-256 (tests/golden/vectors_part.json, make_golden.py --part)."""
    data = _input(case["input"])
    opt = ZopfliOptions(case["numiterations"], case["blocksplitting"], 15)
    out, bp = api.deflate_part(data, case["instart"], case["inend"], case["btype"], case["final"], opt, lib=gpu_lib)
    assert len(out) == case["outsize"] and bp == case["bp"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]
    d = zlib.decompressobj(-15, zdict=data[case["instart"] - 32768:case["instart"]])
    assert d.decompress(out) == data[case["instart"]:case["inend"]]


def test_code_budget_smaller_batches():
    """The DP edges of a batch (two bytes each, k_codes) are capped by ZOPFLI_AMD_CODE_BUDGET_MB; beyond it
    the table build tells the host to come back with fewer master blocks (api.cc RunParts).  20 MB of
    text needs ~0.4 GB of codes: with a 100 MB budget the request runs in several batches and must still
    produce the reference's bytes."""
    import subprocess
    import sys
    code = (
        "import hashlib, json, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "out = api.compress(generate('T', 20000000), 0, ZopfliOptions(2, 0))\n"
        "print(hashlib.sha256(out).hexdigest(), len(out))\n" % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for budget in ("100", "98304"):
        env = dict(os.environ, ZOPFLI_AMD_CODE_BUDGET_MB=budget)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[budget] = r.stdout.strip().split()
    assert res["100"] == res["98304"]


def test_reference_cli_linked_against_libzopfli_amd(tmp_path):
    """the reference's own zopfli_bin.c, linked with -lzopfli_amd, writes the
    same .gz / .zlib / .deflate files as the real reference library produces."""
    import subprocess
    from zopfli_amd._build import REF_CLI
    if not os.path.exists(REF_CLI):
        pytest.skip("tests/_build/zopfli_ref_cli_amd not built (needs /root/reference at build time)")
    data = generate("X", 300000, 5)
    src = tmp_path / "input.xml"
    src.write_bytes(data)
    for flag, ext, fmt in (("--gzip", ".gz", 0), ("--zlib", ".zlib", 1), ("--deflate", ".deflate", 2)):
        r = subprocess.run([REF_CLI, flag, "--i5", str(src)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = (tmp_path / ("input.xml" + ext)).read_bytes()
        if fmt == 0:
            assert gzip.decompress(out) == data
        if ol.have_ref():
            assert out == ol.ref_compress(data, fmt, 5)


def test_dynamic_user_of_libzopfli_so_1_switches_by_library_path(tmp_path):
    """an existing user of libzopfli.so.1 — the reference's CLI linked against the reference's
    own shared library, no rpath — gets this library by LD_LIBRARY_PATH alone (same soname) and writes
    the same file."""
    import subprocess
    from zopfli_amd._build import LIB_SONAME, REF_CLI_DYN, REF_SO_DIR
    if not (os.path.exists(REF_CLI_DYN) and os.path.exists(LIB_SONAME)):
        pytest.skip("tests/_build/zopfli_ref_cli_dyn not built (needs /root/reference at build time)")
    data = generate("M", 250000, 3)
    outs = {}
    for name, libdir in (("ref", REF_SO_DIR), ("amd", os.path.dirname(LIB_SONAME))):
        src = tmp_path / (name + ".bin")
        src.write_bytes(data)
        env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([REF_CLI_DYN, "--i5", str(src)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = (tmp_path / (name + ".bin.gz")).read_bytes()
        # which library did the loader pick?
        ldd = subprocess.run(["ldd", REF_CLI_DYN], env=env, capture_output=True, text=True).stdout
        assert libdir in ldd, ldd
    assert outs["amd"] == outs["ref"]
    assert gzip.decompress(outs["amd"]) == data


def test_small_batches_and_threads():
    """Device batches smaller than the request (ZOPFLI_AMD_PARTS_PER_BATCH=2 on 5 master blocks) and
    two caller threads at once (the shared context serialises them) give the bytes of the default run."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys, threading\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "data = [generate('M', 4300000), generate('X', 1200000, 3)]\n"
        "out = [None, None]\n"
        "def run(i): out[i] = api.compress(data[i], 0, ZopfliOptions(2))\n"
        "ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]\n"
        "[t.start() for t in ts]; [t.join() for t in ts]\n"
        "print(' '.join(hashlib.sha256(o).hexdigest() for o in out))\n" % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for batch in ("2", "256"):
        env = dict(os.environ, ZOPFLI_AMD_PARTS_PER_BATCH=batch)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[batch] = r.stdout.strip().split()
    assert res["2"] == res["256"]
    if ol.have_ref():
        assert res["2"][0] == hashlib.sha256(ol.ref_compress(generate("M", 4300000), 0, 2)).hexdigest()


def test_master_blocks_dealt_over_device_contexts():
    """The multi-device path of the C entry points (api.cc RunPartsSharded) on a one-GPU box: three
    contexts on device 0 share the master blocks of one ZopfliCompress call; the stream equals the
    one-context stream (and the reference's)."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "data = generate('M', 4300000) + generate('R', 700000)\n"
        "print(hashlib.sha256(api.compress(data, 0, ZopfliOptions(3))).hexdigest())\n"
        "print(hashlib.sha256(api.deflate_part(data, 1500000, 2600000, 2, 1, ZopfliOptions(3))[0]).hexdigest())\n"
        % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for devs in ("0", "0,0,0"):
        env = dict(os.environ, ZOPFLI_AMD_DEVICES=devs)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[devs] = r.stdout.split()
    assert res["0"] == res["0,0,0"]
    if ol.have_ref():
        data = generate("M", 4300000) + generate("R", 700000)
        assert res["0"][0] == hashlib.sha256(ol.ref_compress(data, 0, 3)).hexdigest()


def test_mid_size_calls_dealt_at_stream_priorities():
    """Round 5's dealing rules (api.cc RunPartsShardedOnce): a call of 4 master blocks or more with block splitting
    — and data with long runs of equal bytes with or without it — goes over three contexts of the device at three
    stream priorities (ZOPFLI_AMD_DEAL_AFTER=0: from the process's first such call on, not its eighth).  The call trace
    shows the three shards, and the streams equal those of a process that never deals (ZOPFLI_AMD_SPLIT_MB=0) and
    the reference's."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "data = generate('T', 2100000) + generate('Z', 1900000, 5) + generate('P', 1300000)\n"
        "o = ZopfliOptions(3)\n"
        "print(hashlib.sha256(api.compress(data, 0, o)).hexdigest())\n"
        "o.blocksplitting = 0\n"
        "print(hashlib.sha256(api.compress(data * 4, 0, o)).hexdigest())\n"
        % os.path.dirname(os.path.dirname(__file__)))
    res = {}
    for name, extra in (("dealt", {"ZOPFLI_AMD_DEAL_AFTER": "0", "ZOPFLI_AMD_TRACE_CALL": "1"}), ("never", {"ZOPFLI_AMD_SPLIT_MB": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = r.stdout.split()
        if name == "dealt":
            assert r.stderr.count("shard 2 (") == 2, r.stderr[-3000:]      # both calls ran as three shards
    assert res["dealt"] == res["never"]
    if ol.have_ref():
        data = generate("T", 2100000) + generate("Z", 1900000, 5) + generate("P", 1300000)
        assert res["dealt"][0] == hashlib.sha256(ol.ref_compress(data, 0, 3)).hexdigest()


def test_rccl_gather_world_of_one(gpu_ctx):
    """zmx_dist_* (dist.cc): librccl loads, a communicator of one rank forms on the device and the
    gather returns rank 0's own blob — all of the RCCL path a one-GPU box can run; with more ranks the
    same calls move the other ranks' blobs (tests/test_cpu_sharding.py covers the sharding and the merge)."""
    from zopfli_amd import Dist
    uid = Dist.unique_id(gpu_ctx.lib)
    assert len(uid) == 128
    d = Dist(gpu_ctx, 0, 1, uid)
    try:
        blob = bytes(range(256)) * 1000
        parts = d.gather(blob)
        assert len(parts) == 1 and parts[0].tobytes() == blob
        assert d.gather(b"")[0].size == 0
        assert d.comm_count() == 1          # ncclCommCount: what bench.py checks against --gpus at N > 1
    finally:
        d.close()


def test_bench_gpus_2_as_typed():
    """`python bench.py --gpus 2 ...` as the driver types it, with no launcher around it: bench.py starts the two ranks
    itself (torch.distributed.run on 127.0.0.1); here both ranks share device 0 and gather over gloo.  The headline is
    BASELINE configs[2]'s shape — ONE stream, the reference's default block splitting, master blocks sharded (strong) —
    with the weak line riding along, and the sharded stream must round-trip."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device-index", "0",
                        "--size", "4000000", "--steps", "1", "--warmup", "0", "--cpu-sample", "200000", "--devices", "0,0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["roundtrip_ok"] is True
    assert "blocksplitting=1" in d["config"]["workload"] and d["config"]["total_bytes"] == 4000000
    assert d["roofline"] and d["cpu_baseline"]["cores"] == 1
    assert d["weak"]["scaling"] == "weak" and d["weak"]["config"]["total_bytes"] == 8000000 and d["weak"]["roundtrip_ok"] is True
    assert d["in_process"].get("same_stream_as_gathered") is True, d["in_process"]


def _write_png(path, width, height, seed):
    """A valid RGBA8 PNG (filter 0, zlib level 6) of a smooth gradient plus noise — no imaging library."""
    import struct
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    img = np.stack([(x * 255 // max(width - 1, 1)), (y * 255 // max(height - 1, 1)), ((x + y) // 3 % 256),
                    np.full_like(x, 255)], axis=-1).astype(np.int32)
    img[..., :3] += rng.integers(-3, 4, size=(height, width, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(height))

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(png)


def test_zopflipng_linked_against_libzopfli_amd(tmp_path):
    """BASELINE config 5 plumbing at 384x256: the reference's zopflipng linked with -lzopfli_amd writes the
    same PNG as the all-reference build (CustomPNGDeflate -> ZopfliDeflate, zopflipng_lib.cc:47-66)."""
    import subprocess
    from zopfli_amd._build import PNG_AMD, PNG_REF
    if not (os.path.exists(PNG_AMD) and os.path.exists(PNG_REF)):
        pytest.skip("tests/_build/zopflipng_* not built (needs /root/reference at build time)")
    src = str(tmp_path / "in.png")
    _write_png(src, 384, 256, 7)
    outs = {}
    for name, exe in (("amd", PNG_AMD), ("ref", PNG_REF)):
        dst = str(tmp_path / (name + ".png"))
        r = subprocess.run([exe, "-y", "--iterations=5", "--filters=0p", src, dst], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, (name, r.stdout[-1000:], r.stderr[-1000:])
        with open(dst, "rb") as f:
            outs[name] = f.read()
    assert outs["amd"] == outs["ref"]
    assert len(outs["amd"]) < os.path.getsize(src)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_run_paths_fuzz(seed):
    """tools/fuzz_runs.py in small: runs of equal bytes with lengths around 32 / 64 / 258 / 516 / 774 / 1024, back to
    back, cut by block ends, with text and noise between them — the stretches of the run variant of the chain
    (zmx_dp5.h: run_stretch, other_stretch, the shortcut) against the real reference, byte for byte."""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_runs
    rng = random.Random(seed)
    for size, n, bs in ((70000, 15, 1), (300000, 3, 0), (1000001, 3, 1), (300000, 15, 0)):
        data = fuzz_runs.make_case(rng, size)
        ref = ol.ref_compress(data, 0, n, bs, 15)
        mine = api.compress(data, 0, ZopfliOptions(n, bs, 15))
        assert mine == ref, f"seed {seed}: {size} bytes, numiterations {n}, blocksplitting {bs}"


def _ref_greedy_store(data, n):
    """The REAL reference's greedy parse of data[0:n) as its own ZopfliLZ77Store (kept alive for its block-size functions)."""
    import ctypes
    lib = ol.ref()
    h = ol.RefHash()
    lib.ZopfliAllocHash(32768, ctypes.byref(h))
    o = ol.RefOptions(0, 0, 15, 1, 0, 15)
    s = ol.RefBlockState()
    lib.ZopfliInitBlockState(ctypes.byref(o), 0, n, 0, ctypes.byref(s))
    st = ol.RefStore()
    lib.ZopfliInitLZ77Store(data, ctypes.byref(st))
    lib.ZopfliLZ77Greedy(ctypes.byref(s), data, 0, n, ctypes.byref(st), ctypes.byref(h))
    lib.ZopfliCleanBlockState(ctypes.byref(s))
    lib.ZopfliCleanHash(ctypes.byref(h))
    return st


BLOCK_COST_CASES = [("T", 400000), ("X", 300000), ("R", 300000), ("P", 300000), ("M", 300000), ("Z", 60000), ("B", 12000),
                    ("T", 2600), ("R", 990), ("T", 9)]


@pytest.mark.gpu
def test_block_costs(gpu_ctx):
    """k_block_cost == ZopfliCalculateBlockSizeAutoType (deflate.c:610-621) of the REAL reference on its own greedy stores:
    every class, whole stores, single symbols, ranges that end on and around the samples of the prefix histograms, ranges
    below the direct-count threshold, and stores on either side of the 1000-symbol fixed-tree switch (deflate.c:615)."""
    import ctypes
    if not ol.have_ref():
        pytest.skip("oracle/_ref is not built")
    lib = ol.ref()
    stores, refs, keep_alive = [], [], []
    for i, (cls, n) in enumerate(BLOCK_COST_CASES):
        data = generate(cls, n, seed=40 + i)
        keep_alive.append(data)
        st = _ref_greedy_store(data, n)
        refs.append(st)
        m = st.size
        stores.append((np.ctypeslib.as_array(st.litlens, shape=(m,)).copy(), np.ctypeslib.as_array(st.dists, shape=(m,)).copy()))
    cs = api.CostStores.from_host(gpu_ctx, stores)
    try:
        rng = np.random.default_rng(5)
        ranges = []
        for q, (ll, _) in enumerate(stores):
            m = len(ll)
            fixed = [(0, m), (0, 1), (m - 1, m), (0, min(m, 300)), (0, min(m, 511)), (0, min(m, 512)), (0, min(m, 513)),
                     (min(255, m - 1), m), (min(256, m - 1), m), (min(257, m - 1), m), (m // 2, m // 2 + 1)]
            for k in (256, 512, 1024, 4096, 65536):
                if k < m:
                    fixed += [(0, k), (k, m), (k - 1, m), (k + 1, m), (1, k + 1)]
            for a, b in fixed:
                if a < b:
                    ranges.append((q, a, b))
            for _ in range(120):
                a, b = sorted(rng.integers(0, m + 1, 2).tolist())
                if a < b:
                    ranges.append((q, a, b))
            for _ in range(40):      # short ranges: the direct count
                a = int(rng.integers(0, m))
                b = min(m, a + 1 + int(rng.integers(0, 700)))
                ranges.append((q, a, b))
        got = cs.block_costs(ranges)
        for (q, a, b), g in zip(ranges, got.tolist()):
            want = lib.ZopfliCalculateBlockSizeAutoType(ctypes.byref(refs[q]), a, b)
            assert g == want, (BLOCK_COST_CASES[q], a, b, g, want)
    finally:
        cs.free()
        for st in refs:
            lib.ZopfliCleanLZ77Store(ctypes.byref(st))


@pytest.mark.gpu
def test_block_costs_from_device_stores(gpu_ctx):
    """zmx_cost_stores_create: sequences glued from the device's own stores (a master block's greedy store; the stores of
    several blocks one after the other) price like the same symbols uploaded from the host."""
    n = 300000
    data = generate("M", n, seed=9)
    gpu_ctx.set_input(data)
    blocks = [(0, 100000), (100000, 180000), (180000, 300000)]
    t = gpu_ctx.build_tables(blocks, matches_only=True)
    try:
        nsym, _ = t.greedy(0)
        host = [t.store(b, 0, nsym[b]) for b in range(3)]
        joined = (np.concatenate([h[0] for h in host]), np.concatenate([h[1] for h in host]))
        dev = api.CostStores.from_tables(t, [[(0, 0, nsym[0])], [(b, 0, nsym[b]) for b in range(3)], [(2, 0, nsym[2]), (1, 0, nsym[1])]])
        up = api.CostStores.from_host(gpu_ctx, [host[0], joined, (np.concatenate([host[2][0], host[1][0]]), np.concatenate([host[2][1], host[1][1]]))])
        try:
            rng = np.random.default_rng(2)
            ranges = []
            for q in range(3):
                m = dev.sizes[q]
                assert m == up.sizes[q]
                ranges += [(q, 0, m)] + [(q,) + tuple(sorted(rng.integers(0, m + 1, 2).tolist())) for _ in range(60)]
            ranges = [r for r in ranges if r[1] < r[2]]
            assert np.array_equal(dev.block_costs(ranges), up.block_costs(ranges))
            # zmx_cost_positions == ZopfliLZ77GetByteRange(0, index) (lz77.c:160-166): the running sum of the symbols' lengths
            span = np.where(joined[1] == 0, 1, joined[0]).astype(np.uint64)
            csum = np.concatenate([[0], np.cumsum(span)])
            idx = [0, 1, 1023, 1024, 1025, 2048, dev.sizes[1] - 1, dev.sizes[1]] + rng.integers(0, dev.sizes[1] + 1, 40).tolist()
            got = dev.positions([(1, int(i)) for i in idx])
            assert got.tolist() == [int(csum[int(i)]) for i in idx]
        finally:
            dev.free()
            up.free()
    finally:
        t.free()


@pytest.mark.gpu
def test_split_search_on_the_device_fuzz():
    """f-1 end to end with EVERY round of both split searches on the device (tools/fuzz_split.py: k_block_cost forced from one
    sequence on, no host rounds): streams glued from every class, copies and noise, random options and containers — the
    real reference's bytes."""
    import subprocess
    import sys
    if not ol.have_ref():
        pytest.skip("oracle/_ref is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_split.py"), "10", "3"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FUZZ_SIZES="9000,70000,300000,1000001,2300000"))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "10 of 10 identical" in r.stdout
