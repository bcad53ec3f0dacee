"""The match-table kernels on inputs built to stress what the exact skip-walk (k_levels + k_rank2 + k_match5,
zmx_match5.h) has to get right where the synthetic classes are mild: the 8192-hit cap of lz77.c:527-530 at and around
its boundary, the switch to the second hash chain (lz77.c:509-519) inside and at the end of runs, periods at and beside
258 and 32768, matches at the maximum distance, many equal prefixes of every level length.  Every record of every
position from k_match5 == from k_match2 == from the per-block default (zmx_match_digest), and k_match5 ==
ZopfliFindLongestMatch (the oracle, position by position) on the small cases."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _units(rng, prefix, nrand, count):
    """count units of `prefix` followed by nrand random bytes: every unit starts with the same 3-byte hash."""
    u = np.empty((count, len(prefix) + nrand), dtype=np.uint8)
    u[:, :len(prefix)] = np.frombuffer(prefix, dtype=np.uint8)
    u[:, len(prefix):] = rng.integers(0, 256, size=(count, nrand), dtype=np.uint8)
    return u.reshape(-1)


def _runs(rng, n, lengths, symbols):
    out = np.empty(n, dtype=np.uint8)
    i = 0
    while i < n:
        k = int(rng.choice(lengths))
        out[i:i + k] = rng.choice(symbols)
        i += k
    return out


def _mutated_repeats(rng, n, block, rate):
    base = rng.integers(0, 256, size=block, dtype=np.uint8)
    out = np.tile(base, n // block + 1)[:n].copy()
    hits = rng.integers(0, n, size=max(1, int(n * rate)))
    out[hits] = rng.integers(0, 256, size=len(hits), dtype=np.uint8)
    return out


def _make(name, n):
    rng = np.random.default_rng(sum(map(ord, name)) * 7919)
    if name == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if name.startswith("period"):
        p = int(name[6:])
        return np.tile(rng.integers(0, 256, size=p, dtype=np.uint8), n // p + 1)[:n].copy()
    if name == "cap4":      # 4-byte units, one hash value: 8192 of them in a window — the cap at its boundary
        return _units(rng, b"abc", 1, n // 4 + 1)[:n].copy()
    if name == "cap3mix":   # 3-byte prefix, then 0 - 2 random bytes: more than 8192 hits in a window
        parts = [_units(rng, b"xyz", int(k), 1) for k in rng.integers(0, 3, size=n // 4)]
        return np.concatenate(parts)[:n].copy()
    if name == "cap5":      # 5-byte units: 6553 hits in a window, just under the cap
        return _units(rng, b"abc", 2, n // 5 + 1)[:n].copy()
    if name == "prefix8":   # equal 8-byte prefixes with different tails: level links of every length get used
        return _units(rng, b"prefix-8", 3, n // 11 + 1)[:n].copy()
    if name == "runs":      # runs of few symbols, lengths at and around 258 and the window
        return _runs(rng, n, [1, 2, 3, 257, 258, 259, 300, 516, 517, 5000, 40000], [0, 0, 0, 65, 66])
    if name == "runs2":     # two-byte alternations inside runs: same[] short, second chain busy
        a = _runs(rng, n, [2, 4, 258, 1000, 33000], [0, 1])
        a[::2] ^= (rng.integers(0, 50, size=len(a[::2])) == 0).astype(np.uint8)
        return a
    if name == "mut1k":     # a 1 KB block repeated with rare single-byte changes: long matches at many distances
        return _mutated_repeats(rng, n, 1024, 0.002)
    if name == "mut32k":    # the same at the window size: candidates at distance 32768 exactly
        return _mutated_repeats(rng, n, 32768, 0.0005)
    if name == "bits":      # two symbols, p = 0.5
        return rng.integers(0, 2, size=n, dtype=np.uint8)
    if name == "bits9":     # two symbols, p = 0.9
        return (rng.random(n) < 0.1).astype(np.uint8)
    if name == "mix":
        names = ["cap4", "runs", "bits", "period259", "prefix8", "zeros", "mut1k", "cap3mix", "period32768"]
        k = n // len(names) + 1
        return np.concatenate([_make(m, k) for m in names])[:n].copy()
    raise KeyError(name)


BIG = [("zeros", 600000), ("period2", 300000), ("period3", 300000), ("period258", 400000), ("period259", 400000),
       ("period32768", 400000), ("period32769", 400000), ("cap4", 500000), ("cap3mix", 500000), ("cap5", 500000),
       ("prefix8", 500000), ("runs", 1500000), ("runs2", 800000), ("mut1k", 800000), ("mut32k", 800000),
       ("bits", 300000), ("bits9", 300000), ("mix", 2500000)]


def _blocks(n):
    """One block over everything would hide block-end effects: three blocks, the cuts at odd places, the first
    one short so that the second has less than a window in front of part of it."""
    a, b = 20011, n * 5 // 9 + 3
    return [(0, a), (a, b), (b, n)]


@pytest.mark.parametrize("case", BIG, ids=lambda c: c[0])
def test_match_kernels_agree_on_adversarial_input(gpu_ctx, case):
    name, n = case
    data = _make(name, n).tobytes()
    gpu_ctx.set_input(data)
    dig = {}
    try:
        for kern in (2, 5, 0):
            gpu_ctx.lib.zmx_set_match_kernel(kern)
            t = gpu_ctx.build_tables(_blocks(n), matches_only=True)
            dig[kern] = t.match_digest()
            t.free()
    finally:
        gpu_ctx.lib.zmx_set_match_kernel(0)
    assert dig[2] == dig[5] == dig[0], (name, dig)


SMALL = [("zeros", 70000), ("period258", 70000), ("period32768", 70000), ("cap4", 80000), ("cap3mix", 80000),
         ("prefix8", 60000), ("runs", 120000), ("runs2", 70000), ("mut1k", 60000), ("bits9", 50000), ("mix", 140000)]


@pytest.mark.parametrize("case", SMALL, ids=lambda c: c[0])
def test_skip_walk_vs_oracle_on_adversarial_input(gpu_ctx, case):
    """k_match5 (forced) == ZopfliFindLongestMatch: length, distance and sublen at EVERY position."""
    name, n = case
    data = _make(name, n).tobytes()
    gpu_ctx.set_input(data)
    blocks = [(0, n // 3 + 1), (n // 3 + 1, n)]
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
            assert not bad, f"{name} block {b} [{s},{e}): first mismatches (pos, gpu, oracle) {bad}"
    finally:
        t.free()


STREAMS = [("zeros", 1300000, 5), ("period258", 300000, 5), ("period32769", 300000, 5), ("cap4", 200000, 5),
           ("cap3mix", 200000, 5), ("cap5", 200000, 5), ("prefix8", 300000, 5), ("runs", 1300000, 5),
           ("runs2", 300000, 5), ("mut1k", 300000, 5), ("mut32k", 300000, 5), ("bits", 100000, 5),
           ("bits9", 150000, 5), ("mix", 1200000, 3)]


@pytest.mark.parametrize("case", STREAMS, ids=lambda c: c[0])
def test_adversarial_streams_vs_reference(gpu_lib, case):
    """ZopfliCompress (gzip, the reference's default options but for numiterations) of the same inputs == the reference's
    stream, byte for byte: the whole path — greedy, block split, squeeze runs, the fixed-tree and stored decisions
    (incompressible periods beyond the window come out stored) — on top of those match tables; three of them cross a
    master block boundary."""
    from zopfli_amd import ZopfliOptions, api
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    name, n, iters = case
    data = _make(name, n).tobytes()
    got = api.compress(data, 0, ZopfliOptions(iters), lib=gpu_lib)
    want = ol.ref_compress(data, 0, iters)
    assert got == want, (name, len(got), len(want))


@pytest.mark.parametrize("seed", [5, 6])
def test_match_fuzz(seed):
    """tools/fuzz_match.py: streams glued from those generators, text, copies and noise, through ZopfliCompress with the
    per-block choice of walk, the skip-walk forced and the hit-by-hit walk forced, random options and containers — byte for
    byte the reference's (12 cases a seed here; 120 more were run for profiles/README.md)."""
    import os
    import subprocess
    import sys
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_match.py")
    r = subprocess.run([sys.executable, tool, "12", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
