"""Fuzz of f-1's device path against the REAL reference (oracle/_ref/libzopfli_ref.so, travels to the GPU box): inputs glued
from pieces of every synthetic class, copies and noise, compressed with the reference's default block splitting — every
round of both split searches forced onto the device (k_block_cost: ZOPFLI_AMD_DEVICE_SPLIT=2, from one sequence on, no
host rounds) — and by the reference with the same options; every output must be byte-identical.

    python tools/fuzz_split.py [cases] [seed]       (GPU box: 1 - 3 s per case of reference time)"""
import os
import random
import sys

os.environ.setdefault("ZOPFLI_AMD_DEVICE_SPLIT", "2")
os.environ.setdefault("ZOPFLI_AMD_DEVICE_SPLIT_FROM", "1")
os.environ.setdefault("ZOPFLI_AMD_DEVICE_SPLIT_MIN", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from zopfli_amd import ZopfliOptions, api, generate  # noqa: E402

SIZES = [int(x) for x in os.environ.get("FUZZ_SIZES", "9000,70000,300000,1000000,1000001,2300000,3100000").split(",")]


def make_case(rng, size):
    out = bytearray()
    while len(out) < size:
        kind = rng.random()
        n = rng.choice([rng.randrange(1, 400), rng.randrange(400, 20000), rng.randrange(20000, 400000)])
        if kind < 0.75:
            cls = rng.choice("TTXRRPMBZ")
            if cls in "BZ":
                n = min(n, 30000)       # (the reference is slow on these)
            out += generate(cls, n, rng.randrange(1 << 30))
        elif kind < 0.9 and len(out) > 1000:
            a = rng.randrange(0, len(out) - 900)
            out += out[a:a + rng.randrange(1, 900)]
        else:
            out += bytes(rng.randrange(rng.choice([2, 16, 256])) for _ in range(min(n, 5000)))
    return bytes(out[:size])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    lib = api.library()
    bad = 0
    for i in range(cases):
        size = rng.choice(SIZES)
        data = make_case(rng, size)
        n = rng.choice([1, 1, 2, 3])
        mx = rng.choice([15, 15, 15, 4, 0, 30])
        fmt = rng.choice([0, 1, 2])
        mine = api.compress(data, fmt, ZopfliOptions(n, 1, mx), lib=lib)
        ref = ol.ref_compress(data, fmt, n, 1, mx)
        ok = mine == ref
        bad += 0 if ok else 1
        print(f"case {i}: {size} B, numiterations {n}, blocksplittingmax {mx}, format {fmt}: {len(mine)} / {len(ref)} B {'identical' if ok else 'DIFFERENT'}", flush=True)
    print(f"{cases - bad} of {cases} identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
