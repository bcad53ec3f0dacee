"""Fuzz of the run-task paths against the REAL reference (oracle/_ref/libzopfli_ref.so, travels to the GPU box):
inputs made of runs of equal bytes with lengths around every constant the chain's run paths care about (3, 32, 64, 258,
259, 516, 517, 774, 1024 +- a few; long runs of tens of thousands), runs of different bytes back to back, runs cut by
the block end, pieces of text and noise in between — compressed by the library and by the reference with the same
options (numiterations 3 and 15, block splitting on and off); every output must be byte-identical.

    python tools/fuzz_runs.py [cases] [seed]       (GPU box: ~1 s per case of reference time)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from zopfli_amd import ZopfliOptions, api, generate  # noqa: E402

TEXTY = os.environ.get("FUZZ_TEXT", "0") != "0"
# FUZZ_SIZES="4200000,6100000,17000000": calls of several master blocks — with ZOPFLI_AMD_DEAL_AFTER=0 they are dealt over
# three contexts of the device at three stream priorities (api.cc), from 4 master blocks on with block splitting, 16 without
SIZES = [int(x) for x in os.environ.get("FUZZ_SIZES", "70000,300000,1000000,1000001,1200000,2100000").split(",")]
EDGES = [1, 2, 3, 4, 31, 32, 33, 63, 64, 65, 127, 128, 129, 257, 258, 259, 260, 289, 290, 515, 516, 517, 518, 773, 774, 775,
         1023, 1024, 1025, 1031, 1032, 1033, 2047, 2048, 2049]


def make_case(rng, size):
    text = generate("T", 200000, rng.randrange(1 << 30))
    out = bytearray()
    alphabet = bytes(rng.sample(range(256), rng.choice([1, 2, 3, 4, 8])))
    while len(out) < size:
        kind = rng.random()
        if TEXTY:                             # FUZZ_TEXT=1: mostly text and copies, a run now and then
            kind = 0.72 + 0.18 * kind if rng.random() < 0.8 else kind
        if kind < 0.55:                       # a run
            r = rng.random()
            if r < 0.6:
                n = rng.choice(EDGES) + rng.choice([0, 0, 0, 1, -1])
            elif r < 0.9:
                n = rng.randrange(1, 6000)
            else:
                n = rng.randrange(6000, 90000)
            out += bytes([rng.choice(alphabet)]) * max(1, n)
        elif kind < 0.7:                      # the same run pattern again (matches that continue past a run's end)
            if len(out) > 600:
                a = rng.randrange(0, len(out) - 500)
                out += out[a:a + rng.randrange(1, 500)]
        elif kind < 0.9:                      # text
            a = rng.randrange(0, len(text) - 3000)
            out += text[a:a + rng.randrange(1, 3000)]
        else:                                 # noise
            out += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 300)))
    return bytes(out[:size])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    lib = api.library()
    bad = 0
    for k in range(cases):
        size = rng.choice(SIZES)
        data = make_case(rng, size)
        n = rng.choice([3, 3, 15]) if size < 3000000 else rng.choice([1, 2, 3])
        bs = rng.choice([0, 1])
        ref = ol.ref_compress(data, 0, n, bs, 15)
        mine = api.compress(data, 0, ZopfliOptions(n, bs, 15), lib=lib)
        ok = mine == ref
        print(f"case {k}: {size} bytes, numiterations {n}, blocksplitting {bs}: {len(mine)} bytes out, "
              f"{'identical' if ok else 'DIFFERENT from the reference (' + str(len(ref)) + ' bytes)'}", flush=True)
        if not ok:
            bad += 1
            with open(os.path.join(ROOT, "gpurun_out", f"fuzz_runs_bad_{seed}_{k}.bin"), "wb") as f:
                f.write(data)
    print(f"{cases - bad} of {cases} identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
