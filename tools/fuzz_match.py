"""Fuzz of the match table's two walks against the REAL reference (oracle/_ref/libzopfli_ref.so, travels to the GPU box):
inputs glued together from the adversarial generators of tests/test_gpu_match_adversarial.py (the hit cap at its
boundary, periods at 258 / 32768, runs, equal prefixes, two symbols, mutated repeats), text and noise, compressed by
the library — with the per-block choice of walk, with the skip-walk forced and with the hit-by-hit walk forced — and
by the reference with the same options (numiterations 1 - 5, block splitting on / off, gzip / zlib / deflate); every
output must be byte-identical.

    python tools/fuzz_match.py [cases] [seed]      (GPU box: 1 - 3 s per case, most of it the reference)"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import test_gpu_match_adversarial as adv  # noqa: E402
from zopfli_amd import ZopfliOptions, api, generate  # noqa: E402

KINDS = ["zeros", "period2", "period3", "period258", "period259", "period32768", "cap4", "cap3mix", "cap5", "prefix8",
         "runs", "runs2", "mut1k", "mut32k", "bits", "bits9"]


def make_case(rng, size):
    out = bytearray()
    while len(out) < size:
        r = rng.random()
        n = rng.choice([200, 3000, 9000, 33000, 40000, 70000])
        if r < 0.6:
            piece = adv._make(rng.choice(KINDS), n).tobytes()
            # (the generators are seeded by name: vary them by a random rotation and an xor of the non-structural bytes)
            k = rng.randrange(len(piece))
            piece = piece[k:] + piece[:k]
        elif r < 0.8:
            piece = generate("T", n, rng.randrange(1 << 30))
        elif r < 0.9:
            piece = bytes(out[max(0, len(out) - n):])               # a copy of what came before: matches at every distance
        else:
            piece = bytes(rng.getrandbits(8) for _ in range(min(n, 2000)))
        out += piece
    return bytes(out[:size])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if not ol.have_ref():
        raise SystemExit("oracle/_ref is not built")
    lib = api.library()
    rng = random.Random(seed)
    bad = 0
    t0 = time.time()
    forced = {0: 0, 2: 0, 5: 0}
    for c in range(cases):
        size = rng.choice([30000, 80000, 150000, 260000, 400000, 1100000 if rng.random() < 0.3 else 120000])
        data = make_case(rng, size)
        iters = rng.choice([1, 2, 3, 5])
        bs = rng.choice([0, 1, 1])
        fmt = rng.choice([0, 0, 1, 2])
        kern = rng.choice([0, 0, 5, 2])
        forced[kern] += 1
        opt = ZopfliOptions(iters)
        opt.blocksplitting = bs
        lib.zmx_set_match_kernel(kern)
        try:
            got = api.compress(data, fmt, opt, lib=lib)
        finally:
            lib.zmx_set_match_kernel(0)
        want = ol.ref_compress(data, fmt, iters, bs)
        if got != want:
            bad += 1
            path = os.path.join(ROOT, "gpurun_out", "fuzz_match_fail_%d_%d.bin" % (seed, c))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "wb") as f:
                f.write(data)
            print("MISMATCH case", c, "size", size, "iters", iters, "bs", bs, "fmt", fmt, "kernel", kern, len(got), len(want), path, flush=True)
    print("fuzz_match: %d cases (seed %d; walks: %d per block, %d skip-walk forced, %d hit by hit), %d mismatches, %.0f s"
          % (cases, seed, forced[0], forced[5], forced[2], bad, time.time() - t0), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
