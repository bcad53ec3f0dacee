"""Generates tests/golden/vectors.json from the REAL reference (oracle/_ref/libzopfli_ref.so,
compiled from /root/reference by oracle/Makefile).  Run in the build container:

    python tests/golden/make_golden.py [--big | --big2 | --big3 | --extra | --part]

Each vector: synthetic class / size / seed (zopfli_amd.datagen) or a literal input, the
ZopfliOptions used, the format, and the SHA-256 + length of the reference's output.
--big adds the bench workloads (20 MB and 100 MB, minutes of CPU each)."""
import hashlib
import json
import multiprocessing as mp
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

LITERALS = {
    "empty": b"",
    "a": b"a",
    "gotest": b"compressthis" + b"_foobar" * 1000 + b"$",  # go/zopfli/zopfli_test.go:37
    "zeros300k": bytes(300000),
}


def make_input(spec):
    from zopfli_amd import generate
    if spec["kind"] == "literal":
        return LITERALS[spec["name"]]
    return generate(spec["cls"], spec["size"], spec.get("seed"))


def run(case):
    import oracle_lib as ol
    data = make_input(case["input"])
    out = ol.ref_compress(data, case["format"], case["numiterations"], case["blocksplitting"],
                          case["blocksplittingmax"])
    case = dict(case)
    case["sha256"] = hashlib.sha256(out).hexdigest()
    case["outsize"] = len(out)
    case["insize"] = len(data)
    return case


def extra_cases():
    """Shapes of the remaining BASELINE configs at sizes the reference finishes in minutes:
    numiterations=50 on a mixed corpus (config 4), zopflipng's deflate call (config 5: raw
    deflate of PNG-like scanlines, 15 / 5 iterations), other formats and split limits."""
    cs = []

    def add(inp, fmt=0, n=15, split=1, smax=15):
        cs.append({"input": inp, "format": fmt, "numiterations": n, "blocksplitting": split,
                   "blocksplittingmax": smax})

    add({"kind": "class", "cls": "M", "size": 2500000}, 0, 50)
    add({"kind": "class", "cls": "T", "size": 1000000}, 0, 50)
    add({"kind": "class", "cls": "P", "size": 1000000}, 2, 15)
    add({"kind": "class", "cls": "P", "size": 1000000}, 2, 5)
    add({"kind": "class", "cls": "T", "size": 1000000}, 1, 15)
    add({"kind": "class", "cls": "X", "size": 1000000}, 0, 15, 1, 5)
    add({"kind": "class", "cls": "M", "size": 2500000}, 0, 5, 1, 0)
    add({"kind": "class", "cls": "T", "size": 300000, "seed": 77}, 0, 1)
    add({"kind": "class", "cls": "B", "size": 100000, "seed": 9}, 0, 15)
    add({"kind": "class", "cls": "Z", "size": 2100000, "seed": 11}, 0, 15)
    return cs


def cases(big):
    cs = []

    def add(inp, fmt=0, n=15, split=1, smax=15):
        cs.append({"input": inp, "format": fmt, "numiterations": n, "blocksplitting": split,
                   "blocksplittingmax": smax})

    for name in LITERALS:
        for fmt in (0, 1, 2):
            add({"kind": "literal", "name": name}, fmt)
    add({"kind": "class", "cls": "T", "size": 65536}, 0, 1)  # BASELINE config 1
    for cls in "TXRZBPM":
        add({"kind": "class", "cls": cls, "size": 65536})
        add({"kind": "class", "cls": cls, "size": 65536}, 2, 5, 0)
    for cls in "TXRZ":
        add({"kind": "class", "cls": cls, "size": 1000000})
        add({"kind": "class", "cls": cls, "size": 1000000}, 0, 15, 0)
    add({"kind": "class", "cls": "B", "size": 200000})
    add({"kind": "class", "cls": "M", "size": 2500000})
    add({"kind": "class", "cls": "T", "size": 4000000})
    add({"kind": "class", "cls": "T", "size": 4000000}, 0, 15, 0)
    if big:
        add({"kind": "class", "cls": "T", "size": 20000000}, 0, 15, 0)
        add({"kind": "class", "cls": "T", "size": 20000000})
        add({"kind": "class", "cls": "T", "size": 100000000}, 0, 15, 0)  # BASELINE config 2
        add({"kind": "class", "cls": "T", "size": 100000000})             # BASELINE config 3
    return cs


def big2_cases():
    """The bench shape on data that is not class T (round-1 verdict): markup-like X and the mixed
    corpus M at 20 MB, numiterations 15, with and without block splitting; X also at 100 MB."""
    cs = []
    for cls in "XM":
        for split in (0, 1):
            cs.append({"input": {"kind": "class", "cls": cls, "size": 20000000}, "format": 0, "numiterations": 15,
                       "blocksplitting": split, "blocksplittingmax": 15})
    for split in (0, 1):
        cs.append({"input": {"kind": "class", "cls": "X", "size": 100000000}, "format": 0, "numiterations": 15,
                   "blocksplitting": split, "blocksplittingmax": 15})
    return cs


def big3_cases():
    """Round-2 verdict item 6: every class at bench size.  The mixed corpus M at 100 MB (bs 0/1), M at 20 MB with
    numiterations=50 (BASELINE configs[3] shape), and R / Z / B / P at 20 MB and 100 MB (class B at 100 MB only
    without block splitting: ~4 h of one core).  Longest first, so a pool drains evenly."""
    cs = []

    def add(cls, size, n=15, split=1):
        cs.append({"input": {"kind": "class", "cls": cls, "size": size}, "format": 0, "numiterations": n,
                   "blocksplitting": split, "blocksplittingmax": 15})

    add("B", 100000000, 15, 0)
    add("M", 20000000, 50, 1)
    for split in (0, 1):
        add("B", 20000000, 15, split)
    for split in (0, 1):
        add("M", 100000000, 15, split)
    for cls in "ZPR":
        for split in (0, 1):
            add(cls, 100000000, 15, split)
    for cls in "ZPR":
        for split in (0, 1):
            add(cls, 20000000, 15, split)
    return cs


def run_to_file(case):
    """run() with the result kept in its own file, so an interrupted --big3 keeps what it finished."""
    i = case["input"]
    name = "%s_%d_n%d_bs%d.json" % (i["cls"], i["size"], case["numiterations"], case["blocksplitting"])
    path = os.path.join(HERE, "_big3_parts", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    done = run(case)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path + ".tmp", "w") as f:
        json.dump(done, f)
    os.replace(path + ".tmp", path)
    return done


def part_cases():
    """ZopfliDeflatePart over ONE block far beyond the reference's 1 MB master blocks (blocksplitting 0: the whole
    range is one deflate block, deflate.c:811-842) — the 32-bit DP row offsets of round 1 stopped at 16 MB."""
    return [dict(input=dict(kind="class", cls="T", size=20500000, seed=None), instart=300000, inend=20400000, btype=2,
                 final=1, numiterations=2, blocksplitting=0)]


def run_part(case):
    import oracle_lib as ol
    data = make_input(case["input"])
    out, bp = ol.ref_deflate_part(data, case["instart"], case["inend"], case["btype"], case["final"],
                                  case["numiterations"], case["blocksplitting"])
    case = dict(case)
    case["sha256"] = hashlib.sha256(out).hexdigest()
    case["outsize"] = len(out)
    case["bp"] = bp
    return case


def main():
    if "--part" in sys.argv:
        done = [run_part(c) for c in part_cases()]
        path = os.path.join(HERE, "vectors_part.json")
        with open(path, "w") as f:
            json.dump(done, f, indent=1)
        print("wrote", path, len(done), "vectors")
        return
    if "--big3" in sys.argv:
        nproc = int(os.environ.get("GOLDEN_PROCS", "6"))
        with mp.Pool(nproc) as pool:
            done = pool.map(run_to_file, big3_cases(), chunksize=1)
        path = os.path.join(HERE, "vectors_big3.json")
        with open(path, "w") as f:
            json.dump(done, f, indent=1)
        print("wrote", path, len(done), "vectors")
        return
    big = "--big" in sys.argv
    extra = "--extra" in sys.argv
    big2 = "--big2" in sys.argv
    path = os.path.join(HERE, "vectors_big2.json" if big2 else "vectors_extra.json" if extra else
                        "vectors_big.json" if big else "vectors.json")
    cs = big2_cases() if big2 else extra_cases() if extra else cases(big)
    if big and not extra and not big2:
        cs = [c for c in cs if c["input"].get("size", 0) >= 20000000]
    with mp.Pool(min(8, len(cs))) as pool:
        done = pool.map(run, cs, chunksize=1)
    with open(path, "w") as f:
        json.dump(done, f, indent=1)
    print("wrote", path, len(done), "vectors")


if __name__ == "__main__":
    main()
