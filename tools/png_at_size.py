"""BASELINE config 5 at size: the reference's zopflipng linked against libzopfli_amd.so on a synthetic
W x W RGBA PNG (default W = 4096: 64 MiB raw, deflate input ~67 MB, 5 iterations as zopflipng chooses
for large images), timed, and byte-compared with the all-reference build when --ref is given."""
import json
import os
import struct
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zopfli_amd._build import PNG_AMD, PNG_AMD2, PNG_REF  # noqa: E402


def write_png(path, w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) // 3 % 256),
                    np.full_like(x, 255)], axis=-1).astype(np.int32)
    img[..., :3] += rng.integers(-3, 4, size=(h, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), img.reshape(h, w * 4)], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


GOLDEN = os.path.join(ROOT, "tests", "golden", "png_at_size.json")


def make_golden(widths):
    """The all-reference zopflipng on the synthetic images, here on the CPU (minutes per image at 4096): SHA-256 and size
    of its output go to tests/golden/png_at_size.json, which the timed runs on the GPU box compare with."""
    import hashlib
    have = {}
    if os.path.exists(GOLDEN):
        with open(GOLDEN) as f:
            have = json.load(f)
    tmp = os.environ.get("TMPDIR", "/tmp")
    for w in widths:
        src, dst = os.path.join(tmp, f"gin{w}.png"), os.path.join(tmp, f"gout{w}.png")
        write_png(src, w, w, 7)
        t0 = time.perf_counter()
        r = subprocess.run([PNG_REF, "-y", src, dst], capture_output=True, text=True, timeout=20000)
        assert r.returncode == 0, r.stdout + r.stderr
        with open(dst, "rb") as f:
            out = f.read()
        with open(src, "rb") as f:
            inp = f.read()
        have[str(w)] = {"sha256": hashlib.sha256(out).hexdigest(), "bytes": len(out), "input_sha256": hashlib.sha256(inp).hexdigest(),
                        "reference_seconds_here": round(time.perf_counter() - t0, 1)}
        with open(GOLDEN, "w") as f:
            json.dump(have, f, indent=1)
        print(w, have[str(w)], flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--make-golden":
        make_golden([int(a) for a in sys.argv[2:]])
        return
    w = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
    with_ref = "--ref" in sys.argv
    extra = [a for a in sys.argv[1:] if a.startswith("--") and a not in ("--ref", "--both")]
    tmp = os.environ.get("TMPDIR", "/tmp")
    src = os.path.join(tmp, f"in{w}.png")
    write_png(src, w, w, 7)
    res = {"width": w, "height": w, "input_bytes": os.path.getsize(src), "args": extra}
    outs = {}
    # amd: the reference's zopflipng_lib.cc on libzopfli_amd.so; amd2: libzopflipng_amd.so (trials side by side, row
    # search on the device); ref: the all-reference build
    runs = (("amd2", PNG_AMD2),) + ((("amd", PNG_AMD),) if "--both" in sys.argv else ()) + ((("ref", PNG_REF),) if with_ref else ())
    for name, exe in runs:
        dst = os.path.join(tmp, f"out_{name}{w}.png")
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-y"] + extra + [src, dst], capture_output=True, text=True, timeout=3000)
        res[name + "_seconds"] = round(time.perf_counter() - t0, 2)
        res[name + "_rc"] = r.returncode
        if r.returncode == 0:
            with open(dst, "rb") as f:
                outs[name] = f.read()
            res[name + "_bytes"] = len(outs[name])
        else:
            res[name + "_err"] = (r.stdout + r.stderr)[-500:]
    if os.path.exists(GOLDEN) and "amd2" in outs and not extra:
        import hashlib
        with open(GOLDEN) as f:
            gold = json.load(f).get(str(w))
        with open(src, "rb") as f:
            same_input = gold is not None and hashlib.sha256(f.read()).hexdigest() == gold["input_sha256"]
        if gold and same_input:
            res["identical_to_reference_golden"] = hashlib.sha256(outs["amd2"]).hexdigest() == gold["sha256"]
            res["reference_seconds_on_build_host"] = gold["reference_seconds_here"]
    if "ref" in outs and "amd2" in outs:
        res["identical"] = outs["ref"] == outs["amd2"]
    if "amd" in outs and "amd2" in outs:
        res["identical_amd_amd2"] = outs["amd"] == outs["amd2"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
