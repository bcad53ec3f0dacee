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
from zopfli_amd._build import PNG_AMD, PNG_REF  # noqa: E402


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


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
    with_ref = "--ref" in sys.argv
    extra = [a for a in sys.argv[1:] if a.startswith("--") and a != "--ref"]
    tmp = os.environ.get("TMPDIR", "/tmp")
    src = os.path.join(tmp, f"in{w}.png")
    write_png(src, w, w, 7)
    res = {"width": w, "height": w, "input_bytes": os.path.getsize(src), "args": extra}
    outs = {}
    for name, exe in (("amd", PNG_AMD),) + ((("ref", PNG_REF),) if with_ref else ()):
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
    if "ref" in outs and "amd" in outs:
        res["identical"] = outs["ref"] == outs["amd"]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
