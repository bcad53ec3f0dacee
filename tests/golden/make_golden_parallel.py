"""Golden SHA-256 of the REAL reference's gzip stream for inputs that take hours on one core.

The reference's ZopfliDeflate (deflate.c:916-923) runs ZopfliDeflatePart over 1 000 000-byte master
blocks that are independent of each other; only the output bit position `bp` is carried.  Here every
master block is one task for a worker process: the reference's own ZopfliDeflatePart
(oracle/_ref/libzopfli_ref.so) on `in[max(0, start - 32768) .. end)` with bp = 0, `final` only on the
last one; the pieces are then joined in stream order at bit granularity (deflate packs bits LSB first)
and wrapped as gzip_container.c:87-118 does.  A piece that holds a STORED block (deflate.c:297-333 pads
to a byte boundary, so its bits depend on the bp it starts at) is found by walking the piece's block
headers with zlib's inflate(Z_BLOCK) and is computed again by the reference with the bp it really gets.  `--check` runs the same procedure on cases
that already have a whole-stream golden and compares.

    python tests/golden/make_golden_parallel.py --cls M --size 200000000 -n 50 --bs 1 [--procs 6]
    python tests/golden/make_golden_parallel.py --check

Results are appended to tests/golden/vectors_big4.json (same record layout as make_golden.py)."""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import struct
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MB = 1000000
WINDOW = 32768
_DATA = None


class _ZStream(__import__("ctypes").Structure):
    import ctypes as _c
    _fields_ = [("next_in", _c.c_void_p), ("avail_in", _c.c_uint), ("total_in", _c.c_ulong),
                ("next_out", _c.c_void_p), ("avail_out", _c.c_uint), ("total_out", _c.c_ulong),
                ("msg", _c.c_char_p), ("state", _c.c_void_p), ("zalloc", _c.c_void_p), ("zfree", _c.c_void_p),
                ("opaque", _c.c_void_p), ("data_type", _c.c_int), ("adler", _c.c_ulong), ("reserved", _c.c_ulong)]


def has_stored_block(piece, nbits, startbit=0, dictionary=b""):
    """True iff one of the deflate blocks of `piece` (a raw deflate fragment that starts at bit `startbit` of its
    first byte) has BTYPE 00.  zlib's inflate(Z_BLOCK) stops at every block boundary and tells the bit position."""
    import ctypes
    z = ctypes.CDLL("libz.so.1")
    z.zlibVersion.restype = ctypes.c_char_p
    s = _ZStream()
    if z.inflateInit2_(ctypes.byref(s), -15, z.zlibVersion(), ctypes.sizeof(s)) != 0:
        raise RuntimeError("inflateInit2")
    if dictionary:   # the 32 KiB before the master block: matches reach back into them
        if z.inflateSetDictionary(ctypes.byref(s), dictionary, len(dictionary)) != 0:
            raise RuntimeError("inflateSetDictionary")
    buf = ctypes.create_string_buffer(piece, len(piece))
    sink = ctypes.create_string_buffer(1 << 20)
    s.next_in = ctypes.cast(buf, ctypes.c_void_p).value
    s.avail_in = len(piece)
    if startbit:
        z.inflatePrime(ctypes.byref(s), 8 - startbit, piece[0] >> startbit)
        s.next_in += 1
        s.avail_in -= 1
    bitpos = startbit
    stored = False
    while True:
        # the header of the block that starts at bitpos
        if bitpos + 3 > startbit + nbits:   # the end of the fragment (what follows in its last byte is padding)
            break
        w = piece[bitpos >> 3] | (piece[(bitpos >> 3) + 1] << 8 if (bitpos >> 3) + 1 < len(piece) else 0)
        hdr = (w >> (bitpos & 7)) & 7
        if ((hdr >> 1) & 3) == 0:
            stored = True
            break
        if hdr & 1:
            break   # final block
        # run to the end of this block
        while True:
            s.next_out = ctypes.cast(sink, ctypes.c_void_p).value
            s.avail_out = len(sink)
            rc = z.inflate(ctypes.byref(s), 5)   # Z_BLOCK
            if rc not in (0, 1, -5):
                raise RuntimeError("inflate rc %d" % rc)
            if (s.data_type & 128) and not (s.data_type & 256):
                consumed = (s.total_in + (1 if startbit else 0)) * 8 - (s.data_type & 63 & 7)
                if consumed > bitpos:
                    bitpos = consumed
                    break
            if rc == 1 or (s.avail_in == 0 and s.avail_out != 0):
                bitpos = len(piece) * 8 + 8
                break
    z.inflateEnd(ctypes.byref(s))
    return stored


def ref_part(buf, instart, inend, final, n, bs, smax, bp0=0):
    """The reference's ZopfliDeflatePart started at bit bp0 of a fresh byte -> (bytes, first bit, number of bits)."""
    import ctypes

    import oracle_lib as ol
    lib = ol.ref()
    o = ol.RefOptions(0, 0, n, bs, 0, smax)
    out, size, bp = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0), ctypes.c_ubyte(bp0)
    if bp0:
        libc = ctypes.CDLL(None)
        libc.calloc.restype = ctypes.c_void_p
        out = ctypes.cast(libc.calloc(1, 1), ctypes.POINTER(ctypes.c_ubyte))
        size = ctypes.c_size_t(1)
    lib.ZopfliDeflatePart(ctypes.byref(o), 2, final, buf, instart, inend, ctypes.byref(bp), ctypes.byref(out),
                          ctypes.byref(size))
    r = ctypes.string_at(out, size.value)
    ol._libc.free(out)
    return r, bp0, len(r) * 8 - ((8 - bp.value) & 7) - bp0


def _piece(task):
    b, nblocks, n, bs, smax = task
    lo = max(0, b * MB - WINDOW)
    hi = min(len(_DATA), (b + 1) * MB)
    t0 = time.time()
    r, _, nbits = ref_part(bytes(_DATA[lo:hi]), b * MB - lo, hi - lo, 1 if b == nblocks - 1 else 0, n, bs, smax)
    return b, r, nbits, time.time() - t0, has_stored_block(r, nbits, 0, bytes(_DATA[lo:b * MB]))


def merge_bits(pieces, redo=None):
    """pieces: [(bytes, nbits, has_stored)] in stream order (each computed from bit 0) -> bytes of the concatenated
    LSB-first bit stream.  redo(i, bp) recomputes piece i started at bit bp (pieces with a stored block)."""
    out = bytearray()
    off = 0
    nredo = 0
    for i, (data, nb, stored) in enumerate(pieces):
        sh = off & 7
        first = 0
        if stored and sh:
            data, first, nb = redo(i, sh)
            nredo += 1
        a = np.frombuffer(data, dtype=np.uint8).astype(np.uint16)
        need = (off + nb + 7) // 8 + 2
        if len(out) < need:
            out.extend(bytes(need - len(out)))
        o = np.frombuffer(out, dtype=np.uint8)
        base = off >> 3
        if first == sh:      # already in place (recomputed at its own bit offset, or byte aligned)
            o[base:base + len(a)] |= a.astype(np.uint8)
        else:
            assert first == 0
            o[base:base + len(a)] |= ((a << sh) & 0xff).astype(np.uint8)
            o[base + 1:base + 1 + len(a)] |= (a >> (8 - sh)).astype(np.uint8)
        del o
        off += nb
    return bytes(out[:(off + 7) // 8]), nredo


def gzip_wrap(deflate, data):
    # gzip_container.c:87-118: magic, CM 8, FLG 0, MTIME 0, XFL 2, OS 3, stream, CRC-32, ISIZE
    return (bytes([31, 139, 8, 0, 0, 0, 0, 0, 2, 3]) + deflate +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff))


def run_case(cls, size, n, bs, smax, procs, seed=None):
    global _DATA
    from zopfli_amd import generate
    _DATA = generate(cls, size, seed)
    nblocks = max(1, (size + MB - 1) // MB)
    t0 = time.time()
    pieces = [None] * nblocks
    core = 0.0
    with mp.get_context("fork").Pool(procs) as pool:
        for b, r, nbits, dt, stored in pool.imap_unordered(_piece, [(b, nblocks, n, bs, smax) for b in range(nblocks)],
                                                   chunksize=1):
            pieces[b] = (r, nbits, stored)
            core += dt
            done = sum(p is not None for p in pieces)
            if done % 10 == 0:
                print("  %d / %d master blocks, %.0f s wall" % (done, nblocks, time.time() - t0), flush=True)

    def redo(i, bp):
        lo = max(0, i * MB - WINDOW)
        hi = min(len(_DATA), (i + 1) * MB)
        return ref_part(bytes(_DATA[lo:hi]), i * MB - lo, hi - lo, 1 if i == nblocks - 1 else 0, n, bs, smax, bp)

    merged, nredo = merge_bits(pieces, redo)
    print("  %d pieces hold a stored block, %d computed again at their real bit offset" %
          (sum(p[2] for p in pieces), nredo), flush=True)
    stream = gzip_wrap(merged, bytes(_DATA))
    inp = {"kind": "class", "cls": cls, "size": size}
    if seed is not None:
        inp["seed"] = seed
    return {"input": inp, "format": 0, "numiterations": n, "blocksplitting": bs, "blocksplittingmax": smax,
            "sha256": hashlib.sha256(stream).hexdigest(), "outsize": len(stream), "insize": size,
            "how": "ZopfliDeflatePart per master block in %d processes, merged at bit granularity "
                   "(make_golden_parallel.py); %.0f s wall, %.0f s of core time" % (procs, time.time() - t0, core)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cls", default="M")
    ap.add_argument("--size", type=int, default=200000000)
    ap.add_argument("-n", type=int, default=50)
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--smax", type=int, default=15)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    if a.check:
        # whole-stream goldens written by make_golden.py: T 4 MB (bs 1 and 0), M 2.5 MB
        with open(os.path.join(HERE, "vectors.json")) as f:
            vec = json.load(f)
        ok = True
        for v in vec:
            i = v["input"]
            if i.get("kind") == "class" and i["size"] >= 2500000 and v["format"] == 0:
                got = run_case(i["cls"], i["size"], v["numiterations"], v["blocksplitting"], v["blocksplittingmax"],
                               a.procs, i.get("seed"))
                same = got["sha256"] == v["sha256"] and got["outsize"] == v["outsize"]
                print(i, "bs", v["blocksplitting"], "identical" if same else "DIFFERENT")
                ok &= same
        sys.exit(0 if ok else 1)
    got = run_case(a.cls, a.size, a.n, a.bs, a.smax, a.procs)
    path = os.path.join(HERE, "vectors_big4.json")
    have = []
    if os.path.exists(path):
        with open(path) as f:
            have = json.load(f)
    have = [h for h in have if not (h["input"] == got["input"] and h["numiterations"] == got["numiterations"] and
                                    h["blocksplitting"] == got["blocksplitting"])]
    have.append(got)
    with open(path, "w") as f:
        json.dump(have, f, indent=1)
    print(json.dumps(got))


if __name__ == "__main__":
    main()
