"""SURVEY 8 f-3 on the GPU: zopflipng's per-row filter search (zmx_png_filter_types) against LodePNG's own `filter`,
and libzopflipng_amd.so — the optimiser library with the strategy trials side by side, the row search and the deflate
on the device — against the reference's zopflipng, file for file."""
import ctypes
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from zopfli_amd import Context, api

pytestmark = pytest.mark.gpu

LFS_MINSUM, LFS_ENTROPY = 5, 6      # lodepng.h:680-698
LCT_GREY, LCT_RGB, LCT_PALETTE, LCT_GREY_ALPHA, LCT_RGBA = 0, 2, 3, 4, 6


def _filter_ref():
    from zopfli_amd._build import PNG_FILTER_REF
    if not os.path.exists(PNG_FILTER_REF):
        pytest.skip("tests/_build/libpng_filter_ref.so not built (needs /root/reference at build time)")
    lib = ctypes.CDLL(PNG_FILTER_REF)
    lib.ref_png_filter.argtypes = [ctypes.c_char_p, ctypes.c_char_p] + [ctypes.c_uint] * 5
    lib.ref_png_filter.restype = ctypes.c_uint
    return lib


def _image(rng, w, h, channels, depth, kind):
    """Raw scanlines (h rows of linebytes bytes) of a gradient with a little noise (what PNG filters are made for),
    of flat colour with rare changes (ties between the filter types), or of noise."""
    bpp = channels * depth
    linebytes = (w * bpp + 7) // 8
    if kind == "noise":
        return rng.integers(0, 256, size=(h, linebytes), dtype=np.uint8), linebytes
    if kind == "flat":
        img = np.full((h, linebytes), 17, dtype=np.uint8)
        for _ in range(max(1, h // 3)):
            img[rng.integers(0, h), rng.integers(0, linebytes)] = rng.integers(0, 256)
        img[h // 2:] = 0
        return img, linebytes
    y, x = np.mgrid[0:h, 0:linebytes]
    img = (x * 3 // max(1, bpp // 8 if bpp >= 8 else 1) + y * 2 + rng.integers(-2, 3, size=(h, linebytes))) & 255
    return img.astype(np.uint8), linebytes


CASES = [
    # (w, h, colortype, channels, bitdepth, kind)
    (257, 33, LCT_RGBA, 4, 8, "gradient"),
    (1024, 64, LCT_RGBA, 4, 8, "gradient"),
    (300, 40, LCT_RGB, 3, 8, "gradient"),
    (513, 20, LCT_GREY, 1, 8, "gradient"),
    (129, 17, LCT_GREY, 1, 16, "gradient"),
    (77, 19, LCT_RGBA, 4, 16, "noise"),
    (1001, 9, LCT_GREY, 1, 4, "gradient"),       # packed pixels: the filters work on bytes, bytewidth 1
    (333, 11, LCT_PALETTE, 1, 2, "noise"),
    (64, 50, LCT_RGBA, 4, 8, "flat"),            # equal scores: the first type wins
    (5, 1, LCT_RGB, 3, 8, "noise"),              # one row: no row above
    (4096, 3, LCT_GREY_ALPHA, 2, 8, "gradient"),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-ct{c[2]}-d{c[4]}-{c[5]}")
def test_png_filter_types_vs_lodepng(gpu_lib, case):
    """zmx_png_filter_types == the filter type byte LodePNG's filter() puts in front of every scanline, for LFS_MINSUM and
    LFS_ENTROPY (lodepng.cpp:5490-5570), on 8- and 16-bit, packed and one-row images, with ties."""
    ref = _filter_ref()
    w, h, colortype, channels, depth, kind = case
    rng = np.random.default_rng(w * 131 + h)
    img, linebytes = _image(rng, w, h, channels, depth, kind)
    bytewidth = (channels * depth + 7) // 8
    raw = img.tobytes()
    want = {}
    for name, lfs in (("minsum", LFS_MINSUM), ("entropy", LFS_ENTROPY)):
        out = ctypes.create_string_buffer(h * (linebytes + 1))
        assert ref.ref_png_filter(out, raw, w, h, colortype, depth, lfs) == 0
        want[name] = np.frombuffer(out.raw, dtype=np.uint8).reshape(h, linebytes + 1)[:, 0].copy()
    ctx = Context(0, gpu_lib)
    try:
        a = np.zeros(h, dtype=np.uint8)
        b = np.zeros(h, dtype=np.uint8)
        fn = gpu_lib.zmx_png_filter_types
        fn.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(ctx.handle, raw, linebytes, h, bytewidth, a.ctypes.data, b.ctypes.data)
        assert rc == 0, ctx.error()
    finally:
        ctx.close()
    assert np.array_equal(a, want["minsum"]), (a[:20], want["minsum"][:20])
    assert np.array_equal(b, want["entropy"]), (b[:20], want["entropy"][:20])


def _chunk(tag, data):
    body = tag + data
    return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)


def _png(path, w, h, colortype, depth, rows, extra=b"", plte=None):
    raw = b"".join(b"\x00" + bytes(r) for r in rows)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colortype, 0, 0, 0))
    if plte is not None:
        png += _chunk(b"PLTE", plte)
    png += extra + _chunk(b"IDAT", zlib.compress(raw, 6)) + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def _make_inputs(tmp_path):
    rng = np.random.default_rng(5)
    files = {}
    # RGBA gradient with noise and a transparent corner (lossy_transparent has something to do)
    w, h = 200, 120
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([x * 255 // (w - 1), y * 255 // (h - 1), (x + y) // 3 % 256, np.full_like(x, 255)], axis=-1).astype(np.int32)
    img[..., :3] += rng.integers(-3, 4, size=(h, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    img[:30, :40, 3] = 0
    files["rgba"] = str(tmp_path / "rgba.png")
    _png(files["rgba"], w, h, 6, 8, img.reshape(h, w * 4), extra=_chunk(b"tEXt", b"Comment\x00kept or not"))
    # palette image, 4 bits, 12 colours
    w, h = 301, 77
    idx = ((np.mgrid[0:h, 0:w][1] // 7 + np.mgrid[0:h, 0:w][0] // 5) % 12).astype(np.uint8)
    packed = np.zeros((h, (w + 1) // 2), dtype=np.uint8)
    packed[:, :w // 2] = (idx[:, 0:w - 1:2] << 4) | idx[:, 1:w:2]
    packed[:, -1] = idx[:, -1] << 4
    files["pal"] = str(tmp_path / "pal.png")
    _png(files["pal"], w, h, 3, 4, packed, plte=bytes(rng.integers(0, 256, size=36, dtype=np.uint8)))
    # 16-bit grey
    w, h = 90, 60
    g = (np.mgrid[0:h, 0:w][0] * 700 + np.mgrid[0:h, 0:w][1] * 300 + rng.integers(0, 50, size=(h, w))).astype(">u2")
    files["g16"] = str(tmp_path / "g16.png")
    _png(files["g16"], w, h, 0, 16, g.view(np.uint8).reshape(h, w * 2))
    # tiny: the "smaller without the palette" branch
    files["tiny"] = str(tmp_path / "tiny.png")
    _png(files["tiny"], 6, 5, 2, 8, rng.integers(0, 2, size=(5, 18), dtype=np.uint8) * 255)
    return files


RUNS = [
    ("rgba", ["--iterations=3"]),
    ("rgba", ["--iterations=3", "--filters=me"]),
    ("rgba", ["--iterations=2", "--lossy_transparent", "--keepchunks=tEXt"]),
    ("pal", ["--iterations=3"]),
    ("g16", ["--iterations=3"]),
    ("g16", ["--iterations=2", "--lossy_8bit", "--filters=e"]),
    ("tiny", ["--iterations=3"]),
]


@pytest.mark.parametrize("which,args", RUNS, ids=lambda v: v if isinstance(v, str) else "_".join(a.lstrip("-") for a in v))
def test_libzopflipng_amd_vs_reference(tmp_path, which, args):
    """The reference's zopflipng command line linked against libzopflipng_amd.so (strategy trials on host threads, MINSUM /
    ENTROPY row search on the device, ZopfliDeflate on the device) writes the reference's PNG byte for byte: automatic
    strategy choice, named strategies, lossy options, kept chunks, palette, 16-bit and tiny images
    (zopflipng_lib.cc:160-470)."""
    from zopfli_amd._build import PNG_AMD2, PNG_REF
    if not (os.path.exists(PNG_AMD2) and os.path.exists(PNG_REF)):
        pytest.skip("tests/_build/zopflipng_amd2 / zopflipng_ref not built (needs /root/reference at build time)")
    src = _make_inputs(tmp_path)[which]
    outs = {}
    for name, exe in (("amd", PNG_AMD2), ("ref", PNG_REF)):
        dst = str(tmp_path / (name + ".png"))
        r = subprocess.run([exe, "-y"] + args + [src, dst], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stdout[-1000:], r.stderr[-1000:])
        with open(dst, "rb") as f:
            outs[name] = f.read()
    assert outs["amd"] == outs["ref"]


def test_device_filters_are_used(tmp_path):
    """The device's row search and LodePNG's own give the same file (ZOPFLIPNG_AMD_HOST_FILTERS=1 turns the device's off):
    the LFS_PREDEFINED hand-over changes nothing but who searched."""
    from zopfli_amd._build import PNG_AMD2
    if not os.path.exists(PNG_AMD2):
        pytest.skip("tests/_build/zopflipng_amd2 not built")
    src = _make_inputs(tmp_path)["rgba"]
    outs = []
    for env in ({}, {"ZOPFLIPNG_AMD_HOST_FILTERS": "1"}):
        dst = str(tmp_path / ("o%d.png" % len(outs)))
        r = subprocess.run([PNG_AMD2, "-y", "--iterations=2", "--filters=m", src, dst], capture_output=True, text=True,
                           timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]
        with open(dst, "rb") as f:
            outs.append(f.read())
    assert outs[0] == outs[1]


def _at_size_png(path, w):
    """tools/png_at_size.py's synthetic W x W RGBA image (the generator is repeated here so that the test reads no file
    outside tests/): a gradient with +-3 of noise, seed 7."""
    rng = np.random.default_rng(7)
    y, x = np.mgrid[0:w, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(w - 1, 1)), ((x + y) // 3 % 256),
                    np.full_like(x, 255)], axis=-1).astype(np.int32)
    img[..., :3] += rng.integers(-3, 4, size=(w, w, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    raw = np.concatenate([np.zeros((w, 1), dtype=np.uint8), img.reshape(w, w * 4)], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, w, 8, 6, 0, 0, 0))
                + _chunk(b"IDAT", zlib.compress(raw, 1)) + _chunk(b"IEND", b""))


@pytest.mark.parametrize("width", [1024, 4096])
def test_png_at_size_vs_reference_golden(tmp_path, width):
    """BASELINE configs[4] AT SIZE in the driver-run suite: the reference's zopflipng command line on libzopflipng_amd.so,
    default options, on the 1024 x 1024 and the 4096 x 4096 RGBA image of tools/png_at_size.py; the output's SHA-256 and
    size are those of the all-reference zopflipng (tests/golden/png_at_size.json, written by
    `tools/png_at_size.py --make-golden` on the build host: 5.6 s / 87 s of the reference there)."""
    import hashlib
    import json
    from zopfli_amd._build import PNG_AMD2
    if not os.path.exists(PNG_AMD2):
        pytest.skip("tests/_build/zopflipng_amd2 not built (needs /root/reference at build time)")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "png_at_size.json")) as f:
        gold = json.load(f)[str(width)]
    src, dst = str(tmp_path / "in.png"), str(tmp_path / "out.png")
    _at_size_png(src, width)
    with open(src, "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == gold["input_sha256"], "the synthetic input is not the golden's"
    r = subprocess.run([PNG_AMD2, "-y", src, dst], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-1000:])
    with open(dst, "rb") as f:
        out = f.read()
    assert len(out) == gold["bytes"]
    assert hashlib.sha256(out).hexdigest() == gold["sha256"]
