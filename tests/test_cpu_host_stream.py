"""Host logic (block splitter, block cost model, iteration control, encoder, containers) on CPU:
the product's host sources linked against the oracle-backed zmx layer
(tests/_build/libzopfli_hosttest.so) must produce the reference's bytes."""
import gzip
import hashlib
import json
import os
import zlib

import pytest

import oracle_lib as ol
from zopfli_amd import ZopfliOptions, api, generate

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "vectors.json")


@pytest.fixture(scope="module")
def host():
    return ol.hosttest_library()


def _input(spec):
    if spec["kind"] == "literal":
        from golden.make_golden import LITERALS
        return LITERALS[spec["name"]]
    return generate(spec["cls"], spec["size"], spec.get("seed"))


def _golden(max_size):
    with open(GOLDEN) as f:
        return [c for c in json.load(f) if c["insize"] <= max_size]


@pytest.mark.parametrize("case", _golden(70000), ids=lambda c: f"{c['input'].get('name', c['input'].get('cls'))}-"
                         f"{c['insize']}-f{c['format']}-n{c['numiterations']}-s{c['blocksplitting']}")
def test_golden_small(host, case):
    data = _input(case["input"])
    opt = ZopfliOptions(case["numiterations"], case["blocksplitting"], case["blocksplittingmax"])
    out = api.compress(data, case["format"], opt, lib=host)
    assert len(out) == case["outsize"]
    assert hashlib.sha256(out).hexdigest() == case["sha256"]


def test_known_answers(host):
    """SURVEY.md Appendix B.3 (defaults)."""
    assert api.compress(b"", 0, lib=host).hex() == "1f8b080000000000020303000000000000000000"
    assert api.compress(b"a", 0, lib=host).hex() == "1f8b08000000000002034b040043beb7e801000000"
    assert api.compress(b"", 2, lib=host).hex() == "0300"
    assert api.compress(b"a", 2, lib=host).hex() == "4b0400"
    go = b"compressthis" + b"_foobar" * 1000 + b"$"
    out = api.compress(go, 0, lib=host)
    assert hashlib.sha256(out).hexdigest() == "01e98e796c1103741baf129c817b15a12d70fc24c07b103ecc37119915cdc7d4"
    assert gzip.decompress(out) == go


def test_go_tests(host):
    """go/zopfli/zopfli_test.go:35-69: round trip + size bounds."""
    import random
    go = b"compressthis" + b"_foobar" * 1000 + b"$"
    out = api.compress(go, 0, lib=host)
    assert gzip.decompress(out) == go and len(out) <= 500
    rnd = bytes(random.Random(1).getrandbits(8) for _ in range(3000))
    out = api.compress(rnd, 0, lib=host)
    assert gzip.decompress(out) == rnd and len(out) <= 3100
    assert len(api.compress(b"", 0, lib=host)) <= 20


@pytest.mark.skipif(not ol.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("btype", [0, 1, 2])
def test_deflate_part_and_bp(host, btype):
    """ZopfliDeflatePart with a dictionary before instart, forced block types, and the bit pointer
    carried between consecutive calls (deflate.h:50-53)."""
    import ctypes
    data = generate("M", 90000)
    a, bpa = api.deflate_part(data, 40000, 90000, btype, 1, ZopfliOptions(5), lib=host)
    b, bpb = ol.ref_deflate_part(data, 40000, 90000, btype, 1, 5)
    assert (a, bpa) == (b, bpb)
    # two consecutive parts sharing one output array
    opt = ZopfliOptions(3)
    out, size, bp = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
    for (s, e, fin) in ((0, 30000, 0), (30000, 60000, 1)):
        host.ZopfliDeflatePart(ctypes.byref(opt), btype, fin, data, s, e, ctypes.byref(bp), ctypes.byref(out),
                               ctypes.byref(size))
    mine = ctypes.string_at(out, size.value)
    assert zlib.decompress(mine, -15) == data[:60000]
    ropt = ol.RefOptions(0, 0, 3, 1, 0, 15)
    rout, rsize, rbp = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
    for (s, e, fin) in ((0, 30000, 0), (30000, 60000, 1)):
        ol.ref().ZopfliDeflatePart(ctypes.byref(ropt), btype, fin, data, s, e, ctypes.byref(rbp), ctypes.byref(rout),
                                   ctypes.byref(rsize))
    assert mine == ctypes.string_at(rout, rsize.value) and bp.value == rbp.value


def test_formats_round_trip(host):
    data = generate("X", 30000)
    opt = ZopfliOptions(2)
    assert gzip.decompress(api.compress(data, 0, opt, lib=host)) == data
    assert zlib.decompress(api.compress(data, 1, opt, lib=host)) == data
    assert zlib.decompress(api.compress(data, 2, opt, lib=host), -15) == data


def test_master_block_chunks_merge(host):
    """zmx_deflate_range on disjoint ranges + zmx_chunks_merge == ZopfliDeflate of the whole input
    (the multi-GPU gather path), including a stored block carried as raw bytes."""
    from zopfli_amd import Context
    data = generate("M", 1300000) + generate("R", 1000000)
    opt = ZopfliOptions(1)
    whole, _ = api.deflate(data, 2, 1, opt, lib=host)
    ctx = Context(0, host)
    ctx.set_input(data)
    blobs = [ctx.deflate_range(opt, 0, 1000000, 0), ctx.deflate_range(opt, 1000000, len(data), 1)]
    assert ctx.merge(blobs) == whole
    ctx.close()
