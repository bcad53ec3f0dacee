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


def _blob(chunks):
    """The chunk blob layout of zmx_deflate_range (deflate.cc): u64 count, then per chunk u8 kind,
    u8 final, u64 a, u64 b, payload (bit chunks: a = nbits, b = payload bytes; stored: a = 0)."""
    out = len(chunks).to_bytes(8, "little")
    for kind, final, bits_or_raw in chunks:
        if kind == 0:
            bits = bits_or_raw
            payload = bytearray((len(bits) + 7) // 8)
            for i, b in enumerate(bits):
                payload[i >> 3] |= b << (i & 7)
            out += bytes([0, 0]) + len(bits).to_bytes(8, "little") + len(payload).to_bytes(8, "little") + bytes(payload)
        else:
            out += bytes([1, final]) + (0).to_bytes(8, "little") + len(bits_or_raw).to_bytes(8, "little") + bits_or_raw
    return out


def _merge_reference(chunks, prefix_bits):
    """Bit-at-a-time model of AddBits / AddNonCompressedBlock (deflate.c:38-72, 625-665)."""
    bits = list(prefix_bits)
    for kind, final, payload in chunks:
        if kind == 0:
            bits += payload
            continue
        pos, n = 0, len(payload)
        while True:
            piece = min(65535, n - pos)
            last = pos + piece >= n
            bits += [1 if (final and last) else 0, 0, 0]
            bits += [0] * (-len(bits) % 8)
            for v in (piece & 255, piece >> 8, (~piece & 0xffff) & 255, (~piece & 0xffff) >> 8):
                bits += [(v >> k) & 1 for k in range(8)]
            for byte in payload[pos:pos + piece]:
                bits += [(byte >> k) & 1 for k in range(8)]
            if last:
                break
            pos += piece
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        out[i >> 3] |= b << (i & 7)
    return bytes(out), len(bits) & 7


@pytest.mark.parametrize("seed", range(6))
def test_chunks_merge_bit_offsets(host, seed):
    """zmx_chunks_merge places every chunk by a prefix sum of bit lengths and shifts the chunks in
    parallel: chunks shorter than a byte, chunks ending on byte boundaries, stored blocks after any
    bit offset (incl. a header that straddles a byte) and multi-piece stored blocks."""
    import ctypes
    import random
    ctypes.CDLL(None).mallopt(-6, 0xA5)   # M_PERTURB: fresh malloc/realloc memory is not zero
    rng = random.Random(seed)
    chunks = []
    for _ in range(rng.randint(1, 40)):
        r = rng.random()
        if r < 0.15:
            n = rng.choice([0, 1, 5, 70000, 131070, 140000]) if seed % 2 else rng.choice([0, 1, 2, 300])
            chunks.append((1, rng.randint(0, 1), bytes(rng.getrandbits(8) for _ in range(n))))
        else:
            n = rng.choice([0, 1, 2, 3, 5, 7, 8, 9, 15, 16, 17, 63, 64, 65, rng.randint(1, 3000)])
            chunks.append((0, 0, [rng.getrandbits(1) for _ in range(n)]))
    # split the chunk list over 1..3 blobs, as ranks would deliver it
    cuts = sorted(rng.sample(range(len(chunks) + 1), min(2, len(chunks) + 1)))
    parts = [chunks[:cuts[0]]] + [chunks[a:b] for a, b in zip(cuts, cuts[1:])] + [chunks[cuts[-1]:]]
    blobs = [_blob(p) for p in parts]
    for nprefix_bits in (0, 3, 8, 13):
        prefix_bits = [rng.getrandbits(1) for _ in range(nprefix_bits)]
        want, want_bp = _merge_reference(chunks, prefix_bits)
        # seed (*out, *outsize, *bp) with the prefix, reference conventions
        pre = bytearray((nprefix_bits + 7) // 8)
        for i, b in enumerate(prefix_bits):
            pre[i >> 3] |= b << (i & 7)
        libc = ctypes.CDLL(None)
        libc.malloc.restype = ctypes.c_void_p
        libc.malloc.argtypes = [ctypes.c_size_t]
        libc.free.argtypes = [ctypes.c_void_p]
        cap = 1
        while cap < len(pre):
            cap <<= 1
        addr = libc.malloc(cap) if pre else None
        if pre:
            ctypes.memmove(addr, bytes(pre), len(pre))
        out = ctypes.cast(addr, ctypes.POINTER(ctypes.c_ubyte))
        size, bp = ctypes.c_size_t(len(pre)), ctypes.c_ubyte(nprefix_bits & 7)
        arr = (ctypes.c_void_p * len(blobs))(*[ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value for b in blobs])
        sizes = (ctypes.c_size_t * len(blobs))(*[len(b) for b in blobs])
        assert host.zmx_chunks_merge(arr, sizes, len(blobs), ctypes.byref(bp), ctypes.byref(out), ctypes.byref(size)) == 0
        got = ctypes.string_at(out, size.value)
        libc.free(ctypes.cast(out, ctypes.c_void_p))
        assert bp.value == want_bp
        assert got == want
    ctypes.CDLL(None).mallopt(-6, 0)


def test_multi_device_sharding_through_the_c_entry_point():
    """ZopfliCompress with the master blocks dealt over several devices inside the library (api.cc
    RunPartsSharded; here three pretend devices of the oracle-backed layer): the stream equals the
    one-device stream and the reference's, for a size that leaves the devices unequal shares, a
    stored-block region (random bytes) and a part smaller than a master block."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "host = ol.hosttest_library()\n"
        "data = generate('M', 2300000) + generate('R', 400000)\n"
        "print(hashlib.sha256(api.compress(data, 0, ZopfliOptions(1), lib=host)).hexdigest())\n"
        "print(hashlib.sha256(api.deflate_part(data, 1500000, 2600000, 2, 1, ZopfliOptions(1), lib=host)[0]).hexdigest())\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    res = {}
    for ndev in ("1", "3"):
        env = dict(os.environ, ZOPFLI_HOSTTEST_DEVICES=ndev, ZOPFLI_AMD_DEVICES="all")
        env.pop("LOCAL_RANK", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        res[ndev] = r.stdout.split()
    assert res["1"] == res["3"]
    data = generate("M", 2300000) + generate("R", 400000)
    assert res["3"][0] == hashlib.sha256(ol.ref_compress(data, 0, 1)).hexdigest()


def test_rounds_of_master_blocks():
    """A call of more master blocks than a round holds (2 000 = 2 GB: the device layer's positions are 32 bits, the
    reference's insize a size_t) is done in rounds, each dealt over the contexts like a call of its own, the chunks and the
    container checksum joined in stream order (api.cc RunPartsSharded).  ZOPFLI_AMD_ROUND_PARTS=1 forces rounds on 2.3
    master blocks: gzip (CRC-32 over the rounds), zlib (Adler-32) and raw deflate equal the one-round streams and the
    reference's."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "host = ol.hosttest_library()\n"
        "data = generate('T', 1300000) + generate('R', 300000) + generate('Z', 700000)\n"
        "for fmt in (0, 1, 2):\n"
        "    print(hashlib.sha256(api.compress(data, fmt, ZopfliOptions(1), lib=host)).hexdigest())\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    res = {}
    for name, extra in (("one", {}), ("rounds", {"ZOPFLI_AMD_ROUND_PARTS": "1"}),
                        ("rounds3dev", {"ZOPFLI_AMD_ROUND_PARTS": "1", "ZOPFLI_HOSTTEST_DEVICES": "3", "ZOPFLI_AMD_DEVICES": "all"})):
        env = dict(os.environ, **extra)
        env.pop("LOCAL_RANK", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = r.stdout.split()
    assert res["one"] == res["rounds"] == res["rounds3dev"]
    data = generate("T", 1300000) + generate("R", 300000) + generate("Z", 700000)
    assert res["rounds"][0] == hashlib.sha256(ol.ref_compress(data, 0, 1)).hexdigest()


def test_calls_of_few_master_blocks_are_dealt_from_the_nth_on():
    """api.cc ContextPool::Acquire, `polite`: a call below 32 master blocks is dealt over three contexts of its device
    (from 4 master blocks on with block splitting), but the contexts beyond the first are only created from the process's
    ZOPFLI_AMD_DEAL_AFTER-th such call on (8 by default: setting a context up costs more than a short-lived program gets
    back).  With 3: the call trace shows one shard for the first two calls and three for the third and fourth; all four
    streams are the reference's."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api\n"
        "host = ol.hosttest_library()\n"
        "data = bytes(4100000)\n"
        "for k in range(4):\n"
        "    print(hashlib.sha256(api.compress(data, 0, ZopfliOptions(1), lib=host)).hexdigest())\n"
        "    sys.stderr.write('== call %%d done\\n' %% k)\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    env = dict(os.environ, ZOPFLI_AMD_DEAL_AFTER="3", ZOPFLI_AMD_TRACE_CALL="1")
    for k in ("LOCAL_RANK", "ZOPFLI_AMD_DEVICES", "ZOPFLI_AMD_SPLIT_MB"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    digests = r.stdout.split()
    assert len(digests) == 4 and len(set(digests)) == 1
    assert digests[0] == hashlib.sha256(ol.ref_compress(bytes(4100000), 0, 1)).hexdigest()
    per_call = r.stderr.split("== call ")
    shards = [seg.count("shard 2 (") for seg in per_call[:4]]
    assert shards == [0, 0, 1, 1], (shards, r.stderr[-3000:])


@pytest.mark.parametrize("fail", [0, 1, 2])
def test_failed_shard_is_done_again_on_another_context(fail):
    """A shard of a request that fails (ZOPFLI_AMD_TEST_FAIL_SHARD: its first attempt returns an error before it does
    anything — a device out of memory, a broken context) is computed again on a context that finished its own shard
    (api.cc RunPartsSharded) and the stream is the one-device stream; three pretend devices, each shard failing in turn."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "host = ol.hosttest_library()\n"
        "data = generate('M', 2300000) + generate('R', 400000)\n"
        "print(hashlib.sha256(api.compress(data, 0, ZopfliOptions(1), lib=host)).hexdigest())\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    res = {}
    for name, extra in (("plain", {}), ("failing", {"ZOPFLI_AMD_TEST_FAIL_SHARD": str(fail)})):
        env = dict(os.environ, ZOPFLI_HOSTTEST_DEVICES="3", ZOPFLI_AMD_DEVICES="all", **extra)
        env.pop("LOCAL_RANK", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = (r.stdout.split(), r.stderr)
    assert res["plain"][0] == res["failing"][0]
    assert "done again on another context" in res["failing"][1]


def test_retry_goes_by_error_class_not_by_text():
    """The injected failure's MESSAGE holds "pool" and "too large" — the words round 5's filter skipped — and is still done
    again (above: every case passes); what decides is zmx_last_error_class (include/zopfli_amd.h), whose values the
    header pins."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "zopfli_amd.h")).read()
    vals = {k: int(v) for k, v in re.findall(r"#define (ZMX_ERR_[A-Z_]+) (\d+)", hdr)}
    assert vals == {"ZMX_ERR_NONE": 0, "ZMX_ERR_DEVICE": 1, "ZMX_ERR_OUT_OF_MEMORY": 2, "ZMX_ERR_REFUSED": 3}
    src = open(os.path.join(os.path.dirname(os.path.dirname(__file__)), "zopfli_amd", "csrc", "host", "api.cc")).read()
    assert 'err.find("pool")' not in src and "err_class == ZMX_ERR_REFUSED" in src


def test_host_block_cache_and_no_mallopt():
    """The library keeps its large host arrays in a cache of its own (block_cache.h) instead of reconfiguring the host
    process's malloc: after a call with block splitting zmx_host_cache_trim() gives bytes back, a second trim nothing;
    with ZOPFLI_AMD_HOST_CACHE_MB=0 nothing is ever cached; the streams are the same; and mallopt is reached only with
    ZOPFLI_AMD_KEEP_HEAP set (mallinfo2's arena does not jump by the 256 MB top pad round 5 asked for)."""
    import subprocess
    import sys
    code = (
        "import ctypes, hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "host = ol.hosttest_library()\n"
        "host.zmx_host_cache_trim.restype = ctypes.c_size_t\n"
        "data = generate('T', 1200000)\n"
        "out = api.compress(data, 0, ZopfliOptions(1, 1, 15), lib=host)\n"
        "a = host.zmx_host_cache_trim(); b = host.zmx_host_cache_trim()\n"
        "print(hashlib.sha256(out).hexdigest(), a, b)\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    res = {}
    for mb in ("1024", "0"):
        env = dict(os.environ, ZOPFLI_AMD_HOST_CACHE_MB=mb)
        env.pop("ZOPFLI_AMD_KEEP_HEAP", None)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mb] = r.stdout.split()
    assert res["1024"][0] == res["0"][0]
    assert int(res["1024"][1]) > 1000000 and int(res["1024"][2]) == 0, res
    assert int(res["0"][1]) == 0, res


@pytest.mark.parametrize("more", [0, 1])
def test_verbose_text_equals_the_references(more):
    """ZopfliOptions::verbose / verbose_more: the library prints the reference's stderr text — block split
    points (blocksplitter.c:148-180), "Iteration i: n bit" per block (squeeze.c:493), treesize and
    compressed block size with the reference's byte counts (deflate.c:719-744), the final summary — in
    the reference's order, although it computes all blocks of a call side by side."""
    import subprocess
    import sys
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    body = (
        "import sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "data = generate('M', 1300000) + generate('R', 3000) + generate('X', 200000)\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    mine = body + ("out = api.compress(data, 0, ZopfliOptions(3, 1, 15, 1, %d), lib=ol.hosttest_library())\n" % more)
    theirs = body + ("out = ol.ref_compress(data, 0, 3, 1, 15, 1, %d)\n" % more)
    texts = []
    for code in (mine, theirs):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        texts.append(r.stderr)
    assert "block split points" in texts[1] and "treesize" in texts[1] and "Iteration" in texts[1]
    assert texts[0] == texts[1]


def _checksum_cases(n):
    """(begin, end) ranges around every boundary of k_checksum's geometry (1 KiB lanes, 256 KiB pieces,
    4-byte body loads), for an input of n bytes"""
    import random
    rng = random.Random(7)
    marks = [0, 1, 2, 3, 4, 5, 1023, 1024, 1025, 2048, 262143, 262144, 262145, 524288, n - 262144, n - 1025, n - 3, n - 1, n]
    marks = sorted({m for m in marks if 0 <= m <= n})
    cases = [(a, b) for a in marks for b in marks if a <= b]
    cases += [tuple(sorted((rng.randrange(n + 1), rng.randrange(n + 1)))) for _ in range(40)]
    return cases


def test_checksum_pieces_and_combine(host):
    """zmx_checksum's piece arithmetic (host/checksum.cc, fed by the CPU stand-in of k_checksum with the
    device's geometry) against zlib's crc32 / adler32: gzip_container.c:75, zlib_container.c:29."""
    from zopfli_amd import Context
    data = generate("X", 300000) + generate("R", 300000) + bytes(200000) + b"\xff" * 70000
    ctx = Context(0, host)
    try:
        ctx.set_input(data)
        for a, b in _checksum_cases(len(data)):
            assert ctx.checksum(api.CRC32, a, b) == zlib.crc32(data[a:b]), (a, b)
            assert ctx.checksum(api.ADLER32, a, b) == zlib.adler32(data[a:b]), (a, b)
        with pytest.raises(RuntimeError):
            ctx.checksum(api.CRC32, 5, len(data) + 1)
        with pytest.raises(RuntimeError):
            ctx.checksum(7, 0, 1)
    finally:
        ctx.close()
    for cut in (0, 1, 1000, 262144, 600001, len(data)):
        left, right = data[:cut], data[cut:]
        assert host.zmx_checksum_combine(api.CRC32, zlib.crc32(left), zlib.crc32(right), len(right)) == zlib.crc32(data)
        assert host.zmx_checksum_combine(api.ADLER32, zlib.adler32(left), zlib.adler32(right), len(right)) == zlib.adler32(data)


def test_concurrent_callers():
    """The reference has no globals: concurrent calls on distinct buffers are allowed.  Here a request
    takes one of a device's contexts (api.cc ContextPool: ZOPFLI_AMD_LANES of them per device, three by default) and
    other callers overlap with it or wait: four threads, each with its own input, get what they get alone."""
    import threading

    import oracle_lib as ol
    from zopfli_amd import ZopfliOptions, api, generate
    lib = ol.hosttest_library()
    inputs = [generate(cls, n, seed) for cls, n, seed in (("T", 150000, 3), ("X", 90000, 4), ("M", 200000, 5), ("R", 30000, 6))]
    want = [api.compress(d, 0, ZopfliOptions(2), lib=lib) for d in inputs]
    got = [None] * len(inputs)

    def work(i):
        got[i] = api.compress(inputs[i], 0, ZopfliOptions(2), lib=lib)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(inputs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert got == want


def test_host_bit_writer_on_every_block():
    """ZOPFLI_AMD_DEVICE_ENCODE=0: every block's bits from the host bit writer (bit_writer.h, block_cost.cc EncodeBlock —
    codes reversed once per block, a symbol's code and extra bits in one piece, four bytes out at a time) instead of
    from zmx_encode_blocks: the streams of every class, with fixed-tree and dynamic blocks, long matches at long
    distances (13 extra bits) and literals only, must equal the reference's."""
    import subprocess
    import sys
    code = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "host = ol.hosttest_library()\n"
        "bad = []\n"
        "cases = [(c, n, it, bs, fmt) for c, n in (('T', 90000), ('X', 60000), ('R', 30000), ('Z', 120000), ('B', 20000), ('P', 70000), ('M', 150000))\n"
        "         for it, bs, fmt in ((1, 1, 0), (3, 0, 2), (2, 1, 1))]\n"
        "cases += [('T', 300, 2, 1, 0), ('R', 40, 1, 1, 2), ('Z', 1000, 1, 0, 0)]\n"
        "for c, n, it, bs, fmt in cases:\n"
        "    data = generate(c, n)\n"
        "    got = api.compress(data, fmt, ZopfliOptions(it, bs), lib=host)\n"
        "    want = ol.ref_compress(data, fmt, it, bs)\n"
        "    if got != want: bad.append((c, n, it, bs, fmt, len(got), len(want)))\n"
        "print('BAD' if bad else 'OK', bad)\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ, ZOPFLI_AMD_DEVICE_ENCODE="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[0] == "OK", r.stdout[-2000:]


def test_device_split_path_of_the_host_code():
    """f-1 on the device, the HOST side of it (deflate.cc, block_split.cc: BlockSplitSizesBatch — no host stores for the
    first split, every round's block sizes asked of zmx_block_costs, the split points' byte positions from one walk over
    the symbols; the second split's sequences glued from the blocks' stores): forced on with the test library's CPU
    stand-ins for the zmx_cost_stores_* entry points, the stream and the verbose text (the split points' line) equal the
    reference's, and equal the host-evaluated path's."""
    import subprocess
    import sys
    if not ol.have_ref():
        pytest.skip("oracle/_ref not built")
    body = (
        "import hashlib, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle_lib as ol\n"
        "from zopfli_amd import ZopfliOptions, api, generate\n"
        "data = generate('M', 1100000) + generate('R', 400000) + generate('T', 700000)\n"
        % (os.path.dirname(os.path.dirname(__file__)), os.path.dirname(__file__)))
    mine = body + "out = api.compress(data, 0, ZopfliOptions(2, 1, 15, 1, 0), lib=ol.hosttest_library())\nprint(hashlib.sha256(out).hexdigest())\n"
    theirs = body + "out = ol.ref_compress(data, 0, 2, 1, 15, 1, 0)\nprint(hashlib.sha256(out).hexdigest())\n"
    runs = []
    forced = {"ZOPFLI_AMD_DEVICE_SPLIT": "2", "ZOPFLI_AMD_DEVICE_SPLIT_FROM": "1", "ZOPFLI_AMD_TRACE_CALL": "1"}
    for code, extra in ((mine, forced), (mine, {"ZOPFLI_AMD_DEVICE_SPLIT": "0"}), (theirs, {}),
                        # the device's answers stop coming in mid-search (the 4th round on): the first split falls back to the
                        # host's search on the downloaded symbols, the second split's later rounds to the host's pool
                        (mine, dict(forced, ZOPFLI_HOSTTEST_COSTS_FAIL_AFTER="3"))):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append((r.stdout.split()[-1], r.stderr))
    assert "first split on the device" in runs[0][1] and "on the device (" in runs[0][1]
    assert runs[0][0] == runs[1][0] == runs[2][0] == runs[3][0]
    assert "no device block sizes" in runs[3][1]
    lines = [[l for l in t.splitlines() if l.startswith("block split points")] for _, t in runs]
    assert lines[0] == lines[1] == lines[2] == lines[3] and len(lines[2]) >= 3
