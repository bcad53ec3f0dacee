"""N > 1 path on CPU: two gloo ranks shard one stream by master block, gather the chunk blobs to
rank 0 and merge — the result must be byte-identical to the one-rank stream (and to the real
reference when oracle/_ref is present).  The ranks run the product's host code over the
oracle-backed zmx layer (tests/_build/libzopfli_hosttest.so); on GPUs the same functions run over
RCCL (bench.py)."""
import gzip
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, size, niter, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_lib as ol
    from zopfli_amd import Context, ZopfliOptions, generate, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        host = ol.hosttest_library()
        data = generate("M", size)
        ctx = Context(0, host)
        out = sharding.gzip_sharded(ctx, ZopfliOptions(niter), data, rank, world, torch.device("cpu"), dist)
        ctx.close()
        if rank == 0:
            with open(out_path, "wb") as f:
                f.write(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("size", [2300000, 900000])
def test_two_ranks_equal_one_rank(tmp_path, size):
    import oracle_lib as ol
    from zopfli_amd import ZopfliOptions, api, generate

    out_path = str(tmp_path / "sharded.gz")
    mp.spawn(_worker, args=(2, _free_port(), size, 1, out_path), nprocs=2, join=True)
    with open(out_path, "rb") as f:
        sharded = f.read()
    data = generate("M", size)
    assert gzip.decompress(sharded) == data
    single = api.compress(data, 0, ZopfliOptions(1), lib=ol.hosttest_library())
    assert sharded == single
    if ol.have_ref():
        assert sharded == ol.ref_compress(data, 0, 1)


def test_shard_ranges_cover_stream():
    from zopfli_amd import sharding
    for n in (0, 1, 999999, 1000000, 1000001, 12345678):
        for w in (1, 2, 3, 8):
            r = sharding.shard_ranges(n, w)
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(s % sharding.MASTER_BLOCK == 0 for s, _ in r if s < n)


def test_crc32_combine():
    import zlib

    from zopfli_amd import sharding
    a, b = os.urandom(1000), os.urandom(777)
    assert sharding.crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b)
    assert sharding.crc32_combine(zlib.crc32(a), 0, 0) == zlib.crc32(a)


@pytest.mark.parametrize("world", [2, 4])
def test_gather_logic_over_sockets(world):
    """dist.cc's gather bookkeeping (zopfli_amd/csrc/host/dist_core.h: the size all-gather, the offsets, rank 0's
    own blob staying put, an empty blob, the agreed failure when ONE rank cannot prepare its buffers) with `world`
    ranks — forked processes around a socket hub instead of RCCL (tests/hostlib/gather_socket_test.cc).  On a GPU the
    same function runs over ncclAllGather / ncclSend / ncclRecv."""
    import ctypes

    import oracle_lib as ol
    lib = ol.hosttest_library()
    lib.zamd_test_gather.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.zamd_test_gather.restype = ctypes.c_int
    assert lib.zamd_test_gather(world, -1) == 0
    for fail in range(world):
        assert lib.zamd_test_gather(world, fail) == 0, f"rank {fail} failing its preparation"


def _hetero():
    """A corpus whose cost is concentrated: text, then a stretch of long runs of equal bytes, then markup and two-symbol
    data — what contiguous equal-count shards put on one rank."""
    from zopfli_amd import generate
    return generate("T", 24000000) + generate("Z", 8000000) + generate("X", 16000000) + generate("B", 8000000, seed=3)


@pytest.mark.parametrize("world", [2, 4])
def test_cost_aware_ranges_agree_and_balance(world):
    """Cost-aware dealing (deflate.c:916-923 is the unit): the ranges are a function of the bytes (two
    computations agree, and so do the library's in-process dealer and sharding.py — both call deal.cc), they cover the
    stream in whole master blocks, and on a heterogeneous corpus no shard's estimated cost is more than 20 % above the
    mean while equal-count shards are off by far more."""
    import oracle_lib as ol
    from zopfli_amd import sharding
    lib = ol.hosttest_library()
    data = _hetero()
    n = len(data)
    r1 = sharding.shard_ranges(n, world, data, lib)
    r2 = sharding.shard_ranges(n, world, bytearray(data), lib)      # (a bytearray or a memoryview is as good as bytes)
    assert r1 == r2 == sharding.shard_ranges(n, world, memoryview(data), lib)
    assert sharding.shard_costs(n, r1, bytearray(data), lib) == sharding.shard_costs(n, r1, data, lib)
    assert r1[0][0] == 0 and r1[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(r1, r1[1:]))
    assert all(s % sharding.MASTER_BLOCK == 0 for s, _ in r1 if s < n)
    assert all(e > s for s, e in r1)
    cost = sharding.shard_costs(n, r1, data, lib)
    mean = sum(cost) / world
    assert max(cost) <= 1.2 * mean, (r1, cost)
    block = sharding.shard_costs(n, [(b, min(n, b + sharding.MASTER_BLOCK)) for b in range(0, n, sharding.MASTER_BLOCK)], data, lib)
    assert max(cost) <= mean + max(block)          # what a prefix walk guarantees at any granularity
    by_count = sharding.shard_costs(n, sharding.shard_ranges(n, world), data, lib)
    assert max(by_count) > max(cost) + 1.0, (by_count, cost)     # equal counts are off by more than a master block of text


def test_deal_edge_cases():
    """zmx_deal_master_blocks: fewer blocks than shards (empty shards at the end, never a gap), one block, equal costs
    (= equal counts), one dominant block."""
    import ctypes

    import oracle_lib as ol
    lib = ol.hosttest_library()
    deal = lib.zmx_deal_master_blocks
    deal.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    deal.restype = ctypes.c_int

    def run(cost, shards):
        c = (ctypes.c_double * max(1, len(cost)))(*cost)
        first = (ctypes.c_size_t * (shards + 1))()
        assert deal(c, len(cost), shards, first) == 0
        f = list(first)
        assert f[0] == 0 and f[-1] == len(cost) and all(a <= b for a, b in zip(f, f[1:]))
        return f

    assert run([1.0] * 8, 4) == [0, 2, 4, 6, 8]
    assert run([1.0] * 3, 8)[:4] == [0, 1, 2, 3]
    assert run([1.0], 3) == [0, 1, 1, 1]
    f = run([1, 1, 1, 20, 1, 1, 1, 1], 3)
    assert f[1] <= 3 < f[2] or f[1] == 3            # the dominant block has a shard (nearly) to itself
    assert all(b > a for a, b in zip(f, f[1:]))
    assert run([], 2) == [0, 0, 0]


def _worker_hetero(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import oracle_lib as ol
    from zopfli_amd import Context, ZopfliOptions, generate, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        host = ol.hosttest_library()
        data = generate("Z", 1200000) + generate("X", 1300000) + generate("T", 1500000)
        ctx = Context(0, host)
        out = sharding.gzip_sharded(ctx, ZopfliOptions(1), data, rank, world, torch.device("cpu"), dist)
        ctx.close()
        if rank == 0:
            with open(out_path, "wb") as f:
                f.write(out)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_heterogeneous_corpus_two_ranks_cost_dealt(tmp_path):
    """World size 2 over gloo on a heterogeneous corpus with cost-aware ranges (not the equal-count ones): the gathered
    and merged stream is the one-rank stream and the reference's."""
    import oracle_lib as ol
    from zopfli_amd import ZopfliOptions, api, generate, sharding

    data = generate("Z", 1200000) + generate("X", 1300000) + generate("T", 1500000)
    ranges = sharding.shard_ranges(len(data), 2, data, ol.hosttest_library())
    assert ranges != sharding.shard_ranges(len(data), 2)          # the dealing really is by cost here
    out_path = str(tmp_path / "sharded.gz")
    mp.spawn(_worker_hetero, args=(2, _free_port(), out_path), nprocs=2, join=True)
    with open(out_path, "rb") as f:
        sharded = f.read()
    assert gzip.decompress(sharded) == data
    assert sharded == api.compress(data, 0, ZopfliOptions(1), lib=ol.hosttest_library())
    if ol.have_ref():
        assert sharded == ol.ref_compress(data, 0, 1)
