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
