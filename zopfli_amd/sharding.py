"""Master-block sharding across one-process-per-GPU ranks.

ZopfliDeflate cuts its input into 1 000 000-byte master blocks that are compressed independently
(deflate.c:916-923; a master block reads only its own bytes and the 32 KiB before it), so a
stream shards by master block with no data-path collective.  Every rank compresses a contiguous
range of master blocks into position-independent bit chunks (zmx_deflate_range); the blobs are
gathered to rank 0 (RCCL on GPUs, gloo in the CPU tests) and merged at bit granularity
(zmx_chunks_merge), because deflate blocks are not byte aligned and only the last block of the
stream carries BFINAL.
"""
import zlib

MASTER_BLOCK = 1000000   # ZOPFLI_MASTER_BLOCK_SIZE, util.h:60
WINDOW = 32768           # ZOPFLI_WINDOW_SIZE
GZIP_HEADER = bytes([31, 139, 8, 0, 0, 0, 0, 0, 2, 3])   # gzip_container.c:90-101


def shard_ranges(insize, world, data=None, lib=None):
    """Contiguous master-block ranges [(start, end)] per rank; empty ranks get (n, n).

    With `data` (the stream's bytes, which every rank holds) the ranges are balanced by the library's estimate of each
    master block's cost (zmx_master_block_costs / zmx_deal_master_blocks, include/zopfli_amd.h: a function of the bytes
    alone, so every rank computes the same ranges and the in-process dealer of ZopfliCompress the same again); without,
    by count."""
    nmb = max(1, -(-insize // MASTER_BLOCK))
    if data is not None and world > 1 and nmb > world:
        import ctypes
        if lib is None:
            from . import api
            lib = api.library()
        cost = (ctypes.c_double * nmb)()
        fn = lib.zmx_master_block_costs
        fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
        fn.restype = ctypes.c_int
        buf = data if isinstance(data, bytes) else bytes(data)   # (c_char_p takes bytes only: a bytearray is copied)
        if fn(buf, insize, cost, nmb) != nmb:
            raise RuntimeError("zmx_master_block_costs failed")
        first = (ctypes.c_size_t * (world + 1))()
        deal = lib.zmx_deal_master_blocks
        deal.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        deal.restype = ctypes.c_int
        if deal(cost, nmb, world, first) != 0:
            raise RuntimeError("zmx_deal_master_blocks failed")
        return [(min(first[r] * MASTER_BLOCK, insize), min(first[r + 1] * MASTER_BLOCK, insize)) for r in range(world)]
    out = []
    for r in range(world):
        m0, m1 = nmb * r // world, nmb * (r + 1) // world
        out.append((min(m0 * MASTER_BLOCK, insize), min(m1 * MASTER_BLOCK, insize)))
    return out


def shard_costs(insize, ranges, data, lib=None):
    """The estimated cost of each of `ranges` (what shard_ranges balances), for reports and tests."""
    import ctypes
    if lib is None:
        from . import api
        lib = api.library()
    nmb = max(1, -(-insize // MASTER_BLOCK))
    cost = (ctypes.c_double * nmb)()
    fn = lib.zmx_master_block_costs
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t]
    fn.restype = ctypes.c_int
    if fn(data if isinstance(data, bytes) else bytes(data), insize, cost, nmb) != nmb:
        raise RuntimeError("zmx_master_block_costs failed")
    return [sum(cost[b] for b in range(s // MASTER_BLOCK, -(-e // MASTER_BLOCK))) for s, e in ranges]


def gather_bytes(blob, rank, world, device, dist):
    """Variable-size gather of one bytes object (or uint8 array) per rank to rank 0: sizes by
    all_gather, payload by one gather of max-padded uint8 tensors.  Returns the list (uint8 arrays
    when world > 1) on rank 0, None elsewhere."""
    import torch
    import numpy as np
    if world == 1:
        return [blob]
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if len(blob):
        src = blob if isinstance(blob, np.ndarray) else np.frombuffer(blob, dtype=np.uint8)
        buf[:len(blob)] = torch.from_numpy(src if src.flags.writeable else src.copy()).to(device)
    out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return [out[r][:sizes[r]].cpu().numpy() for r in range(world)]   # uint8 arrays: zmx_chunks_merge reads them in place


def crc32_combine(crc1, crc2, len2):
    """zlib's crc32_combine (GF(2) matrix method): CRC of A+B from CRC(A), CRC(B), len(B)."""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]

    if len2 <= 0:
        return crc1
    odd = [0xedb88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def gzip_sharded(ctx, options, data, rank, world, device, dist):
    """ZopfliGzipCompress of `data` (every rank holds it) sharded by master block over `world`
    ranks.  Returns the gzip stream on rank 0 (byte-identical to the single-GPU stream), None
    elsewhere."""
    ranges = shard_ranges(len(data), world, data, ctx.lib)
    start, end = ranges[rank]
    last_nonempty = max((r for r in range(world) if ranges[r][1] > ranges[r][0]), default=0)
    blob = b""
    crc = 0
    if end > start or (len(data) == 0 and rank == 0):
        base = max(0, start - WINDOW)
        ctx.set_input(data[base:end])        # own master blocks + the dictionary before them
        blob = ctx.deflate_range(options, start - base, end - base, 1 if rank == last_nonempty else 0)
        crc = zlib.crc32(data[start:end])
    blobs = gather_bytes(blob, rank, world, device, dist)
    meta = gather_bytes(crc.to_bytes(4, "little"), rank, world, device, dist)
    if rank != 0:
        return None
    total_crc = 0
    for r in range(world):
        total_crc = crc32_combine(total_crc, int.from_bytes(bytes(meta[r]), "little"), ranges[r][1] - ranges[r][0])
    trailer = total_crc.to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
    return ctx.merge([b for b in blobs if len(b)], GZIP_HEADER, trailer)
