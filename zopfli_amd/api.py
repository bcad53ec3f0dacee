"""ctypes mirror of include/zopfli_amd.h (reference names and argument meaning)."""
import ctypes
import os

from ._build import LIB

# (a variant build for A/B measurements and the -DZMX_EXPERIMENTS suite: tools/build_variant.py)
if os.environ.get("ZOPFLI_AMD_LIB"):
    LIB = os.environ["ZOPFLI_AMD_LIB"]

FORMAT_GZIP, FORMAT_ZLIB, FORMAT_DEFLATE = 0, 1, 2  # zopfli.h:70-74
CRC32, ADLER32 = 0, 1  # ZMX_CRC32, ZMX_ADLER32
ZMX_HIST = 320


class ZopfliOptions(ctypes.Structure):
    """zopfli.h:33-64; defaults as ZopfliInitOptions (util.c:28)."""
    _fields_ = [("verbose", ctypes.c_int), ("verbose_more", ctypes.c_int), ("numiterations", ctypes.c_int),
                ("blocksplitting", ctypes.c_int), ("blocksplittinglast", ctypes.c_int),
                ("blocksplittingmax", ctypes.c_int)]

    def __init__(self, numiterations=15, blocksplitting=1, blocksplittingmax=15, verbose=0, verbose_more=0):
        super().__init__(verbose, verbose_more, numiterations, blocksplitting, 0, blocksplittingmax)


class ZmxBlock(ctypes.Structure):
    _fields_ = [("instart", ctypes.c_uint64), ("inend", ctypes.c_uint64)]


_lib = None
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]
_libc.malloc.argtypes = [ctypes.c_size_t]
_libc.malloc.restype = ctypes.c_void_p
_libc.realloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
_libc.realloc.restype = ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_ubyte)


def _pow2ceil(n):
    cap = 1
    while cap < n:
        cap <<= 1
    return cap


def _owned_array(addr, size):
    """A numpy uint8 view of a malloc'ed buffer that frees it when the array is collected."""
    import weakref

    import numpy as np
    if size == 0:
        _libc.free(addr)
        return np.zeros(0, dtype=np.uint8)
    buf = (ctypes.c_ubyte * size).from_address(addr)
    weakref.finalize(buf, _libc.free, addr)
    return np.frombuffer(buf, dtype=np.uint8)


def bind(lib):
    """Declares the prototypes of include/zopfli_amd.h on a loaded library."""
    P, sz, vp = ctypes.POINTER, ctypes.c_size_t, ctypes.c_void_p
    opt = P(ZopfliOptions)
    lib.ZopfliInitOptions.argtypes = [opt]
    lib.ZopfliInitOptions.restype = None
    lib.ZopfliCompress.argtypes = [opt, ctypes.c_int, ctypes.c_char_p, sz, P(_u8p), P(sz)]
    lib.ZopfliCompress.restype = None
    for name in ("ZopfliGzipCompress", "ZopfliZlibCompress"):
        f = getattr(lib, name)
        f.argtypes = [opt, ctypes.c_char_p, sz, P(_u8p), P(sz)]
        f.restype = None
    lib.ZopfliDeflate.argtypes = [opt, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz, P(ctypes.c_ubyte), P(_u8p),
                                  P(sz)]
    lib.ZopfliDeflate.restype = None
    lib.ZopfliDeflatePart.argtypes = [opt, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, sz, sz, P(ctypes.c_ubyte),
                                      P(_u8p), P(sz)]
    lib.ZopfliDeflatePart.restype = None
    lib.zmx_device_count.restype = ctypes.c_int
    lib.zmx_last_error.restype = ctypes.c_char_p
    lib.zmx_last_error_class.restype = ctypes.c_int
    lib.zmx_host_cache_trim.restype = ctypes.c_size_t
    lib.zmx_ctx_create.argtypes = [ctypes.c_int, P(vp)]
    lib.zmx_ctx_destroy.argtypes = [vp]
    lib.zmx_ctx_destroy.restype = None
    lib.zmx_set_input.argtypes = [vp, ctypes.c_char_p, sz]
    lib.zmx_tables_build.argtypes = [vp, P(ZmxBlock), sz, P(vp)]
    lib.zmx_tables_build_matches.argtypes = [vp, P(ZmxBlock), sz, P(vp)]
    lib.zmx_tables_build_from.argtypes = [vp, vp, P(ZmxBlock), sz, P(vp)]
    lib.zmx_tables_free.argtypes = [vp, vp]
    lib.zmx_tables_trim.argtypes = [vp, vp]
    lib.zmx_tables_free.restype = None
    lib.zmx_lz77_greedy.argtypes = [vp, vp, ctypes.c_int, P(ctypes.c_uint32), P(ctypes.c_uint32)]
    lib.zmx_squeeze_run.argtypes = [vp, vp, P(ctypes.c_double), P(ctypes.c_double), P(ctypes.c_int32),
                                    P(ctypes.c_uint32), P(ctypes.c_uint32)]
    lib.zmx_store_download.argtypes = [vp, vp, sz, ctypes.c_int, P(ctypes.c_uint16), P(ctypes.c_uint16), sz]
    lib.zmx_find_longest_match.argtypes = [vp, vp, sz, sz, P(ctypes.c_uint16), P(ctypes.c_uint16),
                                           P(ctypes.c_uint16)]
    lib.zmx_length_array_download.argtypes = [vp, vp, sz, P(ctypes.c_uint16)]
    lib.zmx_hash_links_download.argtypes = [vp, vp, sz, P(ctypes.c_uint16), P(ctypes.c_uint16), P(ctypes.c_uint16)]
    lib.zmx_deflate_range.argtypes = [vp, opt, sz, sz, ctypes.c_int, P(_u8p), P(sz)]
    lib.zmx_checksum.argtypes = [vp, ctypes.c_int, sz, sz, P(ctypes.c_uint32)]
    lib.zmx_checksum_combine.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64]
    lib.zmx_checksum_combine.restype = ctypes.c_uint32
    lib.zmx_chunks_merge.argtypes = [P(vp), P(sz), sz, P(ctypes.c_ubyte), P(_u8p), P(sz)]
    lib.zmx_last_timing.argtypes = [P(ctypes.c_double)]
    lib.zmx_last_kernel_timing.argtypes = [P(ctypes.c_double)]
    lib.zmx_last_host_timing.argtypes = [P(ctypes.c_double)]
    lib.zmx_last_seg_stats.argtypes = [P(ctypes.c_double)]
    lib.zmx_last_match_timing.argtypes = [P(ctypes.c_double)]
    lib.zmx_set_match_kernel.argtypes = [ctypes.c_int]
    lib.zmx_dist_unique_id.argtypes = [ctypes.c_char_p]
    lib.zmx_dist_init.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, P(vp)]
    lib.zmx_dist_destroy.argtypes = [vp]
    lib.zmx_dist_destroy.restype = None
    lib.zmx_dist_comm_count.argtypes = [vp]
    lib.zmx_dist_gather.argtypes = [vp, vp, sz, P(vp), P(sz)]
    # (the harness reads the squeeze runs' phase times: last_timing()["dp_kernel"]; a plain caller of the library does not pay for them)
    lib.zmx_set_kernel_timing.argtypes = [ctypes.c_int]
    lib.zmx_set_kernel_timing.restype = None
    lib.zmx_set_kernel_timing(1)
    return lib


def library(path=None):
    """Loads libzopfli_amd.so (built in-tree by __graft_entry__.build()).  No fallback."""
    global _lib
    if path is not None:
        return bind(ctypes.CDLL(path))
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                               "zopfli_amd has no CPU fallback")
        _lib = bind(ctypes.CDLL(LIB))
    return _lib


def _take(out, size):
    data = ctypes.string_at(out, size.value)
    _libc.free(out)
    return data


def compress(data, fmt=FORMAT_GZIP, options=None, lib=None):
    """ZopfliCompress (zopfli_lib.c:28)."""
    lib = lib or library()
    options = options or ZopfliOptions()
    out, size = _u8p(), ctypes.c_size_t(0)
    lib.ZopfliCompress(ctypes.byref(options), fmt, data, len(data), ctypes.byref(out), ctypes.byref(size))
    return _take(out, size)


def deflate(data, btype=2, final=1, options=None, lib=None):
    """ZopfliDeflate (deflate.c:908) from bp = 0; returns (bytes, bp)."""
    lib = lib or library()
    options = options or ZopfliOptions()
    out, size, bp = _u8p(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
    lib.ZopfliDeflate(ctypes.byref(options), btype, final, data, len(data), ctypes.byref(bp), ctypes.byref(out),
                      ctypes.byref(size))
    return _take(out, size), bp.value


def deflate_part(data, instart, inend, btype=2, final=1, options=None, lib=None):
    """ZopfliDeflatePart (deflate.c:811) from bp = 0; returns (bytes, bp)."""
    lib = lib or library()
    options = options or ZopfliOptions()
    out, size, bp = _u8p(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
    lib.ZopfliDeflatePart(ctypes.byref(options), btype, final, data, instart, inend, ctypes.byref(bp),
                          ctypes.byref(out), ctypes.byref(size))
    return _take(out, size), bp.value


def last_timing(lib=None):
    lib = lib or library()
    t = (ctypes.c_double * 8)()
    lib.zmx_last_timing(t)
    keys = ["tables", "greedy", "squeeze", "cost_model", "split", "encode", "dp_kernel", "squeeze_launches"]
    d = dict(zip(keys, list(t)))
    k = (ctypes.c_double * 4)()
    lib.zmx_last_kernel_timing(k)
    d["wtab_kernel"], d["trace_kernel"] = k[0], k[2]
    h = (ctypes.c_double * 2)()
    lib.zmx_last_host_timing(h)
    d["download"], d["serialize"] = h[0], h[1]
    m = (ctypes.c_double * 4)()
    lib.zmx_last_match_timing(m)
    d["match_kernel"], d["hash_kernels"], d["table_builds"], d["positions_matched"] = m[0], m[1], m[2], m[3]
    w = (ctypes.c_double * 3)()
    lib.zmx_last_match_walk(w)
    d["skip_walk_lane_iterations"], d["skip_walk_wave_iterations"], d["skip_walk_positions"] = w[0], w[1], w[2]
    return d


def last_seg_stats(lib=None):
    """zmx_last_seg_stats: how the chain's tasks of the last call fared."""
    lib = lib or library()
    t = (ctypes.c_double * 8)()
    lib.zmx_last_seg_stats(t)
    keys = ["tasks", "accepted", "rerun_state", "rerun_level", "rerun_tie", "positions_rerun", "rerun_values", "positions"]
    return dict(zip(keys, list(t)))


class Context:
    """The zmx_* device layer: one HIP device with a resident input."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or library()
        self.handle = ctypes.c_void_p()
        self._input = None
        if self.lib.zmx_ctx_create(device, ctypes.byref(self.handle)) != 0:
            raise RuntimeError("zmx_ctx_create: " + self.error())

    def error(self):
        return (self.lib.zmx_last_error() or b"").decode()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.error()}")

    def close(self):
        if self.handle:
            self.lib.zmx_ctx_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def set_input(self, data):
        self._input = data
        self._check(self.lib.zmx_set_input(self.handle, data, len(data)), "zmx_set_input")

    def build_tables(self, blocks, parent=None, matches_only=False):
        """zmx_tables_build, zmx_tables_build_matches (no DP rows: for the greedy pass and as a parent), or
        zmx_tables_build_from when `parent` (Tables over enclosing blocks) is given."""
        arr = (ZmxBlock * len(blocks))(*[ZmxBlock(s, e) for s, e in blocks])
        t = ctypes.c_void_p()
        if matches_only:
            self._check(self.lib.zmx_tables_build_matches(self.handle, arr, len(blocks), ctypes.byref(t)),
                        "zmx_tables_build_matches")
        elif parent is None:
            self._check(self.lib.zmx_tables_build(self.handle, arr, len(blocks), ctypes.byref(t)), "zmx_tables_build")
        else:
            self._check(self.lib.zmx_tables_build_from(self.handle, parent.handle, arr, len(blocks), ctypes.byref(t)),
                        "zmx_tables_build_from")
        return Tables(self, t, list(blocks))

    def checksum(self, kind, begin, end):
        """zmx_checksum: CRC-32 (kind CRC32) or Adler-32 (ADLER32) of resident bytes [begin, end), on the device."""
        v = ctypes.c_uint32(0)
        self._check(self.lib.zmx_checksum(self.handle, kind, begin, end, ctypes.byref(v)), "zmx_checksum")
        return v.value

    def deflate_range(self, options, instart, inend, final=1, as_array=False):
        """zmx_deflate_range: serialised chunks of ZopfliDeflate over resident bytes [instart, inend).
        as_array=True returns a numpy view of the library's malloc'ed blob (no copy)."""
        blob, size = _u8p(), ctypes.c_size_t(0)
        self._check(self.lib.zmx_deflate_range(self.handle, ctypes.byref(options), instart, inend, final,
                                               ctypes.byref(blob), ctypes.byref(size)), "zmx_deflate_range")
        if as_array:
            return _owned_array(ctypes.cast(blob, ctypes.c_void_p).value, size.value)
        return _take(blob, size)

    def merge(self, blobs, prefix=b"", trailer=b"", as_array=False):
        """zmx_chunks_merge after `prefix` (e.g. a gzip header), then `trailer`: prefix + deflate
        stream + trailer.  Blobs may be bytes or uint8 numpy arrays (not copied); the stream is
        assembled in one malloc'ed buffer (the (out, outsize) convention of the reference, capacity
        = next power of two) and returned as bytes, or as a numpy view of it with as_array=True."""
        import numpy as np
        n = len(blobs)
        keep = [b if isinstance(b, np.ndarray) else np.frombuffer(b, dtype=np.uint8) for b in blobs]
        arr = (ctypes.c_void_p * n)(*[k.ctypes.data for k in keep])
        sizes = (ctypes.c_size_t * n)(*[k.size for k in keep])
        out, size, bp = _u8p(), ctypes.c_size_t(0), ctypes.c_ubyte(0)
        if prefix:
            addr = _libc.malloc(_pow2ceil(len(prefix)))
            ctypes.memmove(addr, prefix, len(prefix))
            out, size = ctypes.cast(addr, _u8p), ctypes.c_size_t(len(prefix))
        self._check(self.lib.zmx_chunks_merge(arr, sizes, n, ctypes.byref(bp), ctypes.byref(out),
                                              ctypes.byref(size)), "zmx_chunks_merge")
        addr, total = ctypes.cast(out, ctypes.c_void_p).value, size.value
        if trailer:
            new = total + len(trailer)
            if addr is None or _pow2ceil(new) > (_pow2ceil(total) if total else 0):
                addr = _libc.realloc(addr, _pow2ceil(new))
            ctypes.memmove(addr + total, trailer, len(trailer))
            total = new
        if as_array:
            return _owned_array(addr, total)
        data = ctypes.string_at(addr, total) if total else b""
        _libc.free(addr)
        return data


class Dist:
    """zmx_dist_*: the ranks' blobs gathered to rank 0 over RCCL by the library itself (one process
    per GPU).  `unique_id()` on rank 0, handed to the others by the launcher, then `Dist(ctx, rank,
    world, id)` on every rank."""

    @staticmethod
    def unique_id(lib=None):
        lib = lib or library()
        buf = ctypes.create_string_buffer(128)
        if lib.zmx_dist_unique_id(buf) != 0:
            raise RuntimeError("zmx_dist_unique_id: " + (lib.zmx_last_error() or b"").decode())
        return buf.raw

    def __init__(self, ctx, rank, world, uid):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.handle = ctypes.c_void_p()
        ctx._check(ctx.lib.zmx_dist_init(ctx.handle, rank, world, uid, ctypes.byref(self.handle)), "zmx_dist_init")

    def gather(self, blob):
        """Returns the list of the ranks' blobs (uint8 arrays, views of one malloc'ed buffer) on rank 0,
        None elsewhere."""
        import numpy as np
        src = blob if isinstance(blob, np.ndarray) else np.frombuffer(blob, dtype=np.uint8)
        out = ctypes.c_void_p()
        sizes = (ctypes.c_size_t * self.world)()
        self.ctx._check(self.ctx.lib.zmx_dist_gather(self.handle, src.ctypes.data if src.size else None, src.size,
                                                     ctypes.byref(out), sizes), "zmx_dist_gather")
        if self.rank != 0:
            return None
        total = sum(sizes)
        whole = _owned_array(out.value, total)
        parts, off = [], 0
        for r in range(self.world):
            parts.append(whole[off:off + sizes[r]])
            off += sizes[r]
        return parts

    def comm_count(self):
        """ncclCommCount of the communicator: the ranks RCCL itself saw."""
        return int(self.ctx.lib.zmx_dist_comm_count(self.handle))

    def close(self):
        if self.handle:
            self.ctx.lib.zmx_dist_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class CostStores:
    """zmx_cost_stores: LZ77 symbol sequences resident on the device with their sampled prefix histograms;
    block_costs = ZopfliCalculateBlockSizeAutoType (deflate.c:610-621) of many ranges at once, on the device."""

    def __init__(self, ctx, handle, sizes):
        self.ctx, self.handle, self.sizes = ctx, handle, list(sizes)

    @staticmethod
    def from_host(ctx, stores):
        """stores = [(litlens, dists)] (lz77.h:44-49 convention)."""
        import numpy as np
        n = len(stores)
        lls = [np.ascontiguousarray(s[0], dtype=np.uint16) for s in stores]
        dds = [np.ascontiguousarray(s[1], dtype=np.uint16) for s in stores]
        pl = (ctypes.c_void_p * n)(*[a.ctypes.data for a in lls])
        pd = (ctypes.c_void_p * n)(*[a.ctypes.data for a in dds])
        ns = (ctypes.c_size_t * n)(*[len(a) for a in lls])
        fn = ctx.lib.zmx_cost_stores_create_host
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        h = ctypes.c_void_p()
        ctx._check(fn(ctx.handle, n, ctypes.cast(pl, ctypes.c_void_p), ctypes.cast(pd, ctypes.c_void_p),
                      ctypes.cast(ns, ctypes.c_void_p), ctypes.byref(h)), "zmx_cost_stores_create_host")
        return CostStores(ctx, h, [len(a) for a in lls])

    @staticmethod
    def from_tables(tables, sequences):
        """sequences = [[(block, slot, nsym), ...], ...]: every sequence the concatenation of device stores of `tables`."""
        ctx = tables.ctx
        first, blk, slot, nsym = [0], [], [], []
        for seq in sequences:
            for b, s, k in seq:
                blk.append(int(b)); slot.append(int(s)); nsym.append(int(k))
            first.append(len(blk))
        n, np_ = len(sequences), len(blk)
        fn = ctx.lib.zmx_cost_stores_create
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p, ctypes.c_void_p]
        a_first = (ctypes.c_size_t * (n + 1))(*first)
        a_blk = (ctypes.c_size_t * max(np_, 1))(*blk)
        a_slot = (ctypes.c_int32 * max(np_, 1))(*slot)
        a_nsym = (ctypes.c_size_t * max(np_, 1))(*nsym)
        h = ctypes.c_void_p()
        ctx._check(fn(ctx.handle, tables.handle, n, ctypes.cast(a_first, ctypes.c_void_p), ctypes.cast(a_blk, ctypes.c_void_p),
                      ctypes.cast(a_slot, ctypes.c_void_p), ctypes.cast(a_nsym, ctypes.c_void_p), ctypes.byref(h)),
                   "zmx_cost_stores_create")
        return CostStores(ctx, h, [sum(k for _, _, k in seq) for seq in sequences])

    def block_costs(self, ranges):
        """ranges = [(sequence, lstart, lend)] -> float64 array of block sizes in bits."""
        import numpy as np
        r = np.ascontiguousarray(np.asarray(ranges, dtype=np.uint32).reshape(-1, 3))
        out = np.zeros(len(r), dtype=np.float64)
        fn = self.ctx.lib.zmx_block_costs
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        self.ctx._check(fn(self.ctx.handle, self.handle, len(r), r.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p)), "zmx_block_costs")
        return out

    def positions(self, pairs):
        """pairs = [(sequence, index)] -> the bytes that symbols [0, index) of the sequence stand for (zmx_cost_positions)."""
        import numpy as np
        q = np.ascontiguousarray(np.asarray(pairs, dtype=np.uint32).reshape(-1, 2))
        out = np.zeros(len(q), dtype=np.uint64)
        fn = self.ctx.lib.zmx_cost_positions
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        self.ctx._check(fn(self.ctx.handle, self.handle, len(q), q.ctypes.data_as(ctypes.c_void_p),
                           out.ctypes.data_as(ctypes.c_void_p)), "zmx_cost_positions")
        return out

    def free(self):
        if self.handle:
            fn = self.ctx.lib.zmx_cost_stores_free
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            fn(self.ctx.handle, self.handle)
            self.handle = None


class Tables:
    def __init__(self, ctx, handle, blocks):
        self.ctx, self.handle, self.blocks = ctx, handle, blocks

    def free(self):
        if self.handle:
            self.ctx.lib.zmx_tables_free(self.ctx.handle, self.handle)
            self.handle = None

    def trim(self):
        """zmx_tables_trim: everything but the two stores goes back to the pool."""
        self.ctx._check(self.ctx.lib.zmx_tables_trim(self.ctx.handle, self.handle), "zmx_tables_trim")

    def greedy(self, slot=0):
        import numpy as np
        nb = len(self.blocks)
        nsym = np.zeros(nb, dtype=np.uint32)
        hist = np.zeros((nb, ZMX_HIST), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zmx_lz77_greedy(self.ctx.handle, self.handle, slot,
                                                     nsym.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                     hist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))),
                        "zmx_lz77_greedy")
        return nsym, hist

    def squeeze_run(self, cost, mincost, slot):
        import numpy as np
        nb = len(self.blocks)
        cost = np.ascontiguousarray(cost, dtype=np.float64)
        mincost = np.ascontiguousarray(mincost, dtype=np.float64)
        slot = np.ascontiguousarray(slot, dtype=np.int32)
        nsym = np.zeros(nb, dtype=np.uint32)
        hist = np.zeros((nb, ZMX_HIST), dtype=np.uint32)
        self.ctx._check(self.ctx.lib.zmx_squeeze_run(self.ctx.handle, self.handle,
                                                     cost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                     mincost.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                     slot.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                     nsym.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                                     hist.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))),
                        "zmx_squeeze_run")
        return nsym, hist

    def store(self, block, slot, nsym):
        import numpy as np
        ll = np.zeros(max(int(nsym), 1), dtype=np.uint16)
        dd = np.zeros(max(int(nsym), 1), dtype=np.uint16)
        self.ctx._check(self.ctx.lib.zmx_store_download(self.ctx.handle, self.handle, block, slot,
                                                        ll.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                                        dd.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), int(nsym)),
                        "zmx_store_download")
        return ll[:int(nsym)], dd[:int(nsym)]

    def verify_stores(self, blocks, slots, nsyms):
        """zmx_verify_stores: raises RuntimeError naming the first symbol that does not stand for the input's bytes."""
        n = len(blocks)
        fn = self.ctx.lib.zmx_verify_stores
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        b = (ctypes.c_size_t * n)(*map(int, blocks))
        s = (ctypes.c_int32 * n)(*map(int, slots))
        k = (ctypes.c_size_t * n)(*map(int, nsyms))
        self.ctx._check(fn(self.ctx.handle, self.handle, n, ctypes.cast(b, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                           ctypes.cast(k, ctypes.c_void_p)), "zmx_verify_stores")

    def encode_blocks(self, jobs, codes):
        """zmx_encode_blocks: jobs = [(block, slot, nsym, bit_start, nbits)], codes = uint32 [njobs, 320]
        (reversed code | length << 16).  Returns one bytes object per job (header bits zero)."""
        import numpy as np

        class Job(ctypes.Structure):
            _fields_ = [("block", ctypes.c_uint32), ("slot", ctypes.c_int32), ("nsym", ctypes.c_uint32),
                        ("bit_start", ctypes.c_uint32), ("nbits", ctypes.c_uint64)]
        n = len(jobs)
        arr = (Job * n)(*[Job(*map(int, j)) for j in jobs])
        codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(n, 320)
        bufs = [np.zeros((int(j[3]) + int(j[4]) + 7) // 8 + 8, dtype=np.uint8) for j in jobs]
        ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
        fn = self.ctx.lib.zmx_encode_blocks
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self.ctx._check(fn(self.ctx.handle, self.handle, n, ctypes.cast(arr, ctypes.c_void_p),
                           codes.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ptrs, ctypes.c_void_p)), "zmx_encode_blocks")
        return [bytes(b[:(int(j[3]) + int(j[4]) + 7) // 8]) for b, j in zip(bufs, jobs)]

    def find_longest_match(self, block, pos):
        import numpy as np
        sub = np.zeros(259, dtype=np.uint16)
        d, l = ctypes.c_uint16(0), ctypes.c_uint16(0)
        self.ctx._check(self.ctx.lib.zmx_find_longest_match(self.ctx.handle, self.handle, block, pos,
                                                            sub.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)),
                                                            ctypes.byref(d), ctypes.byref(l)),
                        "zmx_find_longest_match")
        return l.value, d.value, sub

    def match_digest(self):
        """zmx_match_digest: a digest of every match record of these tables."""
        out = (ctypes.c_uint64 * 2)()
        fn = self.ctx.lib.zmx_match_digest
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        self.ctx._check(fn(self.ctx.handle, self.handle, out), "zmx_match_digest")
        return (int(out[0]), int(out[1]))

    def hash_links(self, block):
        """same[], prev1[], prev2[] of the block's positions from windowstart on (zmx_hash_links_download)."""
        import numpy as np
        s, e = self.blocks[block]
        n = e - max(0, s - 32768)
        arrs = [np.zeros(max(n, 1), dtype=np.uint16) for _ in range(3)]
        ptrs = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)) for a in arrs]
        self.ctx._check(self.ctx.lib.zmx_hash_links_download(self.ctx.handle, self.handle, block, *ptrs),
                        "zmx_hash_links_download")
        return [a[:n] for a in arrs]

    def length_array(self, block):
        import numpy as np
        s, e = self.blocks[block]
        la = np.zeros(e - s + 1, dtype=np.uint16)
        self.ctx._check(self.ctx.lib.zmx_length_array_download(self.ctx.handle, self.handle, block,
                                                               la.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))),
                        "zmx_length_array_download")
        return la
