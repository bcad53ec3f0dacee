"""bench.py — zopfli hot path on MI355X.

One step = one pass of the hot path over the workload, match tables, greedy seeds, numiterations squeeze
runs, host cost model, block choice, encoding, merge.  Two ways in, both measured at N = 1:

  value            the reference's entry point: ZopfliCompress(options, GZIP, host buffer, size, &out, &outsize) of
                   libzopfli_amd.so — what a drop-in user calls (zopfli_lib.c:28-42, zopfli_bin.c:99-127; BASELINE.md
                   section 3: "wall-clock around ZopfliCompress only").  Host buffer in, malloc'ed gzip stream out;
                   the copy to HBM and the device context's reuse are inside the clock.
  value_resident   the same work on an input that is already resident in HBM (zmx_set_input outside the clock,
                   zmx_deflate_range + zmx_chunks_merge): the kernels' own rate, and the only form N > 1 has (every
                   rank compresses its shard, the blobs are gathered to rank 0 over RCCL).

`blocksplitting1` repeats the entry-point measurement with the reference's DEFAULT options (blocksplitting = 1,
util.c:28-35: BASELINE configs[2]) so that the number a real caller of ZopfliInitOptions gets is in the line too.

Workload at N=1 (BASELINE.json configs[1]): 100 000 000 bytes of the seeded text-like class T
(enwik8 stand-in), numiterations=15, blocksplitting=0 (one deflate block per 1 MB
master block).  At N>1 every rank compresses its own 100 MB shard of an N x 100 MB input
(weak scaling): the shards are consecutive ranges of master blocks of ONE stream, rank r works
on master blocks [100 r, 100 r + 100) with the preceding 32 KiB as dictionary, the chunk blobs
are gathered to rank 0 over RCCL and merged into one gzip stream.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import gzip
import hashlib
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MB = 1000000

# what the synthetic classes of zopfli_amd/csrc/tools/datagen.c stand for
CLASS_NAMES = {"T": "text-like, the enwik8 stand-in", "X": "markup", "R": "random bytes", "P": "PNG-like filtered scanlines",
               "B": "two symbols", "Z": "long runs of equal bytes", "M": "mixed corpus, the Silesia stand-in"}


def baseline_config_name(cls, size, numiterations, blocksplitting):
    """Which BASELINE.json config a run is (or that it is none): configs[1] = 100 MB of text, n = 15, one block stream;
    configs[2] = the same with ZopfliBlockSplit on; configs[3] = ~200 MB mixed corpus, n = 50, block splitting on."""
    if cls == "T" and size == 100 * MB and numiterations == 15:
        return "BASELINE configs[1]" if blocksplitting == 0 else "BASELINE configs[2] (on the GPUs of this run)"
    if cls == "M" and size >= 200 * MB and numiterations == 50 and blocksplitting == 1:
        return "BASELINE configs[3] (on the GPUs of this run)"
    return "not a BASELINE config (a class / size line)"

WINDOW = 32768
# rocprofv3 PMC passes per class of the 100 MB, n = 15, blocksplitting 0 workload (tools/collect_profiles.sh; quoted only
# when taken on this build's device sources)
# keyed by (class, bytes, numiterations, blocksplitting)
PMC_PROFILES = {("T", 100 * MB, 15, 0): "r06_bench100MB_pmc.json", ("Z", 100 * MB, 15, 0): "r06_classZ100MB_pmc.json",
                ("M", 200 * MB, 50, 1): "r06_config3_M200_n50_pmc.json"}
SHADER_CLOCK_HZ = 2.4e9   # MI355X peak engine clock; s_memtime counts at this rate (measured)


def device_source_hash():
    """SHA-256 (16 hex digits) over the device sources: a committed PMC profile is quoted only for the build it was
    taken on (tools/pmc_summary.py stores the same hash in the profile)."""
    d = os.path.join(ROOT, "zopfli_amd", "csrc", "device")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(sample, options):
    """The real reference (oracle/_ref, -O3 -DNDEBUG) on one host core, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    if not ol.have_ref():
        return None
    t0 = time.perf_counter()
    out = ol.ref_compress(sample, 0, options.numiterations, options.blocksplitting, options.blocksplittingmax)
    dt = time.perf_counter() - t0
    return {"value": round(len(sample) / MB / dt, 4), "unit": "MB/s", "cores": 1, "kind": "reference",
            "sample": f"first {len(sample)} bytes of the workload, same ZopfliOptions, gzip, "
                      f"oracle/_ref/libzopfli_ref.so (gcc -O3 -DNDEBUG), {dt:.1f} s, {len(out)} bytes out"}, out


def _ref_piece(task):
    """One master block through the reference's ZopfliDeflatePart, in a worker process."""
    data, start, end, n, split, smax = task
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    t0 = time.perf_counter()
    out, _bp = ol.ref_deflate_part(data, start, end, 2, 0, n, split)
    return len(out), time.perf_counter() - t0


def usable_cpus():
    """The CPUs this process may use: affinity and cgroup quota, not the box's hardware threads."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_all_cores(shard, options):
    """SURVEY 8(d) secondary baseline: the real reference on EVERY host core — ZopfliDeflatePart per
    master block (deflate.c:916-923: the blocks are independent), one worker process per core, two
    master blocks each, so the whole box works for a few seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import multiprocessing as mp

    import oracle_lib as ol
    if not ol.have_ref():
        return None
    cores = usable_cpus()
    nblocks = min(len(shard) // MB, 2 * cores)
    if nblocks < 1:
        return None
    tasks = []
    for b in range(nblocks):
        lo = max(0, b * MB - WINDOW)
        tasks.append((shard[lo:(b + 1) * MB], b * MB - lo, (b + 1) * MB - lo, options.numiterations,
                      options.blocksplitting, options.blocksplittingmax))
    workers = min(cores, nblocks)
    with mp.get_context("fork").Pool(workers) as pool:
        t0 = time.perf_counter()
        res = pool.map(_ref_piece, tasks, chunksize=1)
        dt = time.perf_counter() - t0
    return {"value": round(nblocks / dt, 3), "unit": "MB/s", "cores": workers, "kind": "reference",
            "sample": f"first {nblocks} master blocks of the workload, ZopfliDeflatePart per master block in "
                      f"{workers} worker processes (one per CPU this container may use: {os.cpu_count()} hardware "
                      f"threads on the box, affinity + cgroup quota allow {cores}), {dt:.1f} s wall, "
                      f"{sum(r[1] for r in res):.0f} s of core time, {sum(r[0] for r in res)} bytes out"}


def measured_copy_gbs(torch, device):
    """What a plain device copy reaches on this GPU (read + write bytes per second), beside the 8 TB/s on paper."""
    try:
        n = 1 << 30
        a = torch.empty(n, dtype=torch.uint8, device=device)
        b = torch.empty(n, dtype=torch.uint8, device=device)
        b.copy_(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        return round(5 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    except Exception:
        return None


def _ref_small(task):
    """One small file through the reference's ZopfliCompress, in a worker process."""
    data, n = task
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    return len(ol.ref_compress(data, 0, n))


def small_files_line(lib, api, generate, ZopfliOptions, numiterations, sets, callers=(1, 3, 16), with_reference=True):
    """What zopfli is mostly used on: MANY SMALL FILES (zopfli.h:40-44: "10 - 15 iterations for small files"; zopflipng takes
    15 iterations below 200 000 bytes, zopflipng_lib.cc:57-58).  Every file is one ZopfliCompress call with the reference's
    default options, K caller threads at a time (the library's contexts: ZOPFLI_AMD_LANES per device; a caller beyond
    them waits for a free one) — aggregate MB/s over the whole set, beside the reference on every host core this process
    may use (one file per worker process at a time).  Class T and X files alternate; every stream is checked to
    round-trip, the first four of each set against the reference's bytes."""
    import concurrent.futures as cf
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    opts = ZopfliOptions(numiterations, 1, 15)
    out = {"options": "the reference's defaults (blocksplitting=1, blocksplittingmax=15), numiterations=%d, gzip" % numiterations,
           "kernel_timing": "off, as for any caller of the library that does not ask for the phase times (zmx_set_kernel_timing: "
                            "the events are a third of a squeeze run's runtime calls); the lines above run with it on",
           "sets": []}
    lib.zmx_set_kernel_timing(0)
    for count, size in sets:
        files = [generate("TX"[i & 1], size, seed=1000 + i) for i in range(count)]
        total = count * size / MB
        rec = {"files": count, "bytes_each": size, "classes": "T and X alternating, seeds 1000 ...", "callers": {}}

        def one(d):
            return api.compress(d, api.FORMAT_GZIP, opts, lib=lib)
        for f in files[:3]:
            one(f)       # (warm: the first contexts exist)
        ok = True
        for k in callers:
            with cf.ThreadPoolExecutor(k) as ex:
                list(ex.map(one, files[:min(len(files), 2 * k)]))      # (warm: the k-th context exists)
                t0 = time.perf_counter()
                res = list(ex.map(one, files))
                dt = time.perf_counter() - t0
            ok = ok and all(gzip.decompress(r) == f for r, f in zip(res[:16], files[:16]))
            rec["callers"][str(k)] = {"value": round(total / dt, 3), "unit": "MB/s", "ms_per_file": round(dt / count * 1e3, 3)}
            last = res
        rec["roundtrip_ok"] = ok
        if with_reference and ol.have_ref():
            rec["bitexact_vs_reference_first4"] = all(last[i] == ol.ref_compress(files[i], 0, numiterations) for i in range(min(4, count)))
            cores = usable_cpus()
            nref = min(count, 2 * cores)
            with mp.get_context("fork").Pool(min(cores, nref)) as pool:
                t0 = time.perf_counter()
                pool.map(_ref_small, [(f, numiterations) for f in files[:nref]], chunksize=1)
                dtr = time.perf_counter() - t0
            rec["reference_all_cores"] = {"value": round(nref * size / MB / dtr, 3), "unit": "MB/s", "cores": min(cores, nref),
                                          "sample": "the first %d files, one per worker process at a time, %.1f s" % (nref, dtr)}
            best = max(v["value"] for v in rec["callers"].values())
            rec["best_vs_reference_all_cores"] = round(best / rec["reference_all_cores"]["value"], 2)
        out["sets"].append(rec)
    lib.zmx_set_kernel_timing(1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100 * MB, help="bytes per GPU (default: the 100 MB workload)")
    ap.add_argument("--blocksplitting", type=int, default=None, help="0 = configs[1] (the default at N = 1), 1 = configs[2] "
                    "(the default of the N > 1 headline)")
    ap.add_argument("--cls", default="T", choices=list("TXRZBPM"),
                    help="synthetic input class (zopfli_amd/csrc/tools/datagen.c): T text-like (default, enwik8 "
                         "stand-in), X markup-like, M mixed corpus (Silesia stand-in), ...")
    ap.add_argument("--numiterations", type=int, default=15)
    ap.add_argument("--cpu-sample", type=int, default=8 * MB)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo "
                    "exercises the same sharding/gather/merge code where only one GPU is visible)")
    ap.add_argument("--gather", default="rccl-c", choices=["rccl-c", "torch"],
                    help="who gathers the ranks' blobs at N>1: the library's own RCCL gather (zmx_dist_*, dist.cc; "
                         "falls back to torch.distributed if it cannot start) or torch.distributed's gather")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N>1: strong = --size bytes in total (BASELINE configs[2]: one 100 MB stream sharded by master "
                         "block), weak = --size bytes per GPU.  Default at N > 1: the strong line with the reference's "
                         "default block splitting is the headline and the weak line (blocksplitting 0, 100 MB per GPU) "
                         "rides along under `weak`")
    ap.add_argument("--device-index", type=int, default=None, help="HIP device of this rank (default LOCAL_RANK)")
    ap.add_argument("--entry", default="both", choices=["both", "zopfli_compress", "resident"],
                    help="N = 1: which way into the library is timed (default both: `value` = ZopfliCompress, "
                         "`value_resident` = resident input); N > 1 always times the sharded resident path")
    ap.add_argument("--no-blocksplitting1", action="store_true", help="skip the extra line with blocksplitting = 1")
    ap.add_argument("--file", default=None, help="a real corpus instead of the synthetic class: the file's bytes (the first "
                    "--size x N of them if it is longer) are the workload; `data` in the line names it")
    ap.add_argument("--devices", default=None, help="N = 1: ZOPFLI_AMD_DEVICES for the ZopfliCompress entry (e.g. 0,0: the "
                    "in-process multi-device path on one GPU)")
    ap.add_argument("--no-small-files", action="store_true", help="N = 1: skip the `small_files` line (many 64 KiB / 1 MB files through "
                    "1 / 3 / 16 concurrent ZopfliCompress callers)")
    ap.add_argument("--small-files-only", action="store_true", help="N = 1: only the `small_files` measurement, on a bigger set")
    ap.add_argument("--no-in-process", action="store_true", help="N > 1: skip the extra measurement of the same total input "
                    "through ZopfliCompress in rank 0's process with ZOPFLI_AMD_DEVICES = N (what a drop-in caller gets)")
    args = ap.parse_args()
    if args.devices:
        os.environ["ZOPFLI_AMD_DEVICES"] = args.devices

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: start the N ranks ourselves (one process per GPU, 127.0.0.1 rendezvous)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import copy

    import torch
    import torch.distributed as dist

    from zopfli_amd import Context, ZopfliOptions, api, generate, sharding

    dev_index = local_rank if args.device_index is None else args.device_index
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)
            device = torch.device("cpu")   # payload tensors of the gather live where the backend can reach them

    # one process per GPU shares the host cores: the library's worker pool (<= 64 threads, read once
    # when it starts) is sized so that the ranks of this node do not oversubscribe them
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > 1 and "ZOPFLI_AMD_THREADS" not in os.environ:
        os.environ["ZOPFLI_AMD_THREADS"] = str(max(8, min(64, (os.cpu_count() or 64) // local_world)))
    lib = api.library()

    # ---- the gather: the library's own RCCL gather unless it cannot start (one communicator for the whole run)
    cdist = None
    comm_ctx = None
    gather_kind = "none (one rank)"
    if world > 1:
        gather_kind = "torch.distributed " + args.backend
        if args.gather == "rccl-c" and args.backend == "nccl":
            from zopfli_amd import Dist
            comm_ctx = Context(dev_index, lib)
            err = ""
            uid = [None]
            try:
                if rank == 0:
                    uid[0] = Dist.unique_id(lib)
            except RuntimeError as e:   # e.g. librccl missing
                err = str(e)
            dist.broadcast_object_list(uid, src=0)
            ok = 0
            if uid[0] is not None:
                try:
                    cdist = Dist(comm_ctx, rank, world, uid[0])
                    ok = 1 if cdist.comm_count() == world else 0   # RCCL itself must have seen N ranks
                    if not ok:
                        err = "ncclCommCount = %d, expected %d" % (cdist.comm_count(), world)
                except RuntimeError as e:
                    err = str(e)
            flag = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                gather_kind = ("library RCCL (zmx_dist_gather: ncclAllGather sizes + grouped ncclSend/ncclRecv; "
                               "ncclCommCount = %d)" % world)
            else:
                if cdist is not None:
                    cdist.close()
                cdist = None
                gather_kind += " (library RCCL gather unavailable: %s)" % err

    if args.small_files_only:
        print(json.dumps({"small_files": small_files_line(lib, api, generate, ZopfliOptions, args.numiterations,
                                                          [(1000, 65536), (200, 1000000)])}), flush=True)
        return

    def run_one(args):
        """One measured configuration (args.scaling / args.blocksplitting resolved); the JSON line on rank 0."""
        return _run_one(args, rank, world, dev_index, device, torch, dist, lib, cdist, gather_kind)

    if world == 1 or args.scaling is not None:
        a = copy.copy(args)
        a.scaling = a.scaling or "weak"
        a.blocksplitting = 0 if a.blocksplitting is None else a.blocksplitting
        line = run_one(a)
    else:
        # N > 1 as the driver types it: BASELINE configs[2] is the headline — ONE stream of --size bytes, the reference's
        # default block splitting, its master blocks sharded over the N ranks (strong scaling) — and the weak line
        # (--size bytes per GPU, blocksplitting 0: N x configs[1]) rides along
        a = copy.copy(args)
        a.scaling = "strong"
        a.blocksplitting = 1 if a.blocksplitting is None else a.blocksplitting
        line = run_one(a)
        w = copy.copy(args)
        w.scaling = "weak"
        w.blocksplitting = 0
        w.no_in_process = True
        w.no_cpu_baseline = True
        wline = run_one(w)
        if rank == 0:
            line["weak"] = {k: wline[k] for k in ("value", "unit", "ms_per_step", "scaling", "steps", "config", "output_bytes",
                                                  "roundtrip_ok", "roofline", "roofline_match", "breakdown_s_per_step")}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if cdist is not None:
        cdist.close()
    if comm_ctx is not None:
        comm_ctx.close()
    if world > 1:
        dist.destroy_process_group()


def _run_one(args, rank, world, dev_index, device, torch, dist, lib, cdist, gather_kind):
    from zopfli_amd import Context, ZopfliOptions, api, generate, sharding
    options = ZopfliOptions(args.numiterations, args.blocksplitting, 15)
    size = args.size
    strong = args.scaling == "strong" and world > 1

    # ---- synthetic input.  weak: shard r = class `cls`, seed = the class's default seed + r, the shards
    #      being consecutive ranges of master blocks of ONE stream.  strong: one stream of `size` bytes
    #      (seed = the default), rank r taking the master blocks sharding.shard_ranges gives it.
    from zopfli_amd.datagen import DEFAULT_SEED
    seed0 = DEFAULT_SEED[args.cls]
    corpus = None
    if args.file:
        with open(args.file, "rb") as f:
            corpus = f.read()
        per = len(corpus) // world if not (args.scaling == "strong" and world > 1) else len(corpus)
        if world > 1:
            per -= per % MB
        size = min(size, per) if "--size" in sys.argv else per
        if size <= 0:
            raise SystemExit("--file: too short for %d ranks of whole master blocks" % world)
    if strong:
        whole = corpus[:size] if corpus is not None else generate(args.cls, size, seed=seed0)
        # (cost-balanced ranges from the bytes: zmx_master_block_costs — every rank computes the same)
        ranges = sharding.shard_ranges(size, world, whole, lib)
        s0, s1 = ranges[rank]
        shard = whole[s0:s1]
        prefix = whole[max(0, s0 - WINDOW):s0]
        total_bytes = size
        nonempty = [r for r in range(world) if ranges[r][1] > ranges[r][0]]
        last_rank = nonempty[-1] if nonempty else 0
        del whole
    else:
        assert size % MB == 0 or world == 1, "shards must be whole master blocks"
        if corpus is not None:
            shard = corpus[rank * size:(rank + 1) * size]
            prefix = corpus[max(0, rank * size - WINDOW):rank * size]
        else:
            shard = generate(args.cls, size, seed=seed0 + rank)
            prefix = b""
            if rank > 0:
                prefix = generate(args.cls, size, seed=seed0 + rank - 1)[-WINDOW:]  # tail of the previous shard = dictionary
        total_bytes = size * world
        last_rank = world - 1
    resident = prefix + shard
    ctx = Context(dev_index, lib)
    ctx.set_input(resident)  # H2D, outside the timed region
    instart, inend = len(prefix), len(resident)
    final = 1 if rank == last_rank else 0
    header = bytes([31, 139, 8, 0, 0, 0, 0, 0, 2, 3])

    def gather_blobs(blob):
        if cdist is not None:
            return cdist.gather(blob)                                     # the library's RCCL gather over xGMI
        return sharding.gather_bytes(blob, rank, world, device, dist)   # torch.distributed's gather (RCCL)

    def gather_crc(crc):
        """(crc32, length) of every rank's shard, on rank 0"""
        rec = crc.to_bytes(4, "little") + len(shard).to_bytes(8, "little")
        parts = cdist.gather(rec) if cdist is not None else sharding.gather_bytes(rec, rank, world, device, dist)
        if parts is None:
            return None
        return [(int.from_bytes(bytes(p[:4]), "little"), int.from_bytes(bytes(p[4:12]), "little")) for p in parts]

    timing_acc = {}
    seg_acc = {}

    def step():
        ta = time.perf_counter()
        shard_crc = ctx.checksum(api.CRC32, instart, inend)   # k_checksum over the resident shard
        if inend > instart or total_bytes == 0:
            blob = ctx.deflate_range(options, instart, inend, final, as_array=True)   # the library's buffer, no copy
        else:
            blob = b""      # (strong scaling with more ranks than master blocks: nothing to do here)
        tb = time.perf_counter()
        for k, v in api.last_timing(lib).items():
            timing_acc[k] = timing_acc.get(k, 0.0) + v
        for k, v in api.last_seg_stats(lib).items():
            seg_acc[k] = seg_acc.get(k, 0.0) + v
        blobs = gather_blobs(blob)
        crcs = gather_crc(shard_crc)
        tc = time.perf_counter()
        timing_acc["deflate_range_call"] = timing_acc.get("deflate_range_call", 0.0) + (tb - ta)
        timing_acc["gather"] = timing_acc.get("gather", 0.0) + (tc - tb)
        if rank != 0:
            return None
        crc = crcs[0][0]
        for c, n in crcs[1:]:
            crc = lib.zmx_checksum_combine(api.CRC32, crc, c, n)
        trailer = crc.to_bytes(4, "little") + (total_bytes & 0xffffffff).to_bytes(4, "little")
        # the gzip stream in one malloc'ed buffer, as a C caller of zmx_chunks_merge gets it
        out = ctx.merge([b for b in blobs if len(b)], header, trailer, as_array=True)
        timing_acc["merge"] = timing_acc.get("merge", 0.0) + (time.perf_counter() - tc)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]

    def entry_step(opts):
        """ZopfliCompress on the host buffer, as a C caller does it; returns (output pointer, size) — caller frees."""
        outp, outsize = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0)
        lib.ZopfliCompress(ctypes.byref(opts), api.FORMAT_GZIP, shard, len(shard), ctypes.byref(outp), ctypes.byref(outsize))
        for k, v in api.last_timing(lib).items():
            timing_acc[k] = timing_acc.get(k, 0.0) + v
        for k, v in api.last_seg_stats(lib).items():
            seg_acc[k] = seg_acc.get(k, 0.0) + v
        return outp, outsize.value

    def run_entry(opts, steps, warmup):
        """`steps` timed ZopfliCompress calls after `warmup` untimed ones; the last output as bytes."""
        for _ in range(warmup):
            o, n = entry_step(opts)
            libc.free(o)
        timing_acc.clear()
        seg_acc.clear()
        sync()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            if last is not None:
                libc.free(last[0])
            last = entry_step(opts)
        sync()
        dt_ = time.perf_counter() - t0
        data = ctypes.string_at(last[0], last[1])   # outside the timed region
        libc.free(last[0])
        return dt_, data

    def run_resident(steps, warmup):
        for _ in range(warmup):
            step()
        timing_acc.clear()
        seg_acc.clear()
        sync()
        t0 = time.perf_counter()
        o = None
        for _ in range(steps):
            o = step()
        sync()
        return time.perf_counter() - t0, o

    entry_dt = entry_out = None
    resident_timing = resident_seg = None
    if world == 1 and args.entry in ("both", "resident"):
        dt, out = run_resident(args.steps, args.warmup)
        resident_timing, resident_seg = dict(timing_acc), dict(seg_acc)
    if world == 1 and args.entry in ("both", "zopfli_compress"):
        # (the resident run's context gives its cached table arrays back first: what the contexts of a device keep
        #  cached counts against ONE budget per device, and the entry point's own contexts are about to want it)
        lib.zmx_ctx_trim_cache.argtypes = [ctypes.c_void_p]
        lib.zmx_ctx_trim_cache(ctx.handle)
        entry_dt, entry_out = run_entry(options, args.steps, args.warmup)   # (timing_acc / seg_acc now hold the entry point's)
        if args.entry == "zopfli_compress":
            dt, out = entry_dt, None
    if world > 1:
        dt, out = run_resident(args.steps, args.warmup)
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- N > 1: the same total input through ZopfliCompress in ONE process that holds all N devices
    #      (ZOPFLI_AMD_DEVICES = N: api.cc RunPartsSharded deals the master blocks over them and merges) — what a
    #      program that links libzopfli.so.1 gets on a multi-GPU node without any launcher.  Rank 0 runs it after the
    #      RCCL measurement while the other ranks wait; its stream must equal the gathered one.
    line = None
    in_process = None
    if world > 1 and not args.no_in_process:
        if rank == 0:
            try:
                os.environ["ZOPFLI_AMD_DEVICES"] = args.devices or str(world)    # (this process has not used the entry points yet)
                if strong:
                    whole_in = corpus[:size] if corpus is not None else generate(args.cls, size, seed=seed0)
                elif corpus is not None:
                    whole_in = corpus[:size * world]
                else:
                    whole_in = b"".join(generate(args.cls, size, seed=seed0 + r) for r in range(world))
                k = max(1, min(args.steps, 3))

                def whole_step():
                    outp, outsize = ctypes.POINTER(ctypes.c_ubyte)(), ctypes.c_size_t(0)
                    lib.ZopfliCompress(ctypes.byref(options), api.FORMAT_GZIP, whole_in, len(whole_in), ctypes.byref(outp), ctypes.byref(outsize))
                    return outp, outsize.value
                o, n = whole_step()
                libc.free(o)
                t0 = time.perf_counter()
                for _ in range(k):
                    o, n = whole_step()
                    if _ + 1 < k:
                        libc.free(o)
                dti = (time.perf_counter() - t0) / k
                got = ctypes.string_at(o, n)
                libc.free(o)
                in_process = {"value": round(len(whole_in) / MB / dti, 4), "unit": "MB/s", "ms_per_step": round(dti * 1e3, 2), "steps": k,
                              "devices": world, "same_stream_as_gathered": (got == out.tobytes()) if out is not None else None,
                              "entry": "ZopfliCompress in rank 0's process with ZOPFLI_AMD_DEVICES=%d (host buffer in, stream out; "
                                       "the other ranks idle)" % world}
            except Exception as e:    # (reported, not fatal: the RCCL line stands on its own)
                in_process = {"error": str(e)[:300]}
        dist.barrier()

    if rank == 0:
        total = total_bytes
        resident_out = out.tobytes() if out is not None else None   # outside the timed region
        value_resident = ms_resident = None
        if world > 1 or args.entry != "zopfli_compress":
            ms_resident = dt / args.steps * 1e3
            value_resident = total / MB / (dt / args.steps)
        if entry_dt is not None:
            # the headline: through the reference's entry point
            ms = entry_dt / args.steps * 1e3
            value = total / MB / (entry_dt / args.steps)
            if resident_out is not None and resident_out != entry_out:
                raise SystemExit("bench.py: ZopfliCompress and the resident path produced different streams")
            out = entry_out
        else:
            ms, value = ms_resident, value_resident
            out = resident_out
        # ---- correctness of the measured output (outside the timed region)
        bitexact = None
        roundtrip = None
        if world == 1 or strong:
            if world == 1:
                roundtrip = gzip.decompress(out) == shard
            else:
                dd = zlib.decompressobj(31)
                roundtrip = (len(dd.decompress(out)) + len(dd.flush())) == total
            sha = hashlib.sha256(out).hexdigest()
            for name in ("vectors_big.json", "vectors_big2.json", "vectors_big3.json", "vectors_big4.json", "vectors.json"):
                p = os.path.join(ROOT, "tests", "golden", name)
                if os.path.exists(p):
                    for c in json.load(open(p)):
                        if (corpus is None and c["input"].get("cls") == args.cls and c["input"].get("seed") in (None, seed0)
                                and c["insize"] == size and c["format"] == 0
                                and c["numiterations"] == args.numiterations
                                and c["blocksplitting"] == args.blocksplitting and c["blocksplittingmax"] == 15):
                            bitexact = (sha == c["sha256"])
        else:
            d = zlib.decompressobj(31)
            n = len(d.decompress(out)) + len(d.flush())
            roundtrip = (n == total)
        # ---- roofline of the dominant kernels: the chain (GetBestLengths: k_dp5_spec + k_dpcheck + k_dp4_fix of
        #      one squeeze run; algorithmic bytes = 31 B per position per run: 28 B match record + 1 B literal +
        #      2 B length_array) and, beside it, the match-table kernel (29 B per position)
        # (kernel durations and task statistics from the RESIDENT run where there is one: a single context, one stream — the
        #  entry point deals a large request over two contexts of the device, whose kernels interleave)
        k_timing = resident_timing if resident_timing is not None else timing_acc
        k_seg = resident_seg if resident_seg is not None else seg_acc
        launches = k_timing.get("squeeze_launches", 0.0)
        ksec = k_timing.get("dp_kernel", 0.0)
        copy_gbs = measured_copy_gbs(torch, device) if world == 1 else None
        roofline = None
        if launches > 0 and ksec > 0:
            per_launch_bytes = 31.0 * len(shard)   # (this rank's positions: the whole input at N = 1)
            achieved = per_launch_bytes / (ksec / launches) / 1e9
            # HBM bytes per launch from the rocprofv3 PMC passes of this exact workload (FETCH_SIZE x 2 +
            # WRITE_SIZE, MI355X_MICROARCH.md): only reported when the committed profile was taken on the
            # same configuration
            traffic = None
            traffic_note = "no PMC profile of this workload"
            PMC_PROFILE = PMC_PROFILES.get((args.cls, size, args.numiterations, args.blocksplitting), "")
            pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)
            if PMC_PROFILE and corpus is None and os.path.exists(pmc) and world == 1:
                with open(pmc) as f:
                    prof = json.load(f)
                if prof.get("device_source_sha16") == device_source_hash():
                    traffic = round(prof.get("chain", {}).get("hbm_bytes", 0) / 1e9, 3) or None
                    traffic_note = "rocprofv3 PMC passes of THIS build (device sources sha16 %s), profiles/%s" % (
                        prof.get("device_source_sha16"), PMC_PROFILE)
                else:
                    traffic_note = ("profiles/%s was taken on device sources sha16 %s, this build is %s: not quoted (re-run "
                                    "tools/collect_profiles.sh)" % (PMC_PROFILE, prof.get("device_source_sha16"), device_source_hash()))
            roofline = {"bound": "hbm", "kernel": "the chain of one squeeze run: k_dp5_spec + k_dpcheck + k_dp4_fix",
                        "achieved": round(achieved, 3), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                        "traffic_unit": "GB per launch", "traffic_source": traffic_note,
                        "algorithmic_gb_per_launch": round(per_launch_bytes / 1e9, 3),
                        "avg_launch_ms": round(ksec / launches * 1e3, 3), "launches_per_step": launches / args.steps,
                        "measured_copy_peak_gbs": copy_gbs}
            tasks = k_seg.get("tasks", 0.0)
            if tasks:
                # how the chain's tasks fared: what was accepted as computed, what was run again serially and why
                roofline["chain"] = {
                    "tasks_per_launch": round(tasks / launches, 1),
                    "accepted_frac": round(k_seg.get("accepted", 0.0) / tasks, 5),
                    "rerun_state_frac": round((k_seg.get("rerun_state", 0.0) + k_seg.get("rerun_values", 0.0)) / tasks, 5),
                    "rerun_level_frac": round(k_seg.get("rerun_level", 0.0) / tasks, 5),
                    "rerun_tie_frac": round(k_seg.get("rerun_tie", 0.0) / tasks, 5),
                    "positions_rerun_frac": round(k_seg.get("positions_rerun", 0.0) / max(k_seg.get("positions", 1.0), 1.0), 5)}
        msec, mpos = k_timing.get("match_kernel", 0.0), k_timing.get("positions_matched", 0.0)
        roofline_match = None
        if msec > 0 and mpos > 0:
            ach = 29.0 * mpos / msec / 1e9
            roofline_match = {"bound": "hbm", "kernel": "the match-table kernel(s): per block k_match5 (exact skip-walk) where k_hits estimates more than 300 hits per position, else k_match2 (ZOPFLI_AMD_MATCH forces one)", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s",
                              "frac": round(ach / 8000.0, 6), "seconds_per_step": round(msec / args.steps, 5),
                              "positions_per_step": mpos / args.steps,
                              "ns_per_position": round(msec / mpos * 1e9, 4),
                              "hash_kernels_seconds_per_step": round(k_timing.get("hash_kernels", 0.0) / args.steps, 5)}
            wpos = k_timing.get("skip_walk_positions", 0.0)
            if wpos > 0:
                # k_match5's own counts: what the skip-walk touched instead of the reference's hit-by-hit visits
                li, wi = k_timing.get("skip_walk_lane_iterations", 0.0), k_timing.get("skip_walk_wave_iterations", 0.0)
                roofline_match["skip_walk"] = {
                    "positions_per_step": wpos / args.steps,
                    "share_of_positions": round(wpos / mpos, 4),
                    "entries_in_flight_per_position": round(li / wpos, 2),
                    "lanes_busy_per_wave_iteration": round(li / max(wi, 1.0), 1),
                    "note": "entries in flight = lanes with an entry's loads outstanding, summed over the wave iterations (an "
                            "entry longer than 15 bytes in common takes one more iteration per 16 bytes); the reference's walk "
                            "visits 100 (text) to 2 200 (two-symbol data) chain entries per position, counted here from ranks"}
        line = {
            "metric": "input MB/s at numiterations=15 (gzip, bit-exact vs reference)",
            "value": round(value, 4), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "entry": ("ZopfliCompress(options, ZOPFLI_FORMAT_GZIP, host buffer, size, &out, &outsize) of libzopfli_amd.so: host "
                      "to device copy and context reuse inside the clock" if entry_dt is not None else
                      "resident input: zmx_deflate_range + zmx_chunks_merge (H2D outside the clock)"),
            "value_resident": None if value_resident is None else round(value_resident, 4),
            "ms_per_step_resident": None if ms_resident is None else round(ms_resident, 2),
            "vs_baseline": None,
            "dtype": "u8 (f32/f64 cost DP)", "data": "synthetic" if corpus is None else "file:" + os.path.basename(args.file),
            "config": {"workload": (f"class-{args.cls} synthetic ({CLASS_NAMES.get(args.cls, 'class ' + args.cls)})" if corpus is None
                                    else "file " + os.path.basename(args.file)) +
                                   f" {size} B {'in total' if strong else 'per GPU'}, numiterations="
                                   f"{args.numiterations}, blocksplitting={args.blocksplitting}, gzip, "
                                   + baseline_config_name(args.cls if corpus is None else None, size, args.numiterations,
                                                          args.blocksplitting),
                       "total_bytes": total, "master_blocks": (total + MB - 1) // MB,
                       "sharding": "master blocks, contiguous per rank (strong scaling: ranges balanced by the estimated cost of their bytes, zmx_master_block_costs; weak: a fixed size per rank), gather of bit chunks to rank 0",
                       "gather": gather_kind},
            "output_bytes": len(out), "roundtrip_ok": roundtrip, "bitexact_vs_reference": bitexact,
            "roofline": roofline,
            "roofline_match": roofline_match,
            "chain_tasks_per_step": {k: round(v / args.steps, 1) for k, v in seg_acc.items()},
            "breakdown_s_per_step": {k: round(v / args.steps, 4) for k, v in timing_acc.items()
                                     if k not in ("squeeze_launches", "table_builds", "positions_matched")},
            "breakdown_note": ("of the run `value` comes from.  Through ZopfliCompress a call of 32 master blocks or more is "
                               "dealt over three contexts of the device: dp_kernel / match_kernel / hash_kernels / "
                               "trace_kernel / wtab_kernel are then SUMS over those contexts of kernels that interleave on "
                               "the device (each stretched by the others); the host phases are the slowest context's.  "
                               "`breakdown_s_per_step_resident` is the same work on ONE context, one stream: the kernel "
                               "durations the roofline objects are computed from."),
            "breakdown_s_per_step_resident": None if resident_timing is None else {
                k: round(v / args.steps, 4) for k, v in resident_timing.items()
                if k not in ("squeeze_launches", "table_builds", "positions_matched")},
        }
        if roofline is not None and entry_dt is not None and timing_acc.get("squeeze_launches"):
            roofline["avg_launch_ms_entry_summed_over_contexts"] = round(
                timing_acc.get("dp_kernel", 0.0) / timing_acc["squeeze_launches"] * 1e3, 3)
        tasks_per_launch = k_seg.get("tasks", 0.0) / launches if launches else 0.0
        if world == 1 and tasks_per_launch and ksec > 0:
            # Predicted strong-scaling ceiling from the chain's latency: a squeeze run cannot take less than one
            # round of task waves (a task is a serial chain of ~4600 positions) however few blocks a GPU holds;
            # the merge on rank 0 and the gather do not shrink either.  T(N) = c + (T1 - c) / N.
            rounds = max(1.0, tasks_per_launch / (256 * 16))
            c = (ksec / rounds + k_timing.get("merge", 0.0) + k_timing.get("gather", 0.0)) / args.steps
            t1 = ms / 1e3
            line["strong_scaling_model"] = {
                "formula": "T(N) = c + (T1 - c)/N, c = runs x (chain time / rounds of task waves) + merge + gather",
                "wave_rounds_per_run": round(rounds, 2), "c_ms": round(c * 1e3, 2),
                "predicted_speedup": {str(n): round(t1 / (c + (t1 - c) / n), 2) for n in (2, 4, 8)}}
        if world == 1 and args.blocksplitting == 0 and not args.no_blocksplitting1 and entry_dt is not None:
            # the reference's default options (util.c:28-35): block splitting on — BASELINE configs[2] on one GPU
            opt1 = ZopfliOptions(args.numiterations, 1, 15)
            dt1, out1 = run_entry(opt1, max(1, min(args.steps, 2)), 1)
            n1 = max(1, min(args.steps, 2))
            bit1 = None
            sha1 = hashlib.sha256(out1).hexdigest()
            for name in ("vectors_big.json", "vectors_big2.json", "vectors_big3.json", "vectors_big4.json", "vectors.json"):
                pth = os.path.join(ROOT, "tests", "golden", name)
                if os.path.exists(pth):
                    for c in json.load(open(pth)):
                        if (corpus is None and c["input"].get("cls") == args.cls and c["input"].get("seed") in (None, seed0)
                                and c["insize"] == size and c["format"] == 0 and c["numiterations"] == args.numiterations
                                and c["blocksplitting"] == 1 and c["blocksplittingmax"] == 15):
                            bit1 = (sha1 == c["sha256"])
            line["blocksplitting1"] = {
                "value": round(total / MB / (dt1 / n1), 4), "unit": "MB/s", "ms_per_step": round(dt1 / n1 * 1e3, 2), "steps": n1,
                "output_bytes": len(out1), "roundtrip_ok": gzip.decompress(out1) == shard, "bitexact_vs_reference": bit1,
                "breakdown_s_per_step": {k: round(v / n1, 4) for k, v in timing_acc.items()
                                         if k in ("tables", "greedy", "squeeze", "cost_model", "split", "encode", "dp_kernel", "match_kernel")},
                "config": "the same input through ZopfliCompress with the reference's default options: blocksplitting=1, "
                          "blocksplittingmax=15 (configs[2] on one GPU)"}
        if in_process is not None:
            line["in_process"] = in_process
        if world == 1 and not args.no_small_files and args.entry != "resident" and corpus is None:
            line["small_files"] = small_files_line(lib, api, generate, ZopfliOptions, args.numiterations, [(200, 65536), (48, 1000000)],
                                                   with_reference=not args.no_cpu_baseline)
        if not args.no_cpu_baseline:
            # (rank 0's host cores; at N > 1 the other ranks wait at the next collective meanwhile)
            sample = shard[:min(args.cpu_sample, size)]
            res = cpu_baseline(sample, options)
            if res:
                line["cpu_baseline"] = res[0]
            allc = cpu_baseline_all_cores(shard, options) if world == 1 else None
            if allc:
                line["cpu_baseline_all_cores"] = allc
    ctx.close()
    return line if rank == 0 else None


if __name__ == "__main__":
    main()
