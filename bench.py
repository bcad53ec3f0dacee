"""bench.py — zopfli hot path on MI355X.

One step = one pass of the hot path over the workload: ZopfliCompress(gzip) semantics on an
input that is already resident in HBM (H2D is outside the timed region), i.e. match tables,
greedy seeds, numiterations squeeze runs, host cost model, block choice, encoding, merge.

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
import threading
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MB = 1000000
WINDOW = 32768
PMC_PROFILE = "r01_v8_bench100MB_pmc.json"
SHADER_CLOCK_HZ = 2.4e9   # MI355X peak engine clock; s_memtime counts at this rate (measured)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=100 * MB, help="bytes per GPU (default: the 100 MB workload)")
    ap.add_argument("--blocksplitting", type=int, default=0, help="0 = configs[1] (default), 1 = configs[2]")
    ap.add_argument("--cls", default="T", choices=list("TXRZBPM"),
                    help="synthetic input class (zopfli_amd/csrc/tools/datagen.c): T text-like (default, enwik8 "
                         "stand-in), X markup-like, M mixed corpus (Silesia stand-in), ...")
    ap.add_argument("--numiterations", type=int, default=15)
    ap.add_argument("--cpu-sample", type=int, default=8 * MB)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo "
                    "exercises the same sharding/gather/merge code where only one GPU is visible)")
    ap.add_argument("--device-index", type=int, default=None, help="HIP device of this rank (default LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import torch
    import torch.distributed as dist

    from zopfli_amd import Context, ZopfliOptions, api, generate, sharding
    from zopfli_amd.sharding import crc32_combine

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
    options = ZopfliOptions(args.numiterations, args.blocksplitting, 15)
    size = args.size
    assert size % MB == 0 or world == 1, "shards must be whole master blocks"

    # ---- synthetic input: shard r = class `cls`, seed = the class's default seed + r
    from zopfli_amd.datagen import DEFAULT_SEED
    seed0 = DEFAULT_SEED[args.cls]
    shard = generate(args.cls, size, seed=seed0 + rank)
    prefix = b""
    if rank > 0:
        prefix = generate(args.cls, size, seed=seed0 + rank - 1)[-WINDOW:]  # tail of the previous shard = dictionary
    resident = prefix + shard
    ctx = Context(dev_index, lib)
    ctx.set_input(resident)  # H2D, outside the timed region
    instart, inend = len(prefix), len(resident)
    final = 1 if rank == world - 1 else 0
    header = bytes([31, 139, 8, 0, 0, 0, 0, 0, 2, 3])

    def gather_blobs(blob):
        return sharding.gather_bytes(blob, rank, world, device, dist)   # one RCCL gather over xGMI

    def gather_crc(crc):
        parts = sharding.gather_bytes(crc.to_bytes(4, "little"), rank, world, device, dist)
        return None if parts is None else [int.from_bytes(bytes(p), "little") for p in parts]

    timing_acc = {}
    seg_acc = {}

    def step():
        crc_box = [0]
        th = threading.Thread(target=lambda: crc_box.__setitem__(0, zlib.crc32(shard)))
        th.start()  # the checksum does not depend on the device work
        ta = time.perf_counter()
        blob = ctx.deflate_range(options, instart, inend, final, as_array=True)   # the library's buffer, no copy
        tb = time.perf_counter()
        for k, v in api.last_timing(lib).items():
            timing_acc[k] = timing_acc.get(k, 0.0) + v
        for k, v in api.last_seg_stats(lib).items():
            seg_acc[k] = seg_acc.get(k, 0.0) + v
        th.join()
        blobs = gather_blobs(blob)
        crcs = gather_crc(crc_box[0])
        tc = time.perf_counter()
        timing_acc["deflate_range_call"] = timing_acc.get("deflate_range_call", 0.0) + (tb - ta)
        timing_acc["gather"] = timing_acc.get("gather", 0.0) + (tc - tb)
        if rank != 0:
            return None
        crc = crcs[0]
        for c in crcs[1:]:
            crc = crc32_combine(crc, c, size)
        total = size * world
        trailer = crc.to_bytes(4, "little") + (total & 0xffffffff).to_bytes(4, "little")
        # the gzip stream in one malloc'ed buffer, as a C caller of zmx_chunks_merge gets it
        out = ctx.merge(blobs, header, trailer, as_array=True)
        timing_acc["merge"] = timing_acc.get("merge", 0.0) + (time.perf_counter() - tc)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    timing_acc.clear()
    seg_acc.clear()
    sync()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        out = out.tobytes()   # outside the timed region
        total = size * world
        ms = dt / args.steps * 1e3
        value = total / MB / (dt / args.steps)
        # ---- correctness of the measured output (outside the timed region)
        bitexact = None
        roundtrip = None
        if world == 1:
            roundtrip = gzip.decompress(out) == shard
            sha = hashlib.sha256(out).hexdigest()
            for name in ("vectors_big.json", "vectors_big2.json", "vectors.json"):
                p = os.path.join(ROOT, "tests", "golden", name)
                if os.path.exists(p):
                    for c in json.load(open(p)):
                        if (c["input"].get("cls") == args.cls and c["input"].get("seed") in (None, seed0)
                                and c["insize"] == size and c["format"] == 0
                                and c["numiterations"] == args.numiterations
                                and c["blocksplitting"] == args.blocksplitting and c["blocksplittingmax"] == 15):
                            bitexact = (sha == c["sha256"])
        else:
            d = zlib.decompressobj(31)
            n = len(d.decompress(out)) + len(d.flush())
            roundtrip = (n == total)
        # ---- roofline of the dominant kernel (k_dp3, the serial DP chain): algorithmic bytes = 31 B per
        #      position per launch (28 B match record + 1 B literal + 2 B length_array)
        launches = timing_acc.get("squeeze_launches", 0.0)
        ksec = timing_acc.get("dp_kernel", 0.0)
        roofline = None
        if launches > 0 and ksec > 0:
            per_launch_bytes = 31.0 * size
            achieved = per_launch_bytes / (ksec / launches) / 1e9
            # HBM bytes per k_dp launch from the rocprofv3 PMC passes of this exact workload
            # (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md; tools/collect_profiles.sh): only
            # reported when the committed profile was taken on the same configuration
            traffic = None
            pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)
            if (os.path.exists(pmc) and size == 100 * MB and args.numiterations == 15
                    and args.blocksplitting == 0 and world == 1):
                with open(pmc) as f:
                    traffic = round(json.load(f).get("k_dp3", {}).get("hbm_bytes", 0) / 1e9, 3) or None
            roofline = {"bound": "hbm", "kernel": "k_dp3", "achieved": round(achieved, 3), "peak": 8000.0,
                        "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                        "traffic_unit": "GB per launch (rocprofv3 PMC, profiles/" + PMC_PROFILE + ")",
                        "algorithmic_gb_per_launch": round(per_launch_bytes / 1e9, 3),
                        "avg_launch_ms": round(ksec / launches * 1e3, 3), "launches_per_step": launches / args.steps}
            # what really bounds k_dp3: one wave per master block walks a chain of dependent updates,
            # 8 VALU instructions per position at the 5.8 cycles a lone wave needs per instruction
            # (tools/ubench_chain.hip)
            block = min(size, MB)
            cyc = (ksec / launches) * SHADER_CLOCK_HZ / block
            roofline["chain"] = {"cycles_per_position": round(cyc, 1), "issue_floor_cycles_per_position": 46.4,
                                 "frac_of_issue_floor": round(46.4 / cyc, 3), "clock_ghz": SHADER_CLOCK_HZ / 1e9}
        line = {
            "metric": "input MB/s at numiterations=15 (gzip, bit-exact vs reference)",
            "value": round(value, 4), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8 (f32/f64 cost DP)", "data": "synthetic",
            "config": {"workload": f"class-{args.cls} synthetic {size} B per GPU (T = text-like enwik8 stand-in), numiterations="
                                   f"{args.numiterations}, blocksplitting={args.blocksplitting}, gzip, "
                                   f"{'configs[1]' if args.blocksplitting == 0 else 'configs[2]'}",
                       "total_bytes": total, "master_blocks": (total + MB - 1) // MB,
                       "sharding": "master blocks, contiguous per rank, RCCL gather of bit chunks"},
            "output_bytes": len(out), "roundtrip_ok": roundtrip, "bitexact_vs_reference": bitexact,
            "roofline": roofline,
            "chain_tasks_per_step": {k: round(v / args.steps, 1) for k, v in seg_acc.items()},
            "breakdown_s_per_step": {k: round(v / args.steps, 4) for k, v in timing_acc.items()
                                     if k != "squeeze_launches"},
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = shard[:min(args.cpu_sample, size)]
            res = cpu_baseline(sample, options)
            if res:
                line["cpu_baseline"] = res[0]
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
