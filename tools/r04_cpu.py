"""round 4: host CPU seconds and cgroup throttling per ZopfliCompress call (100 MB class T by default).
usage: python tools/r04_cpu.py [cls] [size] [blocksplitting] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zopfli_amd import ZopfliOptions, api, generate  # noqa: E402


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


def main():
    cls = sys.argv[1] if len(sys.argv) > 1 else "T"
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    lib = api.library()
    data = generate(cls, size)
    opt = ZopfliOptions(15)
    opt.blocksplitting = bs
    api.compress(data, 0, opt, lib=lib)
    rows = []
    for _ in range(steps):
        s0, c0, t0 = cpu_stat(), time.process_time(), time.perf_counter()
        out = api.compress(data, 0, opt, lib=lib)
        t1, c1, s1 = time.perf_counter(), time.process_time(), cpu_stat()
        rows.append({"wall_ms": round((t1 - t0) * 1e3, 1), "cpu_ms": round((c1 - c0) * 1e3, 1),
                     "throttled_ms": round((s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3, 1),
                     "nr_throttled": s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0),
                     "cgroup_cpu_ms": round((s1.get("usage_usec", 0) - s0.get("usage_usec", 0)) / 1e3, 1)})
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        quota = None
    print(json.dumps({"cls": cls, "size": size, "blocksplitting": bs, "out": len(out), "cpus": os.cpu_count(),
                      "affinity": len(os.sched_getaffinity(0)), "cpu.max": quota,
                      "env": {k: v for k, v in os.environ.items() if k.startswith("ZOPFLI_AMD")}, "steps": rows}))


if __name__ == "__main__":
    main()
