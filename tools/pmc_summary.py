"""Per-kernel, per-launch averages of a rocprofv3 --pmc run (counter_collection.csv) as JSON.
usage: python tools/pmc_summary.py <p_counter_collection.csv> [...more passes] > profiles/<name>.json
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B as reported by rocprofv3; FETCH_SIZE is doubled
on output (`fetch_bytes`) as MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads."""
import collections
import csv
import json
import re
import sys


def main(paths):
    out = collections.defaultdict(dict)
    for path in paths:
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(set)
        with open(path) as f:
            for r in csv.DictReader(f):
                k = re.sub(r"<.*>$", "", re.sub(r"^void\s+", "", r["Kernel_Name"].split("(")[0]))   # k_dp<false> -> k_dp
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k].add(r["Dispatch_Id"])
        for k, v in agg.items():
            n = max(1, len(launches[k]))
            out[k]["launches"] = n
            for c, x in v.items():
                out[k][c] = x / n
    # the chain of one squeeze run = k_dp5_spec (both passes) + k_dpcheck + k_dpscan + k_dp4_fix: totals over
    # all their launches divided by the number of squeeze runs (= launches of k_dp4_fix)
    runs = out.get("k_dp4_fix", {}).get("launches", 0)
    if runs:
        chain = collections.defaultdict(float)
        for k in ("k_dp5_spec", "k_dpcheck", "k_dpscan", "k_dp4_fix"):
            for c, x in out.get(k, {}).items():
                if c != "launches":
                    chain[c] += x * out[k]["launches"] / runs
        chain["launches"] = runs
        out["chain"] = dict(chain)
    for k, v in out.items():
        if "FETCH_SIZE" in v:
            v["fetch_bytes"] = v["FETCH_SIZE"] * 1024 * 2
        if "WRITE_SIZE" in v:
            v["write_bytes"] = v["WRITE_SIZE"] * 1024
        if "fetch_bytes" in v and "write_bytes" in v:
            v["hbm_bytes"] = v["fetch_bytes"] + v["write_bytes"]
    # which build this was taken on (bench.py quotes `traffic` only for the same device sources)
    import hashlib
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zopfli_amd", "csrc", "device")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    out["device_source_sha16"] = h.hexdigest()[:16]
    json.dump(out, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:])
