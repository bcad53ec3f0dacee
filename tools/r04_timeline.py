"""round 4: from a rocprofv3 kernel trace, the device-busy union and the idle gaps of the last ZopfliCompress call
(the calls are separated by the longest gaps)."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], r.get("Queue_Id", "?")))
rows.sort()
# calls: split at gaps > 20 ms
calls, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] - end > 20e6:
        calls.append(cur)
        cur = []
    cur.append(r)
    end = max(end or 0, r[1])
calls.append(cur)
print("calls (kernels each):", [len(c) for c in calls])
for call in calls[-2:]:
    t0, t1 = call[0][0], max(r[1] for r in call)
    busy, gaps, e = 0, [], call[0][0]
    s = call[0][0]
    last = call[0]
    for r in call:
        if r[0] > e:
            busy += e - s
            gaps.append((r[0] - e, e - t0, last[2], r[2]))
            s = r[0]
        if r[1] > e:
            e = r[1]
            last = r
    busy += e - s
    print("call: span %.1f ms, busy union %.1f ms, idle %.1f ms in %d gaps; sum of kernel durations %.1f ms" %
          ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), sum(r[1] - r[0] for r in call) / 1e6))
    big = sorted(gaps, reverse=True)[:12]
    for g in sorted(big, key=lambda x: x[1]):
        print("   gap %.2f ms at +%.1f ms: after %s, before %s" % (g[0] / 1e6, g[1] / 1e6, g[2], g[3]))
    # concurrency histogram: time with n kernels in flight
    ev = []
    for r in call:
        ev.append((r[0], 1))
        ev.append((r[1], -1))
    ev.sort()
    depth, prev, hist = 0, ev[0][0], defaultdict(int)
    for t, d in ev:
        hist[min(depth, 6)] += t - prev
        prev, depth = t, depth + d
    print("   ms with n kernels in flight:", {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
    per = defaultdict(int)
    for r in call:
        per[r[2]] += r[1] - r[0]
    print("   top kernels (ms summed):", [(k, round(v / 1e6, 1)) for k, v in sorted(per.items(), key=lambda x: -x[1])[:10]])
