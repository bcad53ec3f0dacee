"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total/avg/min/max ms.
usage: python tools/rocpd_stats.py <results.db> [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
         f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc")
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'pct':>6s}")
    for name, n, tot, avg, mn, mx in rows:
        print(f"{name[:70]:70s} {n:6d} {tot/1e6:12.3f} {avg/1e6:10.4f} {mn/1e6:10.4f} {mx/1e6:10.4f} {100*tot/total:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
