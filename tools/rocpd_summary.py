#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel time stats and PMC counter means.
usage: rocpd_summary.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cur = con.cursor()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    if "name" in cols and "duration" in cols:
        rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                           "from kernels group by name order by sum(duration) desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'%':>6s}")
        for n, c, s, a, mn, mx in rows:
            if filt and filt not in n:
                continue
            print(f"{n[:70]:70s} {c:6d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / tot:6.2f}")
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if ccols:
        try:
            rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
                               "group by kernel_name, counter_name order by kernel_name").fetchall()
        except sqlite3.OperationalError:
            print("counters_collection columns:", ccols)
            rows = []
        last = None
        for k, cn, c, a, s in rows:
            if filt and filt not in k:
                continue
            if k != last:
                print(f"\n== {k[:100]}  (dispatches {c})")
                last = k
            print(f"   {cn:28s} avg/dispatch {a:18.1f}")


if __name__ == "__main__":
    main()
