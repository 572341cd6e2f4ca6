#!/usr/bin/env python
"""GPU busy / idle analysis of a rocprofv3 --kernel-trace rocpd database: over the steady-state window (the
middle half of the trace) how much of the time at least one kernel runs, how many run at once, which kernels the
time goes to.  usage: timeline_busy.py <results.db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = cur.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall() \
    if "queue_id" in cols and "stream_id" in cols else \
    [r + (0, 0) for r in cur.execute("select name, start, end from kernels order by start").fetchall()]
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo, hi = t0 + (t1 - t0) * 0.45, t0 + (t1 - t0) * 0.85
win = [(n, max(s, lo), min(e, hi), q, st) for n, s, e, q, st in rows if e > lo and s < hi]
ev = []
for n, s, e, q, st in win:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
busy = 0.0
conc_time = {}
cur_n, last = 0, lo
for t, d in ev:
    if cur_n > 0:
        busy += t - last
    conc_time[cur_n] = conc_time.get(cur_n, 0) + (t - last)
    cur_n += d
    last = t
conc_time[cur_n] = conc_time.get(cur_n, 0) + (hi - last)
span = hi - lo
print(f"window {span / 1e6:.2f} ms, kernels {len(win)}, busy (>=1 kernel) {100 * busy / span:.1f} %")
print("time with n kernels running:", {k: f"{100 * v / span:.1f}%" for k, v in sorted(conc_time.items())})
agg = {}
for n, s, e, q, st in win:
    key = n.split("(")[0][-60:]
    agg[key] = agg.get(key, 0) + (e - s)
tot = sum(agg.values())
print(f"sum of kernel durations / window = {tot / span:.2f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:60s} {100 * v / span:6.1f} % of the window")
print("queues:", sorted({q for _, _, _, q, _ in win}), "streams:", sorted({st for _, _, _, _, st in win}))
