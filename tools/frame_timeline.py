#!/usr/bin/env python
"""One frame of a rocprofv3 --kernel-trace database as a launch sequence: kernel, duration, gap to the previous kernel.
usage: frame_timeline.py <results.db> [frame index from the end, default 3]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
pre = [i for i, r in enumerate(rows) if "preprocess_fwd_kernel" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = pre[-k - 1], pre[-k]
prev_end = rows[a][1]
tot_k = tot_g = 0.0
for n, s, e in rows[a:b]:
    gap = (s - prev_end) / 1e3
    dur = (e - s) / 1e3
    tot_k += dur
    tot_g += max(gap, 0.0)
    print(f"{n.split('(')[0][-52:]:52s} {dur:9.2f} us   gap {gap:7.2f}")
    prev_end = e
print(f"kernels {tot_k:.1f} us + gaps {tot_g:.1f} us = {(rows[b - 1][2] - rows[a][1]) / 1e3:.1f} us from the first launch to the last end")
