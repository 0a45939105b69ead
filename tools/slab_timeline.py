#!/usr/bin/env python3
"""Kernels of ONE slab-step of the P-slab emulation from a rocprofv3 kernel trace of tools/slab_profile.py: the launches
between two consecutive tracer-tendency launches of the same slab (every P-th), with stream and duration.
usage: slab_timeline.py <results.db> P"""
import sqlite3, sys, collections
db, P = sys.argv[1], int(sys.argv[2])
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
short = lambda n: n.replace("void gb25::", "").replace("gb25::", "").split("(")[0]
# one whole step of all slabs: between two momentum-edge launches far apart: take the last 1/3 of the trace
n = len(rows)
tail = rows[2 * n // 3:]
span = (tail[-1][2] - tail[0][1]) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tail:
    a = agg[short(r[0])]
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
steps = agg[[k for k in agg if k.startswith("k_tracer_tendencies")][0]][0] / P
print(f"{len(tail)} launches over {span:.0f} us = {steps:.1f} steps of {P} slabs; per slab-step:")
print("| kernel | launches | us each | us per slab-step |")
print("|---|---|---|---|")
tot = 0
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    per = t / (steps * P)
    tot += per
    print(f"| `{k}` | {c / (steps * P):.1f} | {t / c:.1f} | {per:.1f} |")
print(f"\nsum of kernel durations per slab-step {tot:.0f} us; wall per slab-step {span / (steps * P):.0f} us; launches per slab-step {len(tail) / (steps * P):.1f}")
