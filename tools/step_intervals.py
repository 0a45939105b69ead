#!/usr/bin/env python3
"""Step-to-step intervals of a run from a rocprofv3 kernel trace (rocpd SQLite database): the time from each momentum-tendency
launch to the next over the last N steps, and what still runs after the last one -- how much of a short timed loop is start-up
and wind-down rather than steady state.
usage: step_intervals.py <directory holding */*.db> [N = 31]"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/*/*.db")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 31
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
mom = [(r[1], r[2]) for r in rows if "k_momentum" in r[0]]
print("momentum launches:", len(mom))
trc = [(r[1], r[2]) for r in rows if "k_tracer_tendencies" in r[0]]
n = min(n, len(mom) - 1)
for q in range(len(mom) - n, len(mom)):
    a, b = mom[q - 1], mom[q]
    t = [x for x in trc if a[0] <= x[0] < b[0]]
    print(f"{(b[0] - a[0]) / 1e3:8.1f} us   (momentum kernel {(a[1] - a[0]) / 1e3:.1f}" + (f", tracer kernel {(t[0][1] - t[0][0]) / 1e3:.1f})" if t else ")"))
print("after the start of the last momentum launch:")
for r in rows:
    if r[1] >= mom[-1][0]:
        print(f"  {r[0].replace('void gb25::', '')[:80]} stream {r[3]} start {(r[1] - mom[-1][0]) / 1e3:.1f} duration {(r[2] - r[1]) / 1e3:.1f}")
