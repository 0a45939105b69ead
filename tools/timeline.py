#!/usr/bin/env python3
"""Stream timeline of one steady-state time step from a rocprofv3 kernel trace (rocpd SQLite database):
every kernel between two consecutive momentum-tendency launches, with its stream, block count, start, duration and end.
usage: timeline.py <results.db> [which step from the end, default 5] [delimiting kernel, default k_momentum; a slab launches
that one twice per step: use k_tracer_tendencies]"""
import sqlite3
import sys

db = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x*grid_y*grid_z/(workgroup_x*workgroup_y*workgroup_z) "
                   "from kernels order by start").fetchall()
key = sys.argv[3] if len(sys.argv) > 3 else "k_momentum"
idx = [i for i, r in enumerate(rows) if key in r[0]]
i0, i1 = idx[-back - 1], idx[-back]
t0 = rows[i0][1]
print("| kernel | stream | blocks | start (us) | duration (us) | end (us) |")
print("|---|---|---|---|---|---|")
for r in rows[i0:i1 + 1]:
    n = r[0].replace("void gb25::", "").replace("gb25::", "").split("(")[0]
    print(f"| `{n}` | {r[3]} | {r[4]} | {(r[1] - t0) / 1e3:.1f} | {(r[2] - r[1]) / 1e3:.1f} | {(r[2] - t0) / 1e3:.1f} |")
print(f"\nstep = {(rows[i1][1] - t0) / 1e3:.1f} us from {key} launch to {key} launch")
