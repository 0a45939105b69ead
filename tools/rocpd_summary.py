#!/usr/bin/env python3
"""Summarise a rocprofv3 result database (rocpd SQLite, the default output of ROCm 7.2) into small CSV files that can
be committed under profiles/.

  rocpd_summary.py stats    <results.db> <out.csv>   per-kernel calls / total / average / min / max duration
  rocpd_summary.py counters <results.db> <out.csv>   per-kernel, per-counter mean value over the launches
"""
import csv
import sqlite3
import sys


def short(name):
    return name.replace("void ", "").split("(")[0].replace("gb25::", "")


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, c, t, a, lo, hi in rows:
            w.writerow([short(n), c, int(t), round(a, 1), round(100.0 * t / total, 3), int(lo), int(hi)])


def counters(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration), max(vgpr_count), "
                       "max(lds_block_size), max(grid_size), max(workgroup_size) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Counter", "Launches", "MeanValue", "MeanDurationNs", "VGPRs", "LDS", "GridSize",
                    "WorkgroupSize"])
        for r in rows:
            w.writerow([short(r[0]), r[1], r[2], r[3], round(r[4], 1), r[5], r[6], r[7], r[8]])


if __name__ == "__main__":
    {"stats": stats, "counters": counters}[sys.argv[1]](sys.argv[2], sys.argv[3])
