#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in ns) from a rocprofv3 --kernel-trace run stored as a rocpd database:
the same table `--stats` prints, written as CSV for profiles/.  Usage: kernel_stats.py <dir-with-_results.db> <out.csv>"""
import csv
import glob
import sqlite3
import sys

rows = []
for fn in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(fn)
    rows += db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
total = float(sum(r[2] for r in rows)) or 1.0
rows.sort(key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.2f" % (100.0 * r[2] / total), r[4], r[5]])
print(open(sys.argv[2]).read())
