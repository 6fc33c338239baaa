"""DEVELOPER-ONLY: kernel start/end times (us, relative) of the last steps of a rocprofv3 --kernel-trace run, to see the gaps
between launches.  Usage: trace_timeline.py <dir-with-_results.db> [n_last]"""
import glob
import sqlite3
import sys

n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for fn in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(fn)
    rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    rows = rows[-n_last:]
    t0 = rows[0][1]
    prev_end = t0
    for name, s, e, st in rows:
        short = name.split("(")[0].replace("void ", "").replace("pg::", "")[:40]
        print("%-40s stream %-4s start %9.1f  dur %8.1f  gap-after-prev-end %7.1f" % (short, st, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = max(prev_end, e)
