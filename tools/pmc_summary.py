"""DEVELOPER-ONLY: average PMC counter values per kernel from rocprofv3 --pmc runs (rocpd *_results.db files under a dir)."""
import collections
import glob
import sqlite3
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    db = sqlite3.connect(fn)
    for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        name = name.split("(")[0].replace("void ", "").replace("pg::", "")
        acc[name][cname].append(float(val))
want = sys.argv[2:] or None
for k, cs in acc.items():
    if want and not any(w in k for w in want):
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
