#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a small CSV for profiles/.
usage: rocprof_summary.py <results.db> <out.csv> [note]"""
import sqlite3, sys, csv
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    if len(sys.argv) > 3:
        f.write("# " + sys.argv[3] + "\n")
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name.split("(")[0], calls, "%.1f" % tot, "%.2f" % avg, "%.3f" % pct])
