#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a small CSV for profiles/.
usage: rocprof_summary.py <results.db> <out.csv> [note] [n_timed_k_merge]
n_timed_k_merge: the LAST n launches of k_merge are the ones inside bench.py's timed region (the earlier ones belong to the
warm-up on a scratch index); an extra row gives their total / average so that it can be compared with the hipEvent figure
bench.py reports (roofline.avg_launch_ms)."""
import sqlite3, sys, csv
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
n_timed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    if len(sys.argv) > 3:
        f.write("# " + sys.argv[3] + "\n")
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name.split("(")[0], calls, "%.1f" % tot, "%.2f" % avg, "%.3f" % pct])
    if n_timed:
        d = [(e - s) / 1e3 for s, e in db.execute("select start, end from kernels where name like '%k_merge(%' or name like '%k_merge<%' or name like '%k_merge' order by start")]
        d = d[-n_timed:]
        if d:
            w.writerow(["rb2::k_merge [the %d launches of the timed region only]" % len(d), len(d), "%.1f" % sum(d), "%.2f" % (sum(d) / len(d)), ""])
