#!/usr/bin/env python3
"""Per-launch durations of one kernel from a rocprofv3 kernel_trace CSV, in launch order.
usage: ktrace_rounds.py <kernel_trace.csv> <kernel substring> [every]"""
import csv, sys
path, name = sys.argv[1], sys.argv[2]
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path)) if name in r["Kernel_Name"]]
rows.sort()
d = [(e - s) / 1e3 for s, e in rows]
print("launches", len(d), "total_ms %.1f" % (sum(d) / 1e3))
print(" ".join("%.0f" % x for x in d[::every]))
