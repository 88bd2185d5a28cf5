#!/usr/bin/env python3
"""Per-round timeline of the in-place rounds from a rocprofv3 --kernel-trace CSV: for every kernel name the average duration, and the
average gap between the end of one kernel and the start of the next on the same queue, over the steady-state rounds.
    python tools/timeline_rounds.py <kernel_trace.csv> [first_round last_round]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void rb2::", "").replace("rb2::", ""), r.get("Queue_Id", "0")))
rows.sort()
# rounds = from one k_merge_leaf to the next
idx = [i for i, r in enumerate(rows) if r[2].startswith("k_merge_leaf")]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (len(idx) // 2, len(idx) // 2 + 2000)
hi = min(hi, len(idx) - 1)
dur = collections.defaultdict(list); start_off = collections.defaultdict(list); per = []
for k in range(lo, hi):
    a, b = idx[k], idx[k + 1]
    t0 = rows[a][0]
    per.append(rows[b][0] - t0)
    for r in rows[a:b]:
        dur[r[2] + " q" + r[3]].append(r[1] - r[0]); start_off[r[2] + " q" + r[3]].append(r[0] - t0)
print("rounds %d..%d: period %.1f us" % (lo, hi, sum(per) / len(per) / 1e3))
for k in sorted(dur, key=lambda k: sum(start_off[k]) / len(start_off[k])):
    print("  %-60s start +%7.1f us  dur %6.1f us  (n %.2f per round)" % (k[:60], sum(start_off[k]) / len(start_off[k]) / 1e3, sum(dur[k]) / len(dur[k]) / 1e3, len(dur[k]) / (hi - lo)))
