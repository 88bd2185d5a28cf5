#!/usr/bin/env python3
"""Per-kernel sums of every counter in one or more rocprofv3 --pmc counter_collection CSVs (one CSV per pass; a counter that was
collected in several passes -- SQ_WAVES usually -- is averaged over them).  Adds VALU / SALU instructions per wave.
usage: pmc_summary.py <counter_collection.csv> [<more.csv> ...]"""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> counter -> sum over its launches (mean over passes)
nl = collections.defaultdict(int)
for path in sys.argv[1:]:
    one = collections.defaultdict(lambda: collections.defaultdict(float))
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        one[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[k].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
    for k, v in one.items():
        nl[k] = max(nl[k], len(ids[k]))
        for c, x in v.items():
            agg[k][c] = x if c not in agg[k] else (agg[k][c] + x) / 2
names = sorted({c for v in agg.values() for c in v})
print(",".join(["kernel", "launches"] + names + ["VALU_per_wave", "SALU_per_wave"]))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    w = v.get("SQ_WAVES", 0)
    per = lambda c: ("%.1f" % (v[c] / w)) if w and c in v else ""
    print(",".join(['"%s"' % k, str(nl[k])] + ["%.5g" % v.get(c, 0) for c in names] + [per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU")]))
