#!/usr/bin/env python3
"""Per-kernel sums of every counter in a rocprofv3 --pmc counter_collection CSV.
usage: pmc_summary.py <counter_collection.csv> [<more.csv> ...]"""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
nl = collections.defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        nl[k].add(r.get("Dispatch_Id", r.get("Correlation_Id")))
names = sorted({c for v in agg.values() for c in v})
print(",".join(["kernel", "launches"] + names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
    print(",".join([k, str(len(nl[k]))] + ["%.4g" % v.get(c, 0) for c in names]))
