#!/usr/bin/env python3
"""Summarise rocprofv3 FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) per kernel.
usage: traffic_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> <k_merge_traffic.json> [command]
Units/corrections follow MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is taken
as is (it matches the bytes k_merge is known to write within 2 %)."""
import csv, collections, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_sha():                                  # same as bench.py: which kernel sources the counters belong to
    h = hashlib.sha1()
    for f in ("rb2_merge.h", "rb2_kernels.h", "rb2_device.h"):
        h.update(open(os.path.join(ROOT, "ropebwt2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


fetch, write, out_csv, out_json = sys.argv[1:5]
cmd = sys.argv[5] if len(sys.argv) > 5 else "python bench.py --no-cpu-baseline --warmup 0"
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for path, col in ((fetch, 0), (write, 1)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][col] += float(r["Counter_Value"])
        if col == 0:
            agg[k][2] += 1
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- %s; bytes = KiB*1024, FETCH doubled (gfx950)\n" % cmd)
    w.writerow(["kernel", "launches", "fetch_GB_corrected", "write_GB", "total_GB_per_launch"])
    for k, (fe, wr, n) in sorted(agg.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1])):
        if n == 0:
            continue
        fb, wb = 2 * fe * 1024, wr * 1024
        w.writerow([k, n, "%.2f" % (fb / 1e9), "%.2f" % (wb / 1e9), "%.4f" % ((fb + wb) / n / 1e9)])
km = [k for k in agg if k.split("<")[0].endswith("k_merge")]
fe = sum(agg[k][0] for k in km); wr = sum(agg[k][1] for k in km); n = sum(agg[k][2] for k in km)
json.dump({"kernel": "k_merge", "launches": n, "src_sha": src_sha(), "bytes_per_launch": (2 * fe + wr) * 1024 / max(n, 1),
           "fetch_bytes_per_launch_corrected": 2 * fe * 1024 / max(n, 1), "write_bytes_per_launch": wr * 1024 / max(n, 1),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `%s` (%d k_merge launches); FETCH_SIZE x2 per MI355X_MICROARCH.md" % (cmd, n)},
          open(out_json, "w"), indent=1)
