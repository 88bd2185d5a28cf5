#!/usr/bin/env python3
"""Time and peak memory of the REAL reference (oracle/_ref/ropebwt2) on a synthetic job: what bench.py's compressible-input line
(secondary.coverage30x_1gpu) is compared with.  Writes the numbers into tests/golden/golden_large.json next to the job's md5
(key reference_run).     python tools/ref_footprint.py coverage30x"""
import json, os, re, resource, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF, GEN = os.path.join(ROOT, "oracle", "_ref", "ropebwt2"), os.path.join(ROOT, "ropebwt2_amd", "bin", "synth_reads")
GOLD = os.path.join(ROOT, "tests", "golden", "golden_large.json")
what = sys.argv[1] if len(sys.argv) > 1 else "coverage30x"
g = json.load(open(GOLD))
job = g[what]
flags = "-LRbs -m1g"
gen = subprocess.Popen([GEN, str(job["n_reads"]), str(job["read_len"]), str(job["seed"]), "0", str(job.get("genome_len", 0))], stdout=subprocess.PIPE)
t0 = time.time()
p = subprocess.run([REF] + flags.split() + ["-o", "/dev/null", "-"], stdin=gen.stdout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
dt = time.time() - t0
gen.wait()
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
err = p.stderr.decode()
ins = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"inserted (\d+) symbols in ([0-9.]+) sec", err)]
syms = sum(s for s, _ in ins)
rec = {"flags": flags, "real_s": round(dt, 1), "insert_s": round(sum(t for _, t in ins), 1), "symbols": syms,
       "max_rss_bytes": ru.ru_maxrss * 1024, "rss_bytes_per_symbol": round(ru.ru_maxrss * 1024 / max(1, syms), 4),
       "threads": 5, "host": "build container, %d cores" % (os.cpu_count() or 0),
       "note": "peak RSS includes the 1 GiB read batch and its 24-byte-per-read sort records; the run-length B+ trees are the rest"}
print(json.dumps(rec))
g[what]["reference_run"] = rec
json.dump(g, open(GOLD, "w"), indent=1)
