#!/usr/bin/env python3
"""PCIe-inclusive rate of the host API: rb2_hip_insert_multi(host buffer) for the batches of configs[1]
(the path mr_insert_multi takes); the batch is synthesised on the device, copied to pageable host memory, then
inserted from there.  Never the `value` of bench.py -- quoted in DESIGN.md section 6."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ropebwt2_amd import HipBwt, build_all
build_all()
L, per_batch, reads = 101, 40844297, 100_000_000
bwt = HipBwt(1, 0)
bwt.reserve(per_batch * (L + 1), per_batch, reads * (L + 1))
p = bwt.dev_alloc(per_batch * (L + 1) + 64)
done, t_ins, t_all = 0, 0.0, []
while done < reads:
    n = min(per_batch, reads - done)
    nb = n * (L + 1)
    bwt.synth_reads(p, done, n, L, seed=42); bwt.sync()
    host = np.empty(nb, np.uint8)
    bwt.L.rb2_hip_memcpy(bwt.h, host.ctypes.data, p, nb, 1)
    t0 = time.perf_counter()
    bwt.insert_multi(host)
    bwt.sync()
    t_all.append(time.perf_counter() - t0)
    done += n
c = bwt.counts()
print(json.dumps({"symbols": reads * (L + 1), "insert_s_incl_pcie": sum(t_all), "gsym_per_s": reads * (L + 1) / sum(t_all) / 1e9,
                  "batch_s": [round(t, 3) for t in t_all], "counts_ok": int(c.sum()) == reads * (L + 1)}))
