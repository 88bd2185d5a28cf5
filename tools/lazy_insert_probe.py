#!/usr/bin/env python3
"""configs[1] through rb2_hip_insert_multi on pageable host buffers, batches fired back to back: the call returning only when the
device is done (RB2_HIP_LAZY_INSERT=0) against the default, where it returns once the text is uploaded and the rounds are queued and
the NEXT batch crosses PCIe beside them.  Same counts and rope checksums; wall time of the whole job."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from ropebwt2_amd import HipBwt, build_all
build_all()
L, per_batch, reads = 101, 40844297, int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
gen = HipBwt(1, 0)
p = gen.dev_alloc(per_batch * (L + 1) + 64)
host, done = [], 0
while done < reads:
    n = min(per_batch, reads - done)
    gen.synth_reads(p, done, n, L, seed=42); gen.sync()
    a = np.empty(n * (L + 1), np.uint8)
    gen.L.rb2_hip_memcpy(gen.h, a.ctypes.data, p, len(a), 1)
    host.append(a); done += n
gen.dev_free(p); gen.close()
out = {}
for rep in range(2):
    for lazy in (0, 1):
        b = HipBwt(1, 0)
        b.set_lazy(lazy)
        if len(sys.argv) > 2 and sys.argv[2] == "reserve":
            b.reserve(per_batch * (L + 1), per_batch, reads * (L + 1))
        b.sync()
        t0 = time.perf_counter()
        ret = []
        for a in host:
            t1 = time.perf_counter(); b.insert_multi(a); ret.append(round(time.perf_counter() - t1, 3))
        b.wait()
        dt = time.perf_counter() - t0
        c = b.counts()
        out["lazy=%d rep %d" % (lazy, rep)] = {"seconds": round(dt, 3), "gsym_per_s": round(reads * (L + 1) / dt / 1e9, 2), "calls_returned_after_s": ret,
                                            "counts_sum_ok": int(c.sum()) == reads * (L + 1), "hashes": [hex(x) for x in b.rope_hashes()]}
        b.close()
hs = {json.dumps(v["hashes"]) for v in out.values()}
out["all_runs_same_ropes"] = len(hs) == 1
print(json.dumps(out, indent=1))
