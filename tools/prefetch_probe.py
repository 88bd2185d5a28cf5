#!/usr/bin/env python3
"""Where does rb2_hip_prefetch pay?  Three 4.2 GB batches through the host-buffer API: no prefetch, the next batch announced from a
second thread while the current one is inserted (one piece / 256 MB pieces), and announced BEFORE the current insert starts.
Measured on one MI355X box: a pageable host-to-device copy that runs beside the insert kernels takes twice as long (0.074 -> 0.15 s for
4.2 GB) and costs the insert ~0.14 s -- an API caller that fires batches back to back gains 7 %; the CLI, whose reader produces a
batch over seconds, hides the whole crossing (profiles/r03_configs2_full_cli_1gpu.txt)."""
import sys, time, threading, numpy as np
sys.path.insert(0, '/root/repo')
from ropebwt2_amd import HipBwt
L=101; n=40_844_297
b = HipBwt(1, 0)
p = b.dev_alloc(n*(L+1))
host=[]
for k in range(3):
    b.synth_reads(p, k*n, n, L, seed=42); b.sync()
    a = np.empty(n*(L+1), np.uint8); b.L.rb2_hip_memcpy(b.h, a.ctypes.data, p, n*(L+1), 1); host.append(a)
b.dev_free(p)
def pf(a, chunk):
    t0=time.perf_counter()
    if chunk:
        for o in range(chunk, len(a)+chunk, chunk):
            b.L.rb2_hip_prefetch(b.h, a.ctypes.data, min(o,len(a)), len(a))
    else:
        b.L.rb2_hip_prefetch(b.h, a.ctypes.data, len(a), len(a))
    print("  prefetch call(s) took %.3f s (chunk %s)" % (time.perf_counter()-t0, chunk), flush=True)
# 1. prefetch alone
pf(host[0], 0); t0=time.perf_counter(); b.insert_multi(host[0]); b.sync(); print("insert after full prefetch %.3f" % (time.perf_counter()-t0))
b.reset()
for chunk in (0, 256<<20):
    t0=time.perf_counter()
    for k,a in enumerate(host):
        th=None
        if k+1 < len(host):
            th=threading.Thread(target=pf, args=(host[k+1], chunk)); th.start()
        t1=time.perf_counter(); b.insert_multi(a); b.sync(); print("  insert %d %.3f" % (k, time.perf_counter()-t1), flush=True)
        if th: th.join()
    print("job with prefetch chunk %s: %.3f s" % (chunk, time.perf_counter()-t0), flush=True)
    b.reset()
t0=time.perf_counter()
for a in host: b.insert_multi(a)
b.sync(); print("job without prefetch %.3f" % (time.perf_counter()-t0))
# no concurrency at all: the next batch is prefetched BEFORE the current insert starts
b.reset()
t0=time.perf_counter()
for k,a in enumerate(host):
    t1=time.perf_counter(); b.insert_multi(a); b.sync(); print("  serial: insert %d %.3f" % (k, time.perf_counter()-t1), flush=True)
    if k+1 < len(host): pf(host[k+1], 0)
print("job with serial prefetch: %.3f s" % (time.perf_counter()-t0), flush=True)
b.reset(); b.reserve(len(host[0]), n, sum(len(a) for a in host))
t0=time.perf_counter()
for k,a in enumerate(host):
    t1=time.perf_counter(); b.insert_multi(a); b.sync(); print("  reserved, serial: insert %d %.3f" % (k, time.perf_counter()-t1), flush=True)
    if k+1 < len(host): pf(host[k+1], 0)
print("job with serial prefetch after reserve: %.3f s" % (time.perf_counter()-t0), flush=True)
