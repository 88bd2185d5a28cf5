"""Where the time of a configs[1] job through the host-buffer entry point goes: rb2_hip_insert_multi per -m4g batch (upload inside the
call, rounds queued behind it), wall-clock after every call and after the final wait.  Variants: pageable / registered buffers, with
and without rb2_hip_reserve before the clock, a warm handle (second job on a handle that has grown its buffers and was cleared is
not possible -- the index only grows -- so 'warm' = a second handle after the first one was closed: what the runtime caches)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from ropebwt2_amd import HipBwt
n, L = int(os.environ.get("N", 100_000_000)), 101
per = (4 << 30) // (L + 1)
gen = HipBwt(1)
host = []
first = 0
while first < n:
    cnt = min(per, n - first)
    p = gen.dev_alloc(cnt * (L + 1))
    gen.synth_reads(p, first, cnt, L, seed=42)
    a = np.empty(cnt * (L + 1), np.uint8)
    gen.L.rb2_hip_memcpy(gen.h, a.ctypes.data, p, a.nbytes, 1)
    gen.dev_free(p)
    host.append(a)
    first += cnt
gen.close()
tot = sum(a.nbytes for a in host)

def job(tag, pinned=False, reserve=False):
    b = HipBwt(1)
    if pinned:
        for a in host: assert b.L.rb2_hip_host_register(a.ctypes.data, a.nbytes) == 0
    if reserve: b.L.rb2_hip_reserve(b.h, max(a.nbytes for a in host), per, tot)
    b.sync()
    t0 = time.perf_counter(); ts = []
    for a in host:
        b.insert_multi(a); ts.append(time.perf_counter() - t0)
    b.sync(); ts.append(time.perf_counter() - t0)
    print("%-28s calls return at %s, done at %.3f s = %.2f Gsym/s" % (tag, " ".join("%.3f" % t for t in ts[:-1]), ts[-1], tot / ts[-1] / 1e9), flush=True)
    if pinned:
        for a in host: b.L.rb2_hip_host_unregister(a.ctypes.data)
    b.close()

for rep in range(int(os.environ.get("REPS", 2))):
    job("pageable"); job("pageable + reserve", reserve=True); job("registered", pinned=True); job("registered + reserve", pinned=True, reserve=True)
