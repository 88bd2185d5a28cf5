for e in 0 1 0 1; do
  if [ $e = 1 ]; then export RB2_NO_PAR_UPLOAD=1; else unset RB2_NO_PAR_UPLOAD; fi
  python - <<'PY'
import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ropebwt2_amd import HipBwt
L, per, reads = 101, 40844297, 100_000_000
g = HipBwt(1, 0)
p = g.dev_alloc(per * (L + 1) + 64)
hosts, done = [], 0
while done < reads:
    n = min(per, reads - done); nb = n * (L + 1)
    g.synth_reads(p, done, n, L, seed=42); g.sync()
    a = np.empty(nb, np.uint8); g.L.rb2_hip_memcpy(g.h, a.ctypes.data, p, nb, 1); hosts.append(a); done += n
g.dev_free(p); g.close()
for rep in range(2):
    b = HipBwt(1, 0)
    t0 = time.perf_counter()
    for a in hosts: b.insert_multi(a)
    b.sync()
    dt = time.perf_counter() - t0
    b.close()
    print("no_par" if os.environ.get("RB2_NO_PAR_UPLOAD") else "par   ", rep, round(dt, 3), "s", round(reads * (L + 1) / dt / 1e9, 2), "Gsym/s")
PY
done
