"""configs[1] as three device-generated batches on one engine; prints, per batch, the time and the window-format statistics
(RB2_COMPACT_STATS=1 to have the formats counted: slows the merge).  With RB2_HIP_TRACE=1 the engine prints per-round kernel times."""
import os, sys, time
sys.path.insert(0, os.getcwd())
from ropebwt2_amd import HipBwt
n, L = 100_000_000, 101
per = (4 << 30) // (L + 1)       # reads per -m4g batch (approx.)
dev = HipBwt(1)
first = 0
while first < n:
    cnt = min(per, n - first)
    p = dev.dev_alloc(cnt * (L + 1))
    dev.synth_reads(p, first, cnt, L, seed=42)
    dev.sync()
    t0 = time.time()
    dev.insert_multi_dev(p, cnt * (L + 1))
    dev.sync()
    st = dev.window_stats() if hasattr(dev.L, "rb2_hip_window_stats") else None
    print("batch", first, cnt, "%.3f s" % (time.time() - t0), st, flush=True)
    dev.dev_free(p)
    first += cnt
