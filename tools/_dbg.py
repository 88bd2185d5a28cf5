import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, helpers as H
from ropebwt2_amd import HipBwt
def check(reads, so=0, both=False):
    buf = H.encode_batch(reads, True, both)
    o = H.Oracle(so); g = HipBwt(so)
    o.insert_multi(buf); g.insert_multi(buf)
    ok = np.array_equal(o.counts(), g.counts())
    bad = []
    for b in range(6):
        if not np.array_equal(o.rope(b), g.rope(b)): bad.append((b, o.rope(b).tolist()[:30], g.rope(b).tolist()[:30]))
    print(reads if len(reads) < 8 else len(reads), "so", so, "counts", ok, "bad", bad[:2], flush=True)
print("start", flush=True)
rng = np.random.RandomState(0)
[rng.randint(1,5,size=1) for _ in range(30)]; [rng.randint(1,5,size=2) for _ in range(8)]
reads=[list(rng.randint(1,5,size=2)) for _ in range(30)]
buf = H.encode_batch(reads)
o = H.Oracle(0); g = HipBwt(0)
o.insert_multi(buf); g.insert_multi(buf)
r=o.rope(1)
print("oracle rope1", r.tolist()); print("dev    rope1", g.rope(1).tolist())
for x in range(0, len(r)+1):
    print(x, g.rank1a(1, x).tolist(), np.bincount(r[:x], minlength=6).tolist())
