"""Developer script: HIP engine vs oracle on a ladder of inputs (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import *
from ropebwt2_amd import HipBwt

def check(name, so, batches):
    o = Oracle(so); g = HipBwt(so)
    for i, buf in enumerate(batches):
        o.insert_multi(buf); g.insert_multi(buf)
        oc, gc = o.counts(), g.counts()
        if not (oc == gc).all():
            print("FAIL", name, "so", so, "batch", i, "counts\n", oc, "\n", gc); return False
        for b in range(6):
            ro, rg = o.rope(b), g.rope(b)
            if len(ro) != len(rg) or not (ro == rg).all():
                n = min(len(ro), len(rg)); d = np.flatnonzero(ro[:n] != rg[:n])
                print("FAIL", name, "so", so, "batch", i, "rope", b, "len", len(ro), len(rg), "first diff", d[:5])
                if len(d): j = d[0]; print(ro[max(0,j-10):j+10], rg[max(0,j-10):j+10])
                return False
    print("ok  ", name, "so", so, "total", o.L.orc_total(o.h))
    g.close()
    return True

ok = True
kat = text_to_reads(b"ACG\nTTA\nACG\nGNA\n\nC\n")
for so in (0, 1, 2):
    ok &= check("kat", so, [encode_batch(kat)])
    ok &= check("kat-both", so, [encode_batch(kat, True, True)])
    ok &= check("kat-2batches", so, [encode_batch(kat[:3]), encode_batch(kat[3:])])
if not ok: sys.exit(1)
for so in (0, 1, 2):
    reads = repetitive_reads(1500, seed=11 + so)
    ok &= check("rep1500", so, [encode_batch(reads)])
    ok &= check("rep1500x3", so, [encode_batch(reads[:500]), encode_batch(reads[500:1000]), encode_batch(reads[1000:], True, True)])
    reads = repetitive_reads(20000, seed=5 + so, genome_len=3000, max_len=120)
    ok &= check("rep20k-2b", so, [encode_batch(reads[:10000]), encode_batch(reads[10000:])])
if not ok: sys.exit(1)
for so in (0, 1, 2):
    codes = splitmix_bases(10000, 101)
    ok &= check("rand10k", so, [encode_batch_fixed(codes)])
    ok &= check("rand10k-2b", so, [encode_batch_fixed(codes[:5000]), encode_batch_fixed(codes[5000:])])
    codes = splitmix_bases(100, 3000, seed=44)
    ok &= check("long100x3000", so, [encode_batch_fixed(codes)])
print("ALL OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
