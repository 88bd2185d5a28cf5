import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from helpers import *
from ropebwt2_amd import HipBwt
rng = np.random.RandomState(1)
rng.randint(1, 5, size=16)
reads = [[int(x)] for x in rng.randint(1, 5, size=17)]
buf = encode_batch(reads)
o = Oracle(0); o.insert_multi(buf)
g = HipBwt(0); g.insert_multi(buf)
exp = o.rope(0); rle = g.rope_rle(0)
print("expected", exp.tolist())
print("   got bytes (len,sym):", [(int(b)>>3, int(b)&7) for b in rle])
