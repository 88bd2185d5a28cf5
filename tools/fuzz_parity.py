#!/usr/bin/env python3
"""Randomised differential run on the GPU box: random batch sequences (orders, read shapes, strands, N's, duplicates, empty strings,
homopolymers, index sizes that cross the dense/sparse switch), through random engine settings (sparse layout forced or default, lazy
host inserts on/off, device or host buffers, 1-8 virtual ranks behind one handle) -- ropes and count matrix against the oracle after
every batch.  Test infrastructure (uses oracle/ through tests/helpers.py); stops at the first mismatch and prints the seed.
usage: fuzz_parity.py [seconds=300] [first_seed=1]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from ropebwt2_amd import HipBwt, MultiBwt, build_all
build_all(); H.build_oracle()

def gen_batch(rng, so):
    kind = rng.choice(["fixed", "var", "repet", "homo", "tiny", "long", "big", "cover", "shorts", "nruns"], p=[.22, .16, .16, .1, .08, .1, .06, .04, .04, .04])
    both = bool(rng.rand() < 0.4)
    if kind == "fixed":
        codes = H.splitmix_bases(int(rng.randint(1, 6000)), int(rng.choice([1, 2, 17, 50, 101, 150])), seed=int(rng.randint(1, 1 << 30)))
        return H.encode_batch_fixed(codes, True, both)
    if kind == "shorts":                                          # very short reads: a sentinel every few symbols -- windows of the dense layout overflow their exception lists and stay plain
        return H.encode_batch([rng.randint(1, 5, size=int(n)).astype(np.uint8) for n in rng.randint(1, 8, size=int(rng.randint(5000, 30000)))], True, both)
    if kind == "nruns":                                           # runs of N between stretches of ACGT
        return H.encode_batch([np.concatenate([np.full(int(rng.randint(100, 500)), 5, np.uint8), rng.randint(1, 5, size=60).astype(np.uint8)]) for _ in range(int(rng.randint(50, 400)))], True, both)
    if kind == "big":                                             # enough strings for the dense regime on a grown index, and for several tiles per piece
        return H.encode_batch_fixed(H.splitmix_bases(int(rng.randint(20000, 60000)), 101, seed=int(rng.randint(1, 1 << 30))), True, both)
    if kind == "cover":                                           # overlapping reads of one genome: non-empty intervals for many rounds, large groups
        g = rng.randint(1, 5, size=3000).astype(np.uint8); L = int(rng.choice([50, 101]))
        st = rng.randint(0, len(g) - L, size=int(rng.randint(3000, 15000)))
        return H.encode_batch_fixed(np.stack([g[x:x + L] for x in st]), True, both)
    if kind == "var":
        n = int(rng.randint(1, 3000))
        reads = [rng.randint(1, 6, size=int(rng.choice([0, 1, 3, 30, 90, 400], p=[.05, .1, .1, .3, .35, .1]))).astype(np.uint8) for _ in range(n)]
        return H.encode_batch(reads, True, both)
    if kind == "repet":
        ml = int(rng.choice([20, 60, 150]))
        return H.encode_batch(H.repetitive_reads(int(rng.randint(50, 4000)), seed=int(rng.randint(1, 1 << 20)), genome_len=max(int(rng.choice([60, 300, 2000])), ml + 10), max_len=ml), True, both)
    if kind == "homo":
        return H.encode_batch([[int(rng.randint(1, 5))] * int(rng.randint(500, 20000))] * int(rng.randint(1, 4)) + [list(rng.randint(1, 5, size=40))] * int(rng.randint(0, 50)), True, both)
    if kind == "tiny":
        return H.encode_batch([[], [1], [], [4, 4], [5]][: int(rng.randint(1, 6))], True, both)
    return H.encode_batch([rng.randint(1, 5, size=int(rng.randint(2000, 12000))).astype(np.uint8) for _ in range(int(rng.randint(1, 40)))], True, both)

def one(seed):
    rng = np.random.RandomState(seed)
    so = int(rng.randint(0, 3))
    env = {}
    if rng.rand() < 0.4: env.update(RB2_SPARSE_LAMBDA="1e18", RB2_SPARSE_MAXPEN="0")
    elif rng.rand() < 0.2: env.update(RB2_SPARSE_LAMBDA="0")
    if rng.rand() < 0.3: env.update(RB2_SPARSE_HEAD=str(int(rng.choice([0, 1, 3]))))
    if rng.rand() < 0.3: env.update(RB2_LEAF_PIPE=str(int(rng.choice([0, 64, 8192]))))
    if rng.rand() < 0.2: env.update(RB2_COMPACT="0")               # round 5: windows of the dense layout never compact
    if rng.rand() < 0.25: env.update(RB2_POS="64")                 # positions in 64-bit storage from the start
    if rng.rand() < 0.2:
        env.update(RB2_TS_MAX="2")                                 # the many-tiles counting kernels for every batch of more than 1024 strings
        if rng.rand() < 0.4: env.update(RB2_TS_FOLD="0")           # ... with the scan over the chunk totals as a launch of its own
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    nr = int(rng.choice([1, 1, 1, 2, 3, 8]))
    desc = dict(seed=seed, so=so, env=env, ranks=nr)
    try:
        o = H.Oracle(so)
        dev = HipBwt(so) if nr == 1 else MultiBwt(so, [0] * nr, "peer")
        if nr == 1: dev.set_lazy(int(rng.rand() < 0.6))
        nb = int(rng.randint(1, 6))
        for i in range(nb):
            buf = gen_batch(rng, so)
            o.insert_multi(buf)
            if nr == 1 and rng.rand() < 0.3:
                p = dev.dev_alloc(len(buf) + 64)
                dev.L.rb2_hip_memcpy(dev.h, p, buf.ctypes.data, len(buf), 0)
                dev.insert_multi_dev(p, len(buf)); dev.dev_free(p)
            else:
                dev.insert_multi(buf)
            if rng.rand() < 0.5 or i == nb - 1:
                if not np.array_equal(dev.counts(), o.counts()): return dict(desc, fail="counts", batch=i)
                for b in range(6):
                    if not np.array_equal(dev.rope(b), o.rope(b)): return dict(desc, fail="rope %d" % b, batch=i)
        dev.close()
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    return None

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        r = one(seed)
        if r is not None:
            print(json.dumps(dict(r, env=r["env"]))); print("MISMATCH after %d cases" % n); sys.exit(1)
        seed += 1; n += 1
    print("fuzz_parity: %d cases (seeds %d..%d) in %.0f s, all equal to the oracle" % (n, seed - n, seed - 1, time.time() - t0))

if __name__ == "__main__":
    main()
