#!/usr/bin/env python3
"""Randomised differential run of the incremental flow (configs[4]'s shape) on the GPU box: the first part of a random read set is built into an
.fmr by the REAL reference or by our CLI (random -l / -n leaf and bucket sizes, random order), our CLI restores it (-i: run bytes decoded on the
device) and inserts the rest -- on one GPU or sharded over 2-8 virtual ranks (RB2_HIP_DEVICES=0,0,..) -- and the .fmd must be what the reference
makes of the same .fmr and the same second part (and its one-shot .fmd of all reads: SURVEY.md 8c, incremental == one-shot).  usage: fuzz_cli_incremental.py [seconds=300] [seed=1]"""
import os, sys, subprocess, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from ropebwt2_amd import build_all
build_all()
CLI = os.path.join(ROOT, "ropebwt2_amd", "bin", "ropebwt2")
if not H.have_ref():
    print("oracle/_ref/ropebwt2 is not built here: nothing to compare with"); sys.exit(0)

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    d = tempfile.mkdtemp()
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        nreads = int(rng.choice([2, 40, 600, 4000]))
        genome = rng.randint(1, 5, size=500).astype(np.uint8)
        lines = []
        for i in range(nreads):
            L = int(rng.choice([0, 1, 12, 60, 101, 400], p=[.03, .05, .12, .3, .4, .1]))
            if rng.rand() < 0.5 and 0 < L < 500:
                st = int(rng.randint(0, 500 - L)); codes = genome[st:st + L]
            else:
                codes = rng.choice([1, 2, 3, 4, 5], size=L, p=[.245, .245, .245, .245, .02]).astype(np.uint8)
            lines.append(bytes(H.SYMS[codes]) + b"\n")
        cut = int(rng.randint(0, nreads + 1))
        a, b = b"".join(lines[:cut]), b"".join(lines[cut:])
        so = [[], ["-s"], ["-r"]][rng.randint(3)]
        strands = [[], ["-R"]][rng.randint(2)]
        base = ["-L"] + so + strands
        want = subprocess.run([H.REF_BIN] + base + ["-d", "-"], input=a + b, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if want.returncode != 0: continue
        leaf = [[], ["-l", str(int(rng.choice([32, 64, 200, 1024]))), "-n", str(int(rng.choice([4, 6, 16, 64])))]][rng.randint(2)]
        f = os.path.join(d, "half.fmr")
        builder = H.REF_BIN if rng.rand() < 0.5 else CLI
        m1 = [] if builder == H.REF_BIN else [[], ["-m", "5k"], ["-m0"]][rng.randint(3)]
        if os.environ.get("FUZZ_M0") and builder != H.REF_BIN: m1 = ["-m0"]
        p1 = subprocess.run([builder] + base + leaf + m1 + ["-b", "-o", f, "-"], input=a, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if p1.returncode != 0:
            if builder == H.REF_BIN: continue
            print("our build of the first part failed", p1.stderr.decode()[-300:]); bad += 1; continue
        env = dict(os.environ)
        nr = int(rng.choice([1, 1, 2, 3, 8]))
        if nr > 1: env["RB2_HIP_DEVICES"] = ",".join(["0"] * nr)
        if rng.rand() < 0.3: env.update(RB2_SPARSE_LAMBDA="1e18", RB2_SPARSE_MAXPEN="0")
        m2 = [[], ["-m", "3k"], ["-m", "100k"]][rng.randint(3)]
        if os.environ.get("FUZZ_M0"): m2, env = ["-m0"], dict(os.environ)      # (no GPU here: the second part goes in string by string on the host)
        p2 = subprocess.run([CLI] + base + m2 + ["-d", "-i", f, "-"], input=b, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        # the oracle of the second phase is the reference restoring the SAME .fmr and reading the SAME second part (kseq's end-of-input quirk makes
        # an empty -- or k x 16384-byte -- part one empty read: the one-shot build of a + b does not see it); the one-shot .fmd must agree whenever no
        # part can trigger the quirk
        inc = subprocess.run([H.REF_BIN] + base + ["-d", "-i", f, "-"], input=b, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if inc.returncode != 0: continue
        if len(a) % 16384 and len(b) % 16384 and len(a + b) % 16384 and inc.stdout != want.stdout:
            print("the reference's incremental build differs from its one-shot build?", " ".join(base + leaf), len(a), len(b)); bad += 1
        want = inc
        n += 1
        if p2.returncode != 0 or p2.stdout != want.stdout:
            bad += 1
            open("/tmp/fuzz_inc_fail_%d_a.txt" % bad, "wb").write(a); open("/tmp/fuzz_inc_fail_%d_b.txt" % bad, "wb").write(b)
            print("MISMATCH", " ".join(base + leaf), "first part by", os.path.basename(os.path.dirname(builder)) or builder, m1, "then", m2, "ranks", nr, "rc", p2.returncode,
                  len(p2.stdout), len(want.stdout), p2.stderr.decode()[-200:])
            if bad >= 5: break
    print("fuzz_cli_incremental: %d cases in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
