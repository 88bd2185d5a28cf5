#!/usr/bin/env python3
"""Randomised differential run of the whole program on the GPU box: our CLI (GPU inserts, random small -m so that a run has many batches,
threaded readers) against the REAL reference binary (oracle/_ref/ropebwt2, its default -m) on random FASTQ / FASTA / line inputs with N's,
lower case, palindromes, empty reads, duplicates and repeats, and random flag sets (-s/-r, -R/-F, -N, -x, -C, -q, both strands).
`-d`: the .fmd bytes must be equal; `-b`: our .fmr, converted by the reference (-d -i), must give the reference's .fmd.
Test infrastructure (needs oracle/_ref, which exists where /root/reference was present at build time).  usage: fuzz_cli.py [seconds=300] [seed=1]"""
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

def gen(rng):
    n = int(rng.choice([1, 5, 60, 400, 2500]))
    genome = "".join(rng.choice(list("ACGT"), size=600))
    recs = []
    for i in range(n):
        L = int(rng.choice([0, 1, 2, 8, 35, 101, 300], p=[.03, .04, .05, .1, .3, .4, .08]))
        if rng.rand() < 0.4 and L:
            st = int(rng.randint(0, len(genome) - L)) if L < len(genome) else 0
            s = genome[st:st + L]
        else:
            s = "".join(rng.choice(list("ACGTNacgtn"), size=L, p=[.23, .23, .23, .23, .02, .015, .015, .015, .015, 0.0]))
        if rng.rand() < 0.05 and L >= 4 and L % 2 == 0:
            h = s[:L // 2].upper().replace("N", "A"); s = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))
        if rng.rand() < 0.05 and recs: s = recs[int(rng.randint(len(recs)))][0]
        q = "".join(chr(33 + int(x)) for x in rng.randint(2, 41, size=len(s)))
        recs.append((s, q))
    fmt = rng.choice(["fq", "fa", "fa1", "line"])
    if fmt == "fq": return fmt, "".join("@r%d d\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(recs)).encode()
    if fmt == "fa": return fmt, "".join(">r%d\n%s\n" % (i, "\n".join(s[j:j + 60] for j in range(0, len(s), 60))) for i, (s, q) in enumerate(recs)).encode()
    if fmt == "fa1": return fmt, "".join(">r%d x\n%s\n" % (i, s) for i, (s, q) in enumerate(recs)).encode()
    return fmt, "".join(s + "\n" for s, q in recs).encode()

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    d = tempfile.mkdtemp()
    t0, n, bad = time.time(), 0, 0
    while time.time() - t0 < secs:
        fmt, data = gen(rng)
        flags = []
        if fmt == "line": flags.append("-L")
        flags += [[], ["-s"], ["-r"]][rng.randint(3)]
        flags += [[], ["-R"], ["-F"]][rng.choice(3, p=[.4, .5, .1])]
        if rng.rand() < 0.2: flags.append("-N")
        if rng.rand() < 0.2: flags += ["-x", str(int(rng.choice([1, 4, 20])))]
        if rng.rand() < 0.2 and b"\n\n" not in data and fmt != "line": flags.append("-C")
        if rng.rand() < 0.2 and fmt == "fq": flags += ["-q", str(int(rng.choice([5, 20, 35])))]
        ours_m = [[], ["-m", "1k"], ["-m", "20k"], ["-m", "1m"]][rng.randint(4)]
        env = dict(os.environ, RB2_PARSE_THREADS=str(int(rng.choice([1, 3, 8]))), RB2_PARSE_CHUNK=str(int(rng.choice([64, 5000, 1 << 24]))))
        if rng.rand() < 0.3: env.update(RB2_SPARSE_LAMBDA="1e18", RB2_SPARSE_MAXPEN="0")
        ref = subprocess.run([H.REF_BIN] + flags + ["-d", "-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if ref.returncode != 0:            # (the reference asserts on a few degenerate inputs: not a comparison)
            continue
        if rng.rand() < 0.6:
            p = subprocess.run([CLI] + flags + ours_m + ["-d", "-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            got, how = p.stdout, "-d"
        else:
            f = os.path.join(d, "o.fmr")
            p = subprocess.run([CLI] + flags + ours_m + ["-b", "-o", f, "-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            c = subprocess.run([H.REF_BIN, "-d", "-i", f, "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            got, how = c.stdout, "-b -> ref -d -i"
        n += 1
        if p.returncode != 0 or got != ref.stdout:
            bad += 1
            keep = "/tmp/fuzz_cli_fail_%d.%s" % (bad, fmt)
            open(keep, "wb").write(data)
            print("MISMATCH", how, " ".join(flags + ours_m), "rc", p.returncode, len(got), len(ref.stdout), keep, {k: env[k] for k in env if k.startswith("RB2_")}, p.stderr.decode()[-200:])
            if bad >= 5: break
    print("fuzz_cli: %d cases in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
    sys.exit(1 if bad else 0)

if __name__ == "__main__":
    main()
