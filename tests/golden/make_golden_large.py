#!/usr/bin/env python3
"""Large golden vector, produced by the REAL reference (oracle/_ref/ropebwt2): md5 of the .fmd of
10 M x 101 bp reads of the SURVEY.md 8c stream (1.02 G symbols), input streamed from ropebwt2_amd/bin/synth_reads (tools/synth_reads.c).
Only digests are stored.  Run in the build container:  python tests/golden/make_golden_large.py"""
import hashlib, json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF, GEN = os.path.join(ROOT, "oracle", "_ref", "ropebwt2"), os.path.join(ROOT, "ropebwt2_amd", "bin", "synth_reads")
N, L, SEED = 10_000_000, 101, 42
out = {"generator": "tests/golden/make_golden_large.py", "reference": "lh3/ropebwt2 r187 (oracle/_ref)",
       "n_reads": N, "read_len": L, "seed": SEED, "fmd_md5": {}}
if os.path.exists(os.path.join(HERE, "golden_large.json")):           # keep entries produced by other invocations
    if len(sys.argv) > 1:                                             # a sub-mode leaves the 10 M digests alone
        out["fmd_md5"] = json.load(open(os.path.join(HERE, "golden_large.json"))).get("fmd_md5", {})
    out.update({k: v for k, v in json.load(open(os.path.join(HERE, "golden_large.json"))).items() if k.startswith("configs") or k.startswith("coverage") or k.startswith("longreads") or k.startswith("skewed")})
if "--longreads" in sys.argv:
    # long-read path: 200 k x 5 kbp, input order, one batch of 5001 rounds
    g = subprocess.Popen([GEN, "200000", "5000", "44"], stdout=subprocess.PIPE)
    p = subprocess.Popen([REF, "-LRd", "-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h = hashlib.md5()
    for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(chunk)
    assert p.wait() == 0 and g.wait() == 0
    out["longreads"] = {"n_reads": 200000, "read_len": 5000, "seed": 44, "flags": "-LRd", "fmd_md5": h.hexdigest()}
    json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
    sys.exit(0)
if "--coverage" in sys.argv:
    # overlapping reads: 30 M x 101 bp windows of one random 100 Mb genome (30x), RLO forward and RCLO both strands
    cov = {"n_reads": 30000000, "read_len": 101, "seed": 42, "genome_len": 100000000, "runs": {}}
    for flags in ("-LRds -m1g", "-Lrd -m2g"):
        g = subprocess.Popen([GEN, "30000000", "101", "42", "0", "100000000"], stdout=subprocess.PIPE)
        p = subprocess.Popen([REF] + flags.split() + ["-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        h, n = hashlib.md5(), 0
        for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
            h.update(chunk); n += len(chunk)
        assert p.wait() == 0 and g.wait() == 0
        cov["runs"][flags] = {"fmd_bytes": n, "fmd_md5": h.hexdigest()}
    out["coverage30x"] = cov
    json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
    sys.exit(0)
if "--skewed" in sys.argv:
    # The > 2^32 regime pinned to the real reference: 60 M x 101 bp of SKEWED composition (85 % A: synth_reads ... 1), so that sub-rope
    # (A,A) of the device index -- the A's of rope A -- holds 4.4 G symbols, more than 32 bits of piece-relative position, at a size the
    # reference builds in minutes (run-length leaves keep its memory small).  RLO and RCLO, forward strand.
    N = 60_000_000
    sk = {"n_reads": N, "read_len": 101, "seed": 42, "skew": 1, "runs": {},
          "provenance": "oracle/_ref/ropebwt2 (the real reference) in the build container: synth_reads 60000000 101 42 0 0 0 1 | ropebwt2 <flags> - ; make_golden_large.py --skewed"}
    for flags in ("-LRds", "-LRdr"):
        g = subprocess.Popen([GEN, str(N), "101", "42", "0", "0", "0", "1"], stdout=subprocess.PIPE)
        p = subprocess.Popen([REF] + flags.split() + ["-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        h, n = hashlib.md5(), 0
        for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
            h.update(chunk); n += len(chunk)
        assert p.wait() == 0 and g.wait() == 0
        sk["runs"][flags] = {"fmd_bytes": n, "fmd_md5": h.hexdigest()}
        print(flags, n, h.hexdigest(), file=sys.stderr)
    out["skewed_60M"] = sk
    json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
    sys.exit(0)
if "--configs2-order" in sys.argv:
    # BASELINE.json configs[2]'s ORDER AND FLAGS at configs[1]'s size: 100 M x 101 bp, RCLO, forward strand (-brR -> -LRdr for the .fmd),
    # ropebwt2's default -m10g (two batches).  Pins the RCLO order against the real reference above 1 M reads.
    g = subprocess.Popen([GEN, "100000000", "101", "42"], stdout=subprocess.PIPE)
    p = subprocess.Popen([REF, "-LRdr", "-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h, n = hashlib.md5(), 0
    for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(chunk); n += len(chunk)
    assert p.wait() == 0 and g.wait() == 0
    out["configs2_order_100M"] = {"n_reads": 100000000, "read_len": 101, "seed": 42, "flags": "-LRdr", "fmd_bytes": n, "fmd_md5": h.hexdigest(),
                                  "provenance": "oracle/_ref/ropebwt2 (the real reference) in the build container: synth_reads 100000000 101 42 | ropebwt2 -LRdr - ; make_golden_large.py --configs2-order"}
    json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
    sys.exit(0)
if "--configs1" in sys.argv:
    # BASELINE.json configs[1] at full size: 100 M x 101 bp, RLO, -m4g (6.5 minutes of reference time, 6.0 GB of .fmd):
    #   synth_reads 100000000 101 42 | ropebwt2 -LRds -m4g -   (run on the GPU box's host, where the test runs)
    g = subprocess.Popen([GEN, "100000000", "101", "42"], stdout=subprocess.PIPE)
    p = subprocess.Popen([REF, "-LRds", "-m4g", "-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h, n = hashlib.md5(), 0
    for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(chunk); n += len(chunk)
    assert p.wait() == 0 and g.wait() == 0
    out["configs1"] = {"n_reads": 100000000, "read_len": 101, "seed": 42, "flags": "-LRds -m4g", "fmd_bytes": n, "fmd_md5": h.hexdigest()}
    json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
    sys.exit(0)
for flags in ("-LRds", "-LRd"):
    g = subprocess.Popen([GEN, str(N), str(L), str(SEED)], stdout=subprocess.PIPE)
    p = subprocess.Popen([REF, flags, "-m1g", "-"], stdin=g.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h = hashlib.md5()
    for chunk in iter(lambda: p.stdout.read(1 << 24), b""):
        h.update(chunk)
    assert p.wait() == 0 and g.wait() == 0
    out["fmd_md5"][flags] = h.hexdigest()
    print(flags, h.hexdigest(), file=sys.stderr)
json.dump(out, open(os.path.join(HERE, "golden_large.json"), "w"), indent=1)
