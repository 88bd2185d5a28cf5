#!/usr/bin/env python3
"""Generate tests/golden/golden.json with the REAL reference (oracle/_ref/ropebwt2, built from
/root/reference by oracle/Makefile).  Only md5 digests and tiny literal outputs are stored --
never reference source.  Run in the build container:  python tests/golden/make_golden.py
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from helpers import *  # noqa

KAT = b"ACG\nTTA\nACG\nGNA\n\nC\n"
out = {"generator": "tests/golden/make_golden.py", "reference": "lh3/ropebwt2 r187 (oracle/_ref)", "kat_input": KAT.decode(), "kat": {}, "sets": {}}
for fl in ("-LR", "-LRs", "-LRr", "-L", "-Ls", "-Lr", "-LRN"):
    out["kat"][fl] = run_ref([fl], KAT).decode().strip()
out["kat"]["-LRT"] = run_ref(["-LRT"], KAT).decode().strip()
out["kat_fmd_hex"] = run_ref(["-LRd"], KAT).hex()
out["kat_fmr_hex"] = run_ref(["-LRb"], KAT).hex()

SETS = [("10k_x_101", 10000, 101, 42), ("100k_x_101", 100000, 101, 42), ("1M_x_101", 1000000, 101, 42),
        ("200_x_10k", 200, 10000, 42), ("3k_x_300_s44", 3000, 300, 44)]
for name, n, L, seed in SETS:
    text = reads_to_text(splitmix_bases(n, L, seed))
    rec = {"n_reads": n, "read_len": L, "seed": seed, "input_md5": md5(text), "fmd_md5": {}, "text_md5": {}}
    for fl in ("-LR", "-LRs", "-LRr", "-Lr", "-L", "-Ls"):      # forward strand in input order / RLO / RCLO; both strands in RCLO, input order, RLO (main.c:227-236)
        rec["fmd_md5"][fl + "d"] = md5(run_ref([fl + "d"], text))
        rec["text_md5"][fl] = md5(run_ref([fl], text))
    out["sets"][name] = rec
    print(name, "done", file=sys.stderr)
# repetitive / variable-length / N-containing class (SURVEY.md section 7 fixture class ii)
for seed in (1, 2):
    reads = repetitive_reads(1500, seed=seed)
    text = lines_from_codes(reads)
    rec = {"kind": "repetitive_reads(1500, seed=%d)" % seed, "input_md5": md5(text), "fmd_md5": {}, "text": {}}
    for fl in ("-LR", "-LRs", "-LRr", "-Lr"):
        rec["fmd_md5"][fl + "d"] = md5(run_ref([fl + "d"], text))
        rec["text"][fl] = md5(run_ref([fl], text))
    out["sets"]["rep1500_seed%d" % seed] = rec
json.dump(out, open(os.path.join(GOLDEN_DIR, "golden.json"), "w"), indent=1)
print("wrote golden.json")
