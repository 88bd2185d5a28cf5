#!/usr/bin/env python3
"""Mutation fuzz of the threaded input readers (CPU only, no GPU): -L / FASTA / FASTQ inputs with random insertions, deletions and replacements from
the characters that matter to kseq ("@", ">", "+", CR, LF, blanks), random block sizes, stdin or a named file -- the batch stream (RB2_DUMP_BATCHES) of
the threaded readers against the sequential kseq-exact reader.  usage: fuzz_readers.py <seconds> [seed]     (8060 cases in 240 s, all equal, round 3)"""
import os, sys, subprocess, tempfile, time, numpy as np
import os as _o; _r = _o.environ.get("GRAFT_REPO_ROOT") or "/root/repo"; sys.path.insert(0, _r); sys.path.insert(0, _r + "/tests")
from test_host_layer import CLI, _fastq_inputs, _fasta_inputs
d = tempfile.mkdtemp()
def dump(flags, data, env, named):
    f = os.path.join(d, "b.bin")
    if os.path.exists(f): os.unlink(f)
    e = dict(os.environ, RB2_DUMP_BATCHES=f, RB2_NO_RESERVE="1"); e.update(env)
    if named:
        pth = os.path.join(d, "in.txt"); open(pth, "wb").write(data)
        p = subprocess.run([CLI] + flags + [pth], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    else:
        p = subprocess.run([CLI] + flags + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, (open(f, "rb").read() if os.path.exists(f) else b"")
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fq, fa = _fastq_inputs(), _fasta_inputs()
bases = [fq["strict"][:20000], fa["wrapped60"][:20000], fa["one_line"][:20000], b"\n".join(l for l in fa["one_line"][:20000].split(b"\n") if not l.startswith(b">"))]
alphabet = b"@>+\n\r\nACGTNacgt \t\n\n>@"
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    data = bytearray(bases[rng.randint(len(bases))])
    for _ in range(int(rng.randint(0, 12))):
        pos = int(rng.randint(0, len(data) + 1)); op = rng.randint(3)
        if op == 0: data[pos:pos] = bytes(alphabet[i] for i in rng.randint(0, len(alphabet), size=int(rng.randint(1, 5))))
        elif op == 1: del data[pos:pos + int(rng.randint(1, 200))]
        elif pos < len(data): data[pos] = alphabet[rng.randint(len(alphabet))]
    if rng.rand() < 0.2: data = data[:int(rng.randint(0, len(data) + 1))]
    data = bytes(data)
    line = rng.rand() < 0.3
    flags = (["-L"] if line else []) + [["-R"], [], ["-N"], ["-C"], ["-x", "4"]][rng.randint(5)] + ["-m%dk" % int(rng.choice([5, 40, 100000]))]
    named = bool(rng.rand() < 0.5)
    rc0, want = dump(flags, data, {"RB2_PARSE_THREADS": "1"}, False)
    env = {"RB2_PARSE_THREADS": str(int(rng.randint(2, 6))), "RB2_PARSE_CHUNK": str(int(rng.choice([64, 300, 4096, 70000])))}
    rc1, got = dump(flags, data, env, named)
    if rc0 != rc1 or got != want:
        open("/tmp/fuzz_reader_fail.bin", "wb").write(data)
        print("MISMATCH", flags, env, named, rc0, rc1, len(want), len(got)); sys.exit(1)
    n += 1
print("reader fuzz ok:", n, "cases")
