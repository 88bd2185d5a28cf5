#!/usr/bin/env python3
"""Randomised comparison of the host-side writers (CPU only): rope_rdump_* (the .fmr without trees) against rope_load_runs + rope_dump for random
max_nodes / block_len / thread counts, and the parallel / streamed .fmd writer against the sequential one for random chunkings, segment sizes, write steps --
on run streams with merging neighbours, EMPTY runs and 2/4/8-byte runs.  (Found in round 3: the parallel .fmd writer let an empty run separate two runs of one symbol.)"""
import subprocess, sys
PARTS = r"""
import ctypes as C, os, sys, numpy as np, tempfile
import os as _o; _r = _o.environ.get("GRAFT_REPO_ROOT") or "/root/repo"; sys.path.insert(0, _r); sys.path.insert(0, _r + "/tests")
from ropebwt2_amd.build import lib_path
from test_host_layer import _mixed_run_stream
L = C.CDLL(lib_path("libropebwt2.so"))
libc = C.CDLL(None)
libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
L.rope_init.restype = C.c_void_p; L.rope_init.argtypes = [C.c_int, C.c_int]
L.rope_load_runs.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]; L.rope_load_runs_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int]
L.rope_dump.argtypes = [C.c_void_p, C.c_void_p]; L.rope_destroy.argtypes = [C.c_void_p]
L.rope_rdump_prepare.restype = C.c_void_p; L.rope_rdump_prepare.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int]
L.rope_rdump_size.restype = C.c_int64; L.rope_rdump_size.argtypes = [C.c_void_p]; L.rope_rdump_write.argtypes = [C.c_void_p, C.c_int, C.c_int64]
d = tempfile.mkdtemp()
rng = np.random.RandomState(5)
os.environ["RB2_LOAD_MIN_SEG"] = "2000"
for it in range(150):
    n = int(rng.choice([0, 1, 5, 200, 5000, 60000, 400000]))
    mn, bl, thr = int(rng.choice([4, 5, 6, 10, 64, 200])), int(rng.choice([30, 32, 64, 100, 512, 1024])), int(rng.randint(1, 17))
    stream = _mixed_run_stream(int(rng.randint(1, 1000)), n)
    r = L.rope_init(mn, bl); L.rope_load_runs(r, stream, len(stream))
    f = os.path.join(d, "t.bin"); fp = libc.fopen(f.encode(), b"wb"); L.rope_dump(r, fp); libc.fclose(fp); L.rope_destroy(r)
    want = open(f, "rb").read()
    h = L.rope_rdump_prepare(stream, len(stream), mn, bl, thr)
    assert L.rope_rdump_size(h) == len(want), (it, n, mn, bl, thr)
    g = os.path.join(d, "g.bin"); fd = os.open(g, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    assert L.rope_rdump_write(h, fd, 0) == 0; os.close(fd)
    assert open(g, "rb").read() == want, (it, n, mn, bl, thr)
print("rdump fuzz ok")
"""
PARTS2 = r"""
import ctypes as C, os, sys, numpy as np, tempfile
import os as _o; _r = _o.environ.get("GRAFT_REPO_ROOT") or "/root/repo"; sys.path.insert(0, _r); sys.path.insert(0, _r + "/tests")
from ropebwt2_amd.build import lib_path
from test_host_layer import _mixed_run_stream
L = C.CDLL(lib_path("libropebwt2.so"))
libc = C.CDLL(None)
libc.fdopen.restype = C.c_void_p; libc.fdopen.argtypes = [C.c_int, C.c_char_p]; libc.fclose.argtypes = [C.c_void_p]
L.rb2_fmd_init.restype = C.c_void_p; L.rb2_fmd_push_runs.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
L.rb2_fmd_finish.argtypes = [C.c_void_p]; L.rb2_fmd_destroy.argtypes = [C.c_void_p]; L.rb2_fmd_write_path.argtypes = [C.c_void_p, C.c_char_p]
L.rb2_fmdp_init.restype = C.c_void_p; L.rb2_fmdp_init.argtypes = [C.c_int, C.c_int64]
L.rb2_fmdp_push_runs.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]; L.rb2_fmdp_finish.restype = C.c_void_p; L.rb2_fmdp_finish.argtypes = [C.c_void_p]
L.rb2_fmdp_set_output.argtypes = [C.c_void_p, C.c_int, C.c_int64]; L.rb2_fmdp_expect.argtypes = [C.c_void_p, C.c_int64]
L.rb2_fmd_write.argtypes = [C.c_void_p, C.c_void_p]
d = tempfile.mkdtemp(); rng = np.random.RandomState(9)
for it in range(120):
    n = int(rng.choice([0, 1, 40, 3000, 80000, 700000]))
    stream = _mixed_run_stream(int(rng.randint(1, 1000)), n)
    # split at run heads
    arr = np.frombuffer(stream, np.uint8)
    cuts = [0]
    while cuts[-1] < len(arr):
        c = min(len(arr), cuts[-1] + int(rng.randint(1, 20000)))
        while c < len(arr) and (arr[c] & 0xC0) == 0x80: c += 1
        cuts.append(c)
    chunks = [stream[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    f = L.rb2_fmd_init()
    for c in chunks: L.rb2_fmd_push_runs(f, c, len(c))
    L.rb2_fmd_finish(f); a = os.path.join(d, "s.fmd"); assert L.rb2_fmd_write_path(f, a.encode()) == 0; L.rb2_fmd_destroy(f)
    want = open(a, "rb").read()
    os.environ["RB2_FMD_OUT_STEP"] = str(int(rng.choice([1, 7, 64, 5000])))
    g = os.path.join(d, "p.fmd"); fd = os.open(g, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644); fp = libc.fdopen(fd, b"wb")
    p = L.rb2_fmdp_init(int(rng.randint(1, 9)), int(rng.choice([256, 3000, 70000, 0])))
    if rng.rand() < 0.5: L.rb2_fmdp_expect(p, int(rng.choice([1, 1000, 10**7, 10**9])))
    if rng.rand() < 0.7: assert L.rb2_fmdp_set_output(p, fd, 0) == 0
    for c in chunks: L.rb2_fmdp_push_runs(p, c, len(c))
    f = L.rb2_fmdp_finish(p); assert L.rb2_fmd_write(f, fp) == 0; L.rb2_fmd_destroy(f); libc.fclose(fp)
    assert open(g, "rb").read() == want, (it, n)
print("fmd fuzz ok")
"""
for src in (PARTS, PARTS2):
    subprocess.run([sys.executable, "-c", src], check=True)
