"""GPU tests at the sizes BASELINE.json's configs name (or the largest slice of them one GPU and a few minutes allow):

  configs[2]  1.2 B x 101 bp, -brR, 8 GPUs     -> the sharded build (8 virtual ranks behind one handle) on one FULL -m4g batch + a second
                                                  batch (60.8 M reads, RCLO, forward strand) against the single-GPU engine,
                                                  and the 100 M-read golden .fmd md5 (real reference) through the sharded path
  configs[3]  10 M x 10 kbp, input order       -> 1 M x 10 kbp (one -m10g batch of 10,001 rounds): size-independent properties
  configs[4]  -bi old.fmr + new reads          -> 5 M (reference-built .fmr) + 5 M through the CLI against the 10 M golden
                                                  and 50 M + 50 M (our .fmr, decoded on the device) against the 100 M golden

Everything is bit-exact: ropes byte for byte, .fmd by md5 of the reference's own output (tests/golden/golden_large.json)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from test_host_layer import CLI

pytestmark = pytest.mark.gpu


def batch_reads(mem_gib, read_len):
    m = int(mem_gib * 1024 ** 3 * 0.97) + 1                  # main.c:136
    return -(-m // (read_len + 1))                           # main.c:238


def test_configs2_shape_sharded_full_batch_vs_single_gpu(hip):
    """8 virtual ranks (the owner map an 8-GPU run uses), RCLO, forward strand: one full -m4g batch (40.8 M reads) and a
    second batch of 20 M on top.  Count matrix after every batch and all six ropes, run byte for run byte, equal to the
    single-GPU engine's (which the other tests pin to the oracle and the reference)."""
    L = 101
    plan = [(0, batch_reads(4, L)), (batch_reads(4, L), 20_000_000)]
    one = hip.HipBwt(2)
    vc = hip.MultiBwt(2, [0] * 8, "peer")
    p = one.dev_alloc(plan[0][1] * (L + 1))
    for first, n in plan:
        one.synth_reads(p, first, n, L, seed=42)
        one.sync()
        one.insert_multi_dev(p, n * (L + 1))
        vc.insert_multi_dev(p, n * (L + 1))
        assert np.array_equal(one.counts(), vc.counts())
    one.dev_free(p)
    tot = sum(n for _, n in plan)
    assert int(one.counts().sum()) == tot * (L + 1) and tot >= 60_000_000
    for b in range(6):
        a, v = one.rope_rle(b), vc.rope_rle(b)
        assert len(a) == len(v) and np.array_equal(a, v), "rope %d differs between 8 virtual ranks and one GPU" % b
    vc.close()
    one.close()


def _fmd_md5_of(run_streams):
    """feed six 43+3 run streams to the host .fmd writer (the CLI's own encoder) and return (bytes, md5) of the file"""
    from ropebwt2_amd.build import lib_path
    Lh = C.CDLL(lib_path("libropebwt2.so"))
    Lh.rb2_fmd_init.restype = C.c_void_p
    Lh.rb2_fmd_push_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    Lh.rb2_fmd_finish.argtypes = [C.c_void_p]
    Lh.rb2_fmd_destroy.argtypes = [C.c_void_p]
    Lh.rb2_fmd_write_path.argtypes = [C.c_void_p, C.c_char_p]
    Lh.rb2_fmd_write_path.restype = C.c_int
    f = Lh.rb2_fmd_init()
    for get in run_streams:
        r = np.ascontiguousarray(get())
        Lh.rb2_fmd_push_runs(f, r.ctypes.data, len(r))
        del r
    Lh.rb2_fmd_finish(f)
    path = "/dev/shm/rb2_test_%d.fmd" % os.getpid()
    try:
        assert Lh.rb2_fmd_write_path(f, path.encode()) == 0
        Lh.rb2_fmd_destroy(f)
        h, n = hashlib.md5(), 0
        with open(path, "rb") as fp:
            for chunk in iter(lambda: fp.read(1 << 24), b""):
                h.update(chunk); n += len(chunk)
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return n, h.hexdigest()


def test_configs1_golden_through_sharded_path(hip):
    """the full configs[1] job (100 M x 101 bp, RLO, three -m4g batches) built by 8 virtual ranks; the ropes are gathered
    from their owners and encoded by the host writer: the 6.0 GB .fmd has the md5 of the real reference's output"""
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))["configs1"]
    L, per = g["read_len"], batch_reads(4, g["read_len"])
    vc = hip.MultiBwt(1, [0] * 8, "peer")
    r0 = vc.engine(0)
    p = r0.dev_alloc(per * (L + 1))
    done = 0
    while done < g["n_reads"]:
        n = min(per, g["n_reads"] - done)
        r0.synth_reads(p, done, n, L, seed=g["seed"])
        r0.sync()
        vc.insert_multi_dev(p, n * (L + 1))
        done += n
    r0.dev_free(p)
    assert int(vc.counts().sum()) == g["n_reads"] * (L + 1)
    nbytes, md5 = _fmd_md5_of([lambda b=b: vc.rope_rle(b) for b in range(6)])
    vc.close()
    assert nbytes == g["fmd_bytes"] and md5 == g["fmd_md5"]


def _sym_at(dev, b, p):
    return int(np.argmax(dev.rank1a(b, p + 1) - dev.rank1a(b, p)))


def test_configs3_long_reads_1M_x_10k(hip):
    """configs[3] at one tenth: 1 M x 10 kbp in input order, one batch of 10,001 rounds (10.0 G symbols).
    (1) count matrix: LF consistency; (2) rope $ in input order IS the sequence of last bases of the reads;
    (3) inverse-BWT walks from sampled rows reproduce sampled reads exactly (10 k LF steps each)."""
    n, L = 1_000_000, 10_000
    dev = hip.HipBwt(0)
    p = dev.dev_alloc(n * (L + 1))
    dev.synth_reads(p, 0, n, L, seed=44)
    dev.insert_multi_dev(p, n * (L + 1))
    dev.dev_free(p)
    c = dev.counts()
    assert c.sum() == n * (L + 1) and c[:, 0].sum() == n and c[0].sum() == n
    for b in range(1, 6):
        assert c[b].sum() == c[:, b].sum()
    last = np.array([H.splitmix_bases(1, L, seed=44, first=k)[0][-1] for k in (0, 1, 2, 499_999, n - 1)])
    rope0 = dev.rope(0)
    assert len(rope0) == n and np.array_equal(rope0[[0, 1, 2, 499_999, n - 1]], last)
    for k in (0, 777_777, n - 1):
        want = H.splitmix_bases(1, L, seed=44, first=k)[0][::-1]
        b, row, got = 0, k, []
        while True:
            s = _sym_at(dev, b, row)
            if s == 0:
                break
            got.append(s)
            row = int(c[:b, s].sum() + dev.rank1a(b, row)[s])
            b = s
        assert np.array_equal(np.array(got, np.uint8), want), "read %d" % k
    dev.close()


def test_configs4_incremental_5M_plus_5M(tmp_path):
    """configs[4] shape: an existing .fmr of 5 M reads -- built by the REAL reference when oracle/_ref is present, else by
    our CLI -- plus 5 M new reads through `-i`; the .fmd equals the reference's one-shot build of all 10 M reads"""
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))
    half, L, seed = g["n_reads"] // 2, g["read_len"], g["seed"]
    old = tmp_path / "old.fmr"
    builder = H.REF_BIN if H.have_ref() else CLI
    pg = subprocess.Popen([H.GEN, str(half), str(L), str(seed)], stdout=subprocess.PIPE)
    pb = subprocess.run([builder, "-LRbs", "-m1g", "-o", str(old), "-"], stdin=pg.stdout, stderr=subprocess.DEVNULL)
    assert pb.returncode == 0 and pg.wait() == 0
    pg = subprocess.Popen([H.GEN, str(half), str(L), str(seed), str(half)], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI, "-LRd", "-m400m", "-i", str(old), "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h = hashlib.md5()
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert h.hexdigest() == g["fmd_md5"]["-LRds"]
    # and the other direction: our .fmr of the first half continues in the reference
    if H.have_ref():
        ours = tmp_path / "ours.fmr"
        pg = subprocess.Popen([H.GEN, str(half), str(L), str(seed)], stdout=subprocess.PIPE)
        assert subprocess.run([CLI, "-LRbs", "-m1g", "-o", str(ours), "-"], stdin=pg.stdout, stderr=subprocess.DEVNULL).returncode == 0
        pg.wait()
        n1 = 200_000                                        # the reference inserts on the CPU: a small second half keeps this quick
        pg = subprocess.Popen([H.GEN, str(n1), str(L), str(seed), str(half)], stdout=subprocess.PIPE)
        ref = subprocess.run([H.REF_BIN, "-LRd", "-i", str(ours), "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        pg.wait()
        pg = subprocess.Popen([H.GEN, str(n1), str(L), str(seed), str(half)], stdout=subprocess.PIPE)
        our = subprocess.run([CLI, "-LRd", "-i", str(ours), "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        pg.wait()
        assert ref.returncode == 0 and our.returncode == 0 and H.md5(ref.stdout) == H.md5(our.stdout)


def test_configs4_incremental_50M_plus_50M(tmp_path):
    """configs[4] at one tenth: an existing .fmr of 50 M reads (5.1 G symbols, 3.8 GB of run-length leaves, written by `-b`)
    restored with `-i`, decoded into packed leaves ON THE DEVICE (k_ld_*), 50 M more reads inserted in two -m4g batches; in
    RLO the incremental build equals the one-shot build, so the .fmd must have the md5 the real reference
    produced for configs[1]."""
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))["configs1"]
    half, L, seed = g["n_reads"] // 2, g["read_len"], g["seed"]
    work = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path)
    old = os.path.join(work, "rb2_test_old_%d.fmr" % os.getpid())
    try:
        pg = subprocess.Popen([H.GEN, str(half), str(L), str(seed)], stdout=subprocess.PIPE)
        pb = subprocess.run([CLI, "-LRbs", "-m4g", "-o", old, "-"], stdin=pg.stdout, stderr=subprocess.DEVNULL)
        assert pb.returncode == 0 and pg.wait() == 0
        pg = subprocess.Popen([H.GEN, str(half), str(L), str(seed), str(half)], stdout=subprocess.PIPE)
        pc = subprocess.Popen([CLI, "-LRds", "-m4g", "-i", old, "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        h, n = hashlib.md5(), 0
        for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
            h.update(chunk); n += len(chunk)
        assert pc.wait() == 0 and pg.wait() == 0
        assert n == g["fmd_bytes"] and h.hexdigest() == g["fmd_md5"]
    finally:
        if os.path.exists(old):
            os.remove(old)
