"""GPU tests of the drop-in boundary: the `ropebwt2` CLI and the mrope C API (libropebwt2.so) with
the batch insertion running in the HIP engine.  Parity target: the reference's .fmd bytes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from test_host_layer import CLI, cli, make_fastx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _build(hip):
    H.build_oracle()


@pytest.mark.parametrize("flag", ["-LR", "-LRs", "-LRr", "-L", "-Ls", "-Lr", "-LRN", "-LRT"])
def test_kat_cli_gpu(golden, flag):
    assert cli([flag], golden["kat_input"].encode()).decode().strip() == golden["kat"][flag]


def test_kat_fmd_bytes_gpu(golden):
    assert cli(["-LRd"], golden["kat_input"].encode()).hex() == golden["kat_fmd_hex"]


@pytest.mark.parametrize("name", ["10k_x_101", "100k_x_101", "200_x_10k", "3k_x_300_s44"])
@pytest.mark.parametrize("flag", ["-LRd", "-LRsd", "-LRrd", "-Lrd", "-Ld", "-Lsd"])
def test_fmd_golden(golden, name, flag):
    g = golden["sets"][name]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    assert H.md5(cli([flag], text)) == g["fmd_md5"][flag]


@pytest.mark.parametrize("flag", ["-LRsd", "-Lrd", "-Ld", "-Lsd"])
def test_fmd_golden_1M_small_batches(golden, flag):
    """1 M reads, -m20m -> 5..10 GPU batches; .fmd is independent of the batching"""
    g = golden["sets"]["1M_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    assert H.md5(cli([flag, "-m20m"], text)) == g["fmd_md5"][flag]


@pytest.mark.parametrize("flags", ["-LRds", "-LRd"])
def test_fmd_golden_10M_streamed(flags):
    """10 M x 101 bp (1.02 G symbols) streamed from the generator through the CLI in ~400 MB batches: the .fmd md5
    equals the one the real reference produced (tests/golden/golden_large.json, make_golden_large.py)"""
    import hashlib, json
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))
    gen = H.GEN
    assert os.path.exists(gen), "ropebwt2_amd/bin/synth_reads not built"
    pg = subprocess.Popen([gen, str(g["n_reads"]), str(g["read_len"]), str(g["seed"])], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI, flags, "-m400m", "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h = hashlib.md5()
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert h.hexdigest() == g["fmd_md5"][flags]


@pytest.mark.parametrize("flags", ["-LRds -m1g", "-Lrd -m2g"])
def test_fmd_golden_coverage_reads(flags):
    """30 M overlapping reads (windows of one random genome, 30x; 3-6 G symbols): non-empty intervals, large groups and a
    compressible BWT in every round -- the regime of real data.  .fmd md5 as produced by the real reference."""
    import hashlib, json
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))["coverage30x"]
    gen = H.GEN
    assert os.path.exists(gen), "ropebwt2_amd/bin/synth_reads not built"
    pg = subprocess.Popen([gen, str(g["n_reads"]), str(g["read_len"]), str(g["seed"]), "0", str(g["genome_len"])], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI] + flags.split() + ["-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h, n = hashlib.md5(), 0
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk); n += len(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert n == g["runs"][flags]["fmd_bytes"] and h.hexdigest() == g["runs"][flags]["fmd_md5"]


def test_fmd_golden_long_reads():
    """200 k x 5 kbp in input order (1 G symbols, one batch of 5001 rounds): the long-string path against the real reference"""
    import hashlib, json
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))["longreads"]
    gen = H.GEN
    pg = subprocess.Popen([gen, str(g["n_reads"]), str(g["read_len"]), str(g["seed"])], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI, g["flags"], "-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h = hashlib.md5()
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert h.hexdigest() == g["fmd_md5"]


def test_fmd_golden_configs1_full_size():
    """BASELINE.json configs[1] at full size -- 100 M x 101 bp, RLO, -m4g (three GPU batches, 10.2 G symbols) -- streamed
    through the CLI: the 6.0 GB .fmd has the md5 the real reference produced for the same input on the same kind of box"""
    import hashlib, json
    g = json.load(open(os.path.join(H.GOLDEN_DIR, "golden_large.json")))["configs1"]
    gen = H.GEN
    assert os.path.exists(gen), "ropebwt2_amd/bin/synth_reads not built"
    pg = subprocess.Popen([gen, str(g["n_reads"]), str(g["read_len"]), str(g["seed"])], stdout=subprocess.PIPE)
    pc = subprocess.Popen([CLI] + g["flags"].split() + ["-"], stdin=pg.stdout, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    h, n = hashlib.md5(), 0
    for chunk in iter(lambda: pc.stdout.read(1 << 24), b""):
        h.update(chunk); n += len(chunk)
    assert pc.wait() == 0 and pg.wait() == 0
    assert n == g["fmd_bytes"] and h.hexdigest() == g["fmd_md5"]


@pytest.mark.parametrize("so_flag", ["", "s", "r"])
def test_incremental_build(golden, so_flag, tmp_path):
    """config 5 shape: -b on the first half, then -i + second half == one-shot build"""
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    half = tmp_path / "half.fmr"
    half.write_bytes(cli(["-LRb" + so_flag], H.reads_to_text(codes[:5000])))
    out = cli(["-LRd", "-i", str(half)], H.reads_to_text(codes[5000:]))
    assert H.md5(out) == g["fmd_md5"]["-LR" + so_flag + "d"]


@pytest.mark.parametrize("so_flag,extra", [("", []), ("s", ["-l", "64", "-n", "6"]), ("r", ["-m", "300k"])])
def test_fmr_straight_from_the_device_equals_the_dump_of_the_host_trees(golden, so_flag, extra, tmp_path):
    """-b -o FILE on a device-built index: run bytes off the device -> leaf records in the file, no host B+ trees
    (dump_without_trees, mrope.c); the same bytes as device -> trees -> rope_dump (RB2_DUMP_VIA_TREES=1 and the piped path),
    and -i of it + nothing reproduces the .fmd golden"""
    g = golden["sets"]["100k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    piped = cli(["-LRb" + so_flag] + extra, text)
    outs = []
    for env in ({}, {"RB2_DUMP_VIA_TREES": "1"}, {"RB2_LOAD_THREADS": "3"}):
        f = tmp_path / ("o%d.fmr" % len(outs))
        p = subprocess.run([CLI, "-LRb" + so_flag] + extra + ["-o", str(f), "-"], input=text, stderr=subprocess.PIPE, env=dict(os.environ, RB2_SYNC_TRACE="1", **env))
        assert p.returncode == 0, p.stderr.decode()[-400:]
        assert (b"[mr_dump] rope" in p.stderr) == ("RB2_DUMP_VIA_TREES" not in env)
        outs.append(f.read_bytes())
    assert outs[0] == outs[1] == outs[2] == piped
    out = subprocess.run([CLI, "-d", "-m0", "-i", str(tmp_path / "o0.fmr"), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert H.md5(out) == g["fmd_md5"]["-LR" + so_flag + "d"]


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_our_fmr_restores_in_reference(golden, tmp_path):
    g = golden["sets"]["100k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    f = tmp_path / "o.fmr"
    f.write_bytes(cli(["-LRbr"], text))
    ref = subprocess.run([H.REF_BIN, "-d", "-i", str(f), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert H.md5(ref) == g["fmd_md5"]["-LRrd"]


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("flags", [[], ["-N"], ["-q", "20"], ["-x", "8"], ["-x", "8", "-C"], ["-C", "-s"], ["-F"], ["-R", "-r"], ["-x", "3", "-r"]])
def test_filters_match_reference(flags):
    fq, fa = make_fastx()
    for data in (fq, fa):
        if data is fa and "-q" in flags:
            continue
        assert cli(flags + ["-d"], data) == H.run_ref(flags + ["-d"], data)


def test_mrope_c_api(golden):
    """drive libropebwt2.so the way main.c drives the reference: mr_init / mr_insert_multi /
    inline count helpers / mr_rank2a / mr_insert1 after a GPU batch / iterator"""
    from ropebwt2_amd.build import lib_path
    L = C.CDLL(lib_path("libropebwt2.so"))
    L.mr_init.restype = C.c_void_p
    L.mr_init.argtypes = [C.c_int, C.c_int, C.c_int]
    L.mr_insert_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    L.mr_insert1.argtypes = [C.c_void_p, C.c_void_p]
    L.mr_insert1.restype = C.c_int64
    L.mr_rank2a.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.mr_destroy.argtypes = [C.c_void_p]
    L.mr_sync_host.argtypes = [C.c_void_p]

    class MRope(C.Structure):
        _fields_ = [("so", C.c_uint8), ("thr_min", C.c_int), ("r", C.c_void_p * 6)]

    class Rope(C.Structure):
        _fields_ = [("max_nodes", C.c_int32), ("block_len", C.c_int32), ("c", C.c_int64 * 6)]

    reads = H.repetitive_reads(3000, seed=77, genome_len=500, max_len=70)
    for so in (0, 1, 2):
        o = H.Oracle(so)
        mr = L.mr_init(64, 512, so)
        for part in (reads[:1200], reads[1200:2500]):
            buf = H.encode_batch(part)
            o.insert_multi(buf)
            L.mr_insert_multi(mr, len(buf), buf.ctypes.data, 1)
        m = MRope.from_address(mr)
        counts = np.array([[Rope.from_address(m.r[a]).c[b] for b in range(6)] for a in range(6)])
        assert np.array_equal(counts, o.counts())            # rope_t.c current right after the GPU call
        bwt0 = o.bwt()                                       # mr_rank2a while the BWT only lives in HBM: answered by the device
        rng0 = np.random.RandomState(100 + so)
        for _ in range(12):
            x = int(rng0.randint(0, len(bwt0) + 1)); y = int(rng0.randint(x, len(bwt0) + 1))
            cx = (C.c_int64 * 6)(); cy = (C.c_int64 * 6)()
            L.mr_rank2a(mr, x, y, cx, cy)
            assert list(cx) == np.bincount(bwt0[:x], minlength=6).tolist()
            assert list(cy) == np.bincount(bwt0[:y], minlength=6).tolist()
        for r in reads[2500:]:                               # -m0 style inserts on top of the GPU-built index
            s = np.ascontiguousarray(np.concatenate([np.asarray(r, np.uint8)[::-1], np.zeros(1, np.uint8)]))
            L.mr_insert1(mr, s.ctypes.data)
            o.insert1(np.asarray(r, np.uint8)[::-1])
        bwt = o.bwt()
        rng = np.random.RandomState(so)
        for _ in range(20):
            x = int(rng.randint(0, len(bwt) + 1)); y = int(rng.randint(x, len(bwt) + 1))
            cx = (C.c_int64 * 6)(); cy = (C.c_int64 * 6)()
            L.mr_rank2a(mr, x, y, cx, cy)
            assert list(cx) == np.bincount(bwt[:x], minlength=6).tolist()
            assert list(cy) == np.bincount(bwt[:y], minlength=6).tolist()
        buf = H.encode_batch(reads[:300], True, True)        # and back to the GPU: host ropes are re-uploaded
        o.insert_multi(buf)
        L.mr_insert_multi(mr, len(buf), buf.ctypes.data, 1)
        m = MRope.from_address(mr)
        counts = np.array([[Rope.from_address(m.r[a]).c[b] for b in range(6)] for a in range(6)])
        assert np.array_equal(counts, o.counts())
        # mr_stream_runs: the BWT as run bytes straight from the device (no host trees), then again from the host leaves
        CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint8), C.c_int64)
        L.mr_stream_runs.argtypes = [C.c_void_p, CB, C.c_void_p]
        for sync_first in (False, True):
            chunks = []
            if sync_first:
                L.mr_sync_host(mr)
            L.mr_stream_runs(mr, CB(lambda u, q, n: chunks.append(bytes(q[:n]))), None)
            syms = H.decode_runs(np.frombuffer(b"".join(chunks), np.uint8))
            assert np.array_equal(syms, o.bwt()), "mr_stream_runs (host trees: %s)" % sync_first
        L.mr_destroy(mr)


def test_block_iterator_of_a_device_index_builds_no_host_trees():
    """mr_itr_first / mr_itr_next_block (mrope.c:111-130; what the reference's main.c:288-305 walks for -d) on an index that
    lives in HBM: the run bytes come off the device rope by rope and are cut into the leaves of the bulk-loaded tree without
    building it -- mr_host_resident() stays 0, the blocks equal those of the tree walk (RB2_ITR_VIA_TREES=1) and decode to
    the oracle's BWT; the index keeps taking batches afterwards"""
    from ropebwt2_amd.build import lib_path
    L = C.CDLL(lib_path("libropebwt2.so"))
    L.mr_init.restype = C.c_void_p; L.mr_init.argtypes = [C.c_int, C.c_int, C.c_int]
    L.mr_insert_multi.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
    L.mr_itr_first.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.mr_itr_next_block.restype = C.c_void_p; L.mr_itr_next_block.argtypes = [C.c_void_p]
    L.mr_host_resident.argtypes = [C.c_void_p]
    L.mr_destroy.argtypes = [C.c_void_p]

    class MrItr(C.Structure):
        _fields_ = [("r", C.c_void_p), ("a", C.c_int), ("to_free", C.c_int),
                    ("rope", C.c_void_p), ("pa", C.c_void_p * 80), ("ia", C.c_int * 80), ("d", C.c_int)]

    def walk(mr, to_free=0):
        it = MrItr()
        L.mr_itr_first(mr, C.byref(it), to_free)
        out = []
        while True:
            q = L.mr_itr_next_block(C.byref(it))
            if not q:
                return out
            out.append(C.string_at(q + 2, C.cast(q, C.POINTER(C.c_uint16))[0]))

    reads = H.repetitive_reads(6000, seed=91, genome_len=900, max_len=80)
    for so, block_len in ((0, 512), (1, 64), (2, 512)):
        o = H.Oracle(so)
        mr = L.mr_init(64 if block_len == 512 else 6, block_len, so)
        buf = H.encode_batch(reads[:4000])
        o.insert_multi(buf); L.mr_insert_multi(mr, len(buf), buf.ctypes.data, 1)
        blocks = walk(mr)
        assert L.mr_host_resident(mr) == 0
        assert max(len(b) for b in blocks) <= block_len - 2
        assert np.array_equal(H.decode_runs(np.frombuffer(b"".join(blocks), np.uint8)), o.bwt())
        os.environ["RB2_ITR_VIA_TREES"] = "1"
        try:
            assert walk(mr) == blocks and L.mr_host_resident(mr) == 1
        finally:
            os.environ.pop("RB2_ITR_VIA_TREES", None)
        buf = H.encode_batch(reads[4000:])                   # host trees exist now; the next batch makes the device the owner again
        o.insert_multi(buf); L.mr_insert_multi(mr, len(buf), buf.ctypes.data, 1)
        assert L.mr_host_resident(mr) == 0
        blocks = walk(mr, to_free=1)                         # main.c:288: the freeing walk
        assert np.array_equal(H.decode_runs(np.frombuffer(b"".join(blocks), np.uint8)), o.bwt())
        L.mr_destroy(mr)


DROPIN = os.path.join(H.ORACLE_DIR, "_ref", "ropebwt2_dropin")


@pytest.mark.skipif(not os.path.exists(DROPIN), reason="oracle/_ref/ropebwt2_dropin not built (needs /root/reference at build time)")
@pytest.mark.parametrize("flag", ["-LRd", "-LRsd", "-LRrd", "-Lrd"])
def test_reference_main_linked_against_our_library(golden, flag):
    """The reference's own main.c/rld0.c objects (compiled with the reference's headers) linked
    against libropebwt2.so: mr_insert_multi runs in the HIP engine, the reference's code iterates
    our leaves with its rle_dec1 macro and writes the .fmd -- bytes must equal the goldens."""
    g = golden["sets"]["100k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    p = subprocess.run([DROPIN, flag, "-m3m", "-"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    assert H.md5(p.stdout) == g["fmd_md5"][flag]


@pytest.mark.parametrize("opts", [["-l", "64", "-n", "8"], ["-l", "2048", "-n", "128"], ["-P", "-M", "5"], ["-m", "300k"], ["-m", "0.002g"]])
def test_layout_and_batch_options_do_not_change_the_bwt(golden, opts):
    """-l/-n only shape the host tree, -P/-M are accepted, -m only cuts batches: same .fmd"""
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    assert H.md5(cli(["-LRds"] + opts, text)) == g["fmd_md5"]["-LRsd"]


def test_misaligned_device_buffer(hip):
    """insert_multi_dev with a device pointer that is not 16-byte aligned (kernels use 16-byte loads)"""
    codes = H.splitmix_bases(5000, 33, seed=12)
    buf = H.encode_batch_fixed(codes)
    a = hip.HipBwt(1)
    a.insert_multi(buf)
    b = hip.HipBwt(1)
    p = b.dev_alloc(len(buf) + 64)
    b.L.rb2_hip_memcpy(b.h, p + 5, buf.ctypes.data, len(buf), 0)
    b.insert_multi_dev(p + 5, len(buf))
    b.dev_free(p)
    for r in range(6):
        assert np.array_equal(a.rope_rle(r), b.rope_rle(r))


def test_restore_rejects_garbage(tmp_path):
    f = tmp_path / "bad.fmr"
    f.write_bytes(b"not an fmr file at all")
    p = subprocess.run([CLI, "-d", "-i", str(f), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"not an FMR file" in p.stderr
