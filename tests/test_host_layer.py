"""CPU tests of the plain-C host layer (include/mrope.h, rope.h, rle.h, rb2_fmd.h) and the CLI in
its CPU-only mode (-m0 = mr_insert1).  These pin the parity ARTEFACT writers: .fmd bytes must equal
the reference's, .fmr must restore (in the reference too) to the same BWT.  No GPU is touched."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers as H

ROOT = H.ROOT
CLI = os.path.join(ROOT, "ropebwt2_amd", "bin", "ropebwt2")
SO_FLAG = {0: "-LR", 1: "-LRs", 2: "-LRr"}


@pytest.fixture(scope="module", autouse=True)
def _build():
    from ropebwt2_amd import build_all
    build_all()
    H.build_oracle()


def cli(flags, data, check=True):
    p = subprocess.run([CLI] + list(flags) + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if check:
        assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.fixture(scope="module")
def hostlib():
    from ropebwt2_amd.build import lib_path
    L = C.CDLL(lib_path("libropebwt2.so"))
    L.rb2_fmd_init.restype = C.c_void_p
    L.rb2_fmd_push.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.rb2_fmd_finish.argtypes = [C.c_void_p]
    L.rb2_fmd_destroy.argtypes = [C.c_void_p]
    return L


def test_exports_reference_api(hostlib):
    names = ["mr_init", "mr_destroy", "mr_thr_min", "mr_insert1", "mr_insert_multi", "mr_rank2a", "mr_itr_first",
             "mr_itr_next_block", "mr_print_tree", "mr_dump", "mr_restore",
             "rope_init", "rope_destroy", "rope_insert_run", "rope_rank2a", "rope_itr_first", "rope_itr_next_block",
             "rope_print_node", "rope_dump", "rope_restore",
             "rle_insert_cached", "rle_insert", "rle_split", "rle_count", "rle_rank2a", "rle_print", "rle_auxtab",
             "rb2_fmd_init", "rb2_fmd_push", "rb2_fmd_finish", "rb2_fmd_write", "rb2_fmd_counts", "rb2_fmd_destroy"]
    for n in names:
        assert hasattr(hostlib, n), n


@pytest.mark.parametrize("flag", ["-LR", "-LRs", "-LRr", "-L", "-Ls", "-Lr", "-LRN", "-LRT"])
def test_kat_cli_m0(golden, flag):
    assert cli([flag, "-m0"], golden["kat_input"].encode()).decode().strip() == golden["kat"][flag]


def test_kat_fmd_and_fmr_bytes(golden):
    kat = golden["kat_input"].encode()
    assert cli(["-LRd", "-m0"], kat).hex() == golden["kat_fmd_hex"]
    fmr = cli(["-LRb", "-m0"], kat)
    assert fmr[:4] == b"RB\x02\x00"
    # same content as the reference's tiny .fmr (tree shape is identical for a one-leaf rope)
    assert fmr.hex() == golden["kat_fmr_hex"]


@pytest.mark.parametrize("flag", ["-LRd", "-LRsd", "-LRrd", "-Lrd"])
def test_fmd_golden_10k_m0(golden, flag):
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    assert H.md5(cli([flag, "-m0"], text)) == g["fmd_md5"][flag]


@pytest.mark.parametrize("so", [0, 1, 2])
def test_fmd_writer_on_oracle_bwt(hostlib, golden, so, tmp_path):
    """feed the oracle's BWT runs to our FMD writer: bytes equal the reference's .fmd"""
    g = golden["sets"]["10k_x_101"]
    codes = H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"])
    o = H.Oracle(so)
    o.insert_multi(H.encode_batch_fixed(codes))
    bwt = o.bwt()
    edges = np.flatnonzero(np.diff(bwt)) + 1
    starts = np.concatenate([[0], edges])
    lens = np.diff(np.concatenate([starts, [len(bwt)]]))
    f = hostlib.rb2_fmd_init()
    for s, l in zip(bwt[starts].tolist(), lens.tolist()):
        hostlib.rb2_fmd_push(f, l, s)
    hostlib.rb2_fmd_finish(f)
    out = tmp_path / "x.fmd"
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    fp = libc.fopen(str(out).encode(), b"wb")
    hostlib.rb2_fmd_write.argtypes = [C.c_void_p, C.c_void_p]
    assert hostlib.rb2_fmd_write(f, fp) == 0
    libc.fclose(fp)
    hostlib.rb2_fmd_destroy(f)
    assert H.md5(out.read_bytes()) == g["fmd_md5"][SO_FLAG[so] + "d"]


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_fmr_roundtrip_with_reference(golden, tmp_path):
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    ours = tmp_path / "ours.fmr"
    ours.write_bytes(cli(["-LRbs", "-m0"], text))
    ref = subprocess.run([H.REF_BIN, "-d", "-i", str(ours), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert H.md5(ref) == g["fmd_md5"]["-LRsd"]
    theirs = tmp_path / "ref.fmr"
    theirs.write_bytes(H.run_ref(["-LRbr"], text))
    mine = subprocess.run([CLI, "-d", "-i", str(theirs), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    assert H.md5(mine) == g["fmd_md5"]["-LRrd"]


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("flags", [[], ["-N"], ["-q", "20"], ["-C"], ["-F"], ["-R", "-s"], ["-r"], ["-N", "-C", "-r"]])
def test_fastq_fasta_parsing_and_filters(flags, tmp_path):
    rng = np.random.RandomState(5)
    recs = []
    for i in range(300):
        ln = rng.randint(0, 70)
        seq = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04]) for _ in range(ln))
        if i % 7 == 0 and ln >= 4 and ln % 2 == 0:          # some reverse-complement palindromes for -C
            h = seq[:ln // 2].replace("N", "A")
            seq = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))
        qual = "".join(chr(33 + rng.randint(2, 41)) for _ in range(ln))
        recs.append((seq, qual))
    fq = "".join("@r%d desc\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(recs)).encode()
    fa = "".join(">r%d\n%s\n" % (i, "\n".join(s[j:j + 25] for j in range(0, len(s), 25))) for i, (s, q) in enumerate(recs)).encode()
    for data in (fq, fa):
        if data is fa and "-q" in flags:
            continue
        ref = H.run_ref(flags + ["-m0"], data)
        assert cli(flags + ["-m0"], data) == ref


def test_x_needs_batch_mode():
    p = subprocess.run([CLI, "-x", "5", "-m0", "-"], input=b">a\nACGT\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"cannot be used with '-m0'" in p.stderr        # main.c:166-169


def make_fastx(seed=5, n=300):
    rng = np.random.RandomState(seed)
    recs = []
    for i in range(n):
        ln = rng.randint(0, 70)
        seq = "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04]) for _ in range(ln))
        if i % 7 == 0 and ln >= 4 and ln % 2 == 0:
            h = seq[:ln // 2].replace("N", "A")
            seq = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))
        qual = "".join(chr(33 + rng.randint(2, 41)) for _ in range(ln))
        recs.append((seq, qual))
    fq = "".join("@r%d desc\n%s\n+\n%s\n" % (i, s, q) for i, (s, q) in enumerate(recs)).encode()
    fa = "".join(">r%d\n%s\n" % (i, "\n".join(s[j:j + 25] for j in range(0, len(s), 25))) for i, (s, q) in enumerate(recs)).encode()
    return fq, fa


def test_rle_rope_random_against_model(hostlib):
    """rope_insert_run / rope_rank2a on a small-block rope vs a python list model"""
    L = hostlib
    L.rope_init.restype = C.c_void_p
    L.rope_insert_run.restype = C.c_int64
    L.rope_insert_run.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_void_p]
    L.rope_rank2a.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.rope_destroy.argtypes = [C.c_void_p]
    r = L.rope_init(4, 32)                  # tiny nodes and leaves: many splits, several levels
    model = []
    rng = np.random.RandomState(1)
    for it in range(3000):
        x = int(rng.randint(0, len(model) + 1))
        a = int(rng.randint(0, 6))
        rl = int(rng.choice([1, 1, 1, 2, 5, 17, 300]))
        got = L.rope_insert_run(r, x, a, rl, None)
        assert got == model[:x].count(a)
        model[x:x] = [a] * rl
        if it % 50 == 0:
            cx = (C.c_int64 * 6)()
            cy = (C.c_int64 * 6)()
            x = int(rng.randint(0, len(model) + 1))
            y = int(rng.randint(x, len(model) + 1))
            L.rope_rank2a(r, x, y, cx, cy)
            assert list(cx) == [model[:x].count(s) for s in range(6)]
            assert list(cy) == [model[:y].count(s) for s in range(6)]
    L.rope_destroy(r)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("l,n", [(32, 4), (32, 8), (64, 4)])
def test_restore_reference_fmr_with_full_buckets(l, n, tmp_path):
    """the reference leaves buckets with n == max_nodes at rest and dumps them (rope.c:120-124 splits them on the NEXT
    descent); restoring such a file and inserting with -m0 must split them too (used to overflow the bucket)"""
    rng = np.random.RandomState(l * 100 + n)
    for it in range(6):
        so = ["", "-s", "-r"][it % 3]
        mk = lambda k: H.lines_from_codes([rng.randint(1, 5, size=rng.randint(1, 60)) for _ in range(k)])
        a, b = mk(int(rng.randint(100, 700))), mk(int(rng.randint(100, 700)))
        flags = [x for x in ["-L", "-m0", so, "-l%d" % l, "-n%d" % n] if x]
        f = tmp_path / "ref.fmr"
        f.write_bytes(H.run_ref(flags + ["-b"], a))
        want = H.run_ref(["-L", "-m0", "-i", str(f)], b)
        assert cli(["-L", "-m0", "-i", str(f)], b) == want
        g = tmp_path / "our.fmr"                           # and our .fmr (same options) continues in the reference
        g.write_bytes(cli(flags + ["-b"], a))
        assert H.run_ref(["-L", "-m0", "-i", str(g)], b) == want


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_reader_edge_cases_match_reference():
    """kseq-compatible reading (main.c:177-187, kseq.h): raw sequence lines (blanks become N), trailing CR, truncated
    quality ends the input, header without newline, sizes that are multiples of kseq's 16 KiB buffer, unknown options"""
    line = b"ACGTACGTACGTACG\n"
    fa16k = b">r\n" + b"ACGT" * 4095 + b"\n"
    cases = [b"", b"\n", b">", b">a", b">a\n", b"@a\nACG\n+\nII\n@b\nAC\n+\nII\n", b"@a\nACG\n+", b"@a\nACG\n+\n",
             b"ACGT\r\nAC\r\n", b">x y\nAC GT\r\nA\tC\n>z\n\r\nAC\n", b">r c\n\tGT TG\t\n>s\n\n", fa16k, fa16k + b">", fa16k + b">q\nAC"]
    for data in cases:
        for flags in (["-m0", "-d"], ["-m0", "-q", "20"], ["-m0", "-N", "-r"], ["-m0", "-Z", "-R"]):
            p = subprocess.run([H.REF_BIN] + flags + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
            assert p.returncode == 0
            assert cli(flags, data) == p.stdout, (flags, data[:40])
    for data in (b"", line * 1024, (line * 1024)[:-1], b"ACG", b"ACG\n\n"):   # -L: an input of k x 16384 bytes ends with one extra empty read
        p = subprocess.run([H.REF_BIN, "-L", "-R", "-m0", "-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        assert p.returncode == 0
        assert cli(["-L", "-R", "-m0"], data) == p.stdout, data[:20]


# ---- parallel .fmd writer (fmd.c: speculative segments + stitching) == sequential writer, byte for byte -------------------

def _fmd_lib():
    from ropebwt2_amd.build import lib_path
    L = C.CDLL(lib_path("libropebwt2.so"))
    L.rb2_fmd_init.restype = C.c_void_p
    L.rb2_fmd_push_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.rb2_fmd_finish.argtypes = [C.c_void_p]
    L.rb2_fmd_destroy.argtypes = [C.c_void_p]
    L.rb2_fmd_write_path.argtypes = [C.c_void_p, C.c_char_p]
    L.rb2_fmdp_init.restype = C.c_void_p
    L.rb2_fmdp_init.argtypes = [C.c_int, C.c_int64]
    L.rb2_fmdp_push_runs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.rb2_fmdp_finish.restype = C.c_void_p
    L.rb2_fmdp_finish.argtypes = [C.c_void_p]
    return L


def _write_both(L, chunks, threads, seg, tmp_path):
    """chunks: list of uint8 arrays holding whole runs; returns (sequential bytes, parallel bytes)"""
    f = L.rb2_fmd_init()
    for r in chunks:
        L.rb2_fmd_push_runs(f, r.ctypes.data, len(r))
    L.rb2_fmd_finish(f)
    a = str(tmp_path / "seq.fmd").encode()
    assert L.rb2_fmd_write_path(f, a) == 0
    L.rb2_fmd_destroy(f)
    p = L.rb2_fmdp_init(threads, seg)
    for r in chunks:
        L.rb2_fmdp_push_runs(p, r.ctypes.data, len(r))
    f = L.rb2_fmdp_finish(p)
    b = str(tmp_path / "par.fmd").encode()
    assert L.rb2_fmd_write_path(f, b) == 0
    L.rb2_fmd_destroy(f)
    return open(a, "rb").read(), open(b, "rb").read()


@pytest.mark.parametrize("n,chunk,threads,seg", [(0, 64, 2, 4096), (1, 64, 2, 4096), (5000, 100, 3, 256), (200000, 1000, 4, 4096),
                                                 (3000000, 1024, 8, 65536), (3000000, 777, 5, 1 << 20)])
def test_parallel_fmd_writer_one_byte_runs(n, chunk, threads, seg, tmp_path):
    """device-style input: one-byte runs, equal symbols in adjacent bytes (cut runs) merge in the writer; segments from far
    smaller than the coupling distance (the true orbit does nearly everything) to large (nearly everything is copied)"""
    rng = np.random.RandomState(n % 97 + threads)
    runs = (rng.choice([1, 1, 1, 2, 2, 3, 5, 15], size=n).astype(np.uint8) << 3 | rng.randint(0, 6, size=n).astype(np.uint8)).astype(np.uint8)
    a, b = _write_both(_fmd_lib(), [runs[i:i + chunk] for i in range(0, n, chunk)], threads, seg, tmp_path)
    assert a == b


@pytest.mark.parametrize("threads,seg", [(3, 2048), (4, 50000), (6, 1 << 20)])
def test_parallel_fmd_writer_wide_runs(threads, seg, tmp_path):
    """2/4/8-byte runs up to 2^40 symbols: 32- and 64-bit block headers, runs that straddle segments"""
    from ropebwt2_amd.hipbwt import encode_runs
    rng = np.random.RandomState(7)
    n = 120000
    lens = np.where(rng.rand(n) < 0.7, rng.randint(1, 16, size=n), np.where(rng.rand(n) < 0.5, rng.randint(16, 5000, size=n), rng.randint(1 << 20, 1 << 40, size=n)))
    syms = rng.randint(0, 6, size=n)
    chunks, k = [], 0
    while k < n:
        k2 = min(n, k + int(rng.randint(1, 400)))
        out = bytearray()
        for c, l in zip(syms[k:k2].tolist(), lens[k:k2].tolist()):       # every run on its own: equal neighbours stay separate runs
            out += encode_runs(np.full(1, c, np.uint8)).tobytes() if l == 1 else b""
            if l > 1:
                nb = 1 if l < 16 else 2 if l < 256 else 4 if l < (1 << 19) else 8
                if nb == 1:
                    out.append(l << 3 | c)
                else:
                    tail, ll = [], l
                    for _ in range(nb - 1):
                        tail.append(0x80 | (ll & 0x3f)); ll >>= 6
                    out.append({2: 0xC0, 4: 0xE0, 8: 0xF0}[nb] | ll << 3 | c); out.extend(reversed(tail))
        chunks.append(np.frombuffer(bytes(out), np.uint8))
        k = k2
    a, b = _write_both(_fmd_lib(), chunks, threads, seg, tmp_path)
    assert a == b


@pytest.mark.parametrize("threads,seg", [(1, 256), (3, 2000), (4, 70000), (8, 0)])
def test_parallel_fmd_writer_ignores_empty_runs_like_the_sequential_one(threads, seg, tmp_path):
    """zero-length runs (a hand-made .fmr may hold them; the reference's coder drops them, rld0.c:155) must not separate two runs of one
    symbol -- also not across a segment border --, and fresh words of the output array may hold garbage (RB2_FMD_POISON)"""
    os.environ["RB2_FMD_POISON"] = "1"
    try:
        for seed in (5, 6, 7):
            stream = np.frombuffer(_mixed_run_stream(seed, 80000), np.uint8)
            cuts = [0]
            rng = np.random.RandomState(seed)
            while cuts[-1] < len(stream):
                c = min(len(stream), cuts[-1] + int(rng.randint(1, 30000)))
                while c < len(stream) and (stream[c] & 0xC0) == 0x80:
                    c += 1
                cuts.append(c)
            a, b = _write_both(_fmd_lib(), [stream[x:y] for x, y in zip(cuts[:-1], cuts[1:])], threads, seg, tmp_path)
            assert a == b, seed
    finally:
        del os.environ["RB2_FMD_POISON"]


def test_parallel_fmd_writer_crosses_a_chunk_border(tmp_path):
    """more than 64 MiB of output: the last block of a 2^23-word chunk is one word shorter (rld0.h:75), which the speculative
    encodings cannot know -- the stitcher re-encodes from there until it couples again"""
    rng = np.random.RandomState(3)
    n = 70_000_000
    runs = (rng.choice([1, 2, 3, 7], size=n).astype(np.uint8) << 3 | rng.randint(1, 5, size=n).astype(np.uint8)).astype(np.uint8)
    a, b = _write_both(_fmd_lib(), [runs[i:i + (1 << 20)] for i in range(0, n, 1 << 20)], 6, 0, tmp_path)
    assert len(a) > (1 << 26) + 4096 and a == b


@pytest.mark.parametrize("n,step,offset", [(0, 1, 0), (40, 1, 0), (300000, 64, 13), (3000000, 1024, 0), (3000000, 0, 4096)])
def test_fmd_streamed_to_the_file_while_encoded(n, step, offset, tmp_path):
    """rb2_fmdp_set_output: a writer thread pwrites the final words while the stitcher is still working (step: words per write,
    0 = the 32 MiB default, i.e. everything left to rb2_fmd_write); the file equals the sequential writer's, also behind other bytes"""
    L = _fmd_lib()
    L.rb2_fmdp_set_output.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    libc = C.CDLL(None)
    libc.fdopen.restype = C.c_void_p; libc.fdopen.argtypes = [C.c_int, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.ftello.argtypes = [C.c_void_p]; libc.ftello.restype = C.c_int64
    libc.fseeko.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.rb2_fmd_write.argtypes = [C.c_void_p, C.c_void_p]
    rng = np.random.RandomState(n % 89)
    runs = (rng.choice([1, 1, 2, 3, 9, 15], size=n).astype(np.uint8) << 3 | rng.randint(0, 6, size=n).astype(np.uint8)).astype(np.uint8)
    chunks = [runs[i:i + 5000] for i in range(0, n, 5000)]
    f = L.rb2_fmd_init()
    for r in chunks:
        L.rb2_fmd_push_runs(f, r.ctypes.data, len(r))
    L.rb2_fmd_finish(f)
    a = str(tmp_path / "seq.fmd").encode()
    assert L.rb2_fmd_write_path(f, a) == 0
    L.rb2_fmd_destroy(f)
    want = open(a, "rb").read()
    if step:
        os.environ["RB2_FMD_OUT_STEP"] = str(step)
    try:
        out = tmp_path / "streamed.fmd"
        fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        fp = libc.fdopen(fd, b"wb")
        os.write(fd, b"x" * offset)
        libc.fseeko(fp, offset, 0)
        p = L.rb2_fmdp_init(4, 4096)
        assert L.rb2_fmdp_set_output(p, fd, offset) == 0
        for r in chunks:
            L.rb2_fmdp_push_runs(p, r.ctypes.data, len(r))
        f = L.rb2_fmdp_finish(p)
        assert L.rb2_fmd_write(f, fp) == 0
        assert libc.ftello(fp) == offset + len(want)                    # the FILE continues behind the index
        L.rb2_fmd_destroy(f)
        libc.fclose(fp)
    finally:
        os.environ.pop("RB2_FMD_OUT_STEP", None)
    got = out.read_bytes()
    assert got[:offset] == b"x" * offset and got[offset:] == want


def test_fmd_streaming_is_refused_where_offsets_do_not_work(tmp_path):
    """pipes and O_APPEND files keep the one-shot writer"""
    L = _fmd_lib()
    L.rb2_fmdp_set_output.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    p = L.rb2_fmdp_init(2, 0)
    r, w = os.pipe()
    assert L.rb2_fmdp_set_output(p, w, 0) == -1
    fd = os.open(tmp_path / "a", os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
    assert L.rb2_fmdp_set_output(p, fd, 0) == -1
    fd2 = os.open(tmp_path / "a", os.O_RDONLY)
    assert L.rb2_fmdp_set_output(p, fd2, 0) == -1
    f = L.rb2_fmdp_finish(p)
    assert L.rb2_fmd_write_path(f, str(tmp_path / "b").encode()) == 0
    L.rb2_fmd_destroy(f)
    for x in (r, w, fd, fd2):
        os.close(x)


def test_cli_fmd_to_a_file_is_streamed_and_identical(golden, tmp_path):
    """-o FILE and a redirected stdout take the streamed path; same bytes as through a pipe"""
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    for mode in ("-o", ">", "no_stream"):
        out = tmp_path / ("o_%s.fmd" % mode.strip("->"))
        env = dict(os.environ, RB2_FMD_SEGMENT="3000", RB2_FMD_OUT_STEP="256")
        if mode == "no_stream":
            env["RB2_FMD_NO_STREAM"] = "1"
        if mode == ">":
            with open(out, "wb") as fo:
                p = subprocess.run([CLI, "-LRsd", "-m0", "-v4", "-"], input=text, stdout=fo, stderr=subprocess.PIPE, env=env)
        else:
            p = subprocess.run([CLI, "-LRsd", "-m0", "-v4", "-o", str(out), "-"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()[-400:]
        assert (b"written while it is encoded" in p.stderr) == (mode != "no_stream")
        assert H.md5(out.read_bytes()) == g["fmd_md5"]["-LRsd"], mode


def test_cli_fmd_small_segments_m0(golden):
    """the CLI's .fmd through the parallel writer with segments far smaller than a leaf chunk and with one thread"""
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(g["n_reads"], g["read_len"], g["seed"]))
    for seg, thr in (("700", "3"), ("50000", "1"), ("0", "0")):
        env = dict(os.environ, RB2_FMD_SEGMENT=seg, RB2_FMD_THREADS=thr)
        p = subprocess.run([CLI, "-LRsd", "-m0", "-"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0 and H.md5(p.stdout) == g["fmd_md5"]["-LRsd"]


# ---- the -L reader in batch mode: worker threads encode blocks of whole lines; same strings, same batch boundaries ---------------

def _dump_batches(flags, data, env, tmp_path, tag):
    f = tmp_path / ("batches_%s.bin" % tag)
    if f.exists():
        f.unlink()
    e = dict(os.environ, RB2_DUMP_BATCHES=str(f))
    e.update(env)
    p = subprocess.run([CLI] + flags + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    return f.read_bytes() if f.exists() else b""


@pytest.mark.parametrize("flags", [["-L", "-R"], ["-L"], ["-L", "-F"], ["-L", "-N"], ["-L", "-x", "5"], ["-L", "-x", "4", "-C"], ["-L", "-R", "-C"]])
def test_parallel_line_reader_equals_sequential(flags, tmp_path):
    """RB2_DUMP_BATCHES writes every batch (length + bytes) instead of inserting it: the threaded -L reader must produce the
    same byte stream cut into the same batches as the sequential loop, for block sizes down to a few bytes (carry-over of
    unfinished lines, lines longer than a block), CR line ends, empty lines, a last line without newline, N's and filters"""
    rng = np.random.RandomState(len(" ".join(flags)))
    lines = []
    for i in range(4000):
        L = int(rng.choice([0, 1, 3, 20, 101, 400], p=[.05, .05, .1, .3, .4, .1]))
        s = "".join(rng.choice(list("ACGTNacgtn"), size=L, p=[.22, .22, .22, .22, .03, .02, .02, .02, .02, .01]))
        if rng.rand() < 0.03:
            s += " trailing words"
        if rng.rand() < 0.05 and L >= 2 and L % 2 == 0:                 # its own reverse complement: -C trims it
            h = s[:L // 2].upper().replace("N", "A")
            s = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))
        lines.append(s + ("\r" if rng.rand() < 0.1 else ""))
    for data in ("\n".join(lines) + "\n", "\n".join(lines), "", "\n", "ACGT", "ACGT\n" * 4096, ("A" * 15 + "\n") * 1024):
        data = data.encode()
        for m in ("-m20k", "-m3k", "-m1g"):
            want = _dump_batches(flags + [m], data, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq")
            for chunk, thr in (("64", "3"), ("1000", "2"), ("70000", "5"), ("0", "4")):
                env = {"RB2_PARSE_THREADS": thr}
                if chunk != "0":
                    env["RB2_PARSE_CHUNK"] = chunk
                got = _dump_batches(flags + [m], data, env, tmp_path, "par")
                assert got == want, (flags, m, chunk, thr, len(data), len(got), len(want))


def _dump_batches_file(flags, path, env, tmp_path, tag):
    f = tmp_path / ("batches_%s.bin" % tag)
    if f.exists():
        f.unlink()
    e = dict(os.environ, RB2_DUMP_BATCHES=str(f), RB2_NO_RESERVE="1")
    e.update(env)
    p = subprocess.run([CLI] + flags + [str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    return (f.read_bytes() if f.exists() else b""), p.stderr


@pytest.mark.parametrize("flags", [["-L", "-R"], ["-L"], ["-L", "-N"], ["-L", "-x", "4", "-C"]])
def test_line_blocks_read_by_the_workers_equal_sequential(flags, tmp_path):
    """-L on a named plain file: no reader thread, block k = the lines that START in bytes [k, k + 1) * chunk, pread by its worker
    (pjob_fill_direct).  Same batch stream as the sequential loop on the same bytes from stdin -- lines that span many blocks, blocks
    in which no line starts, CR line ends, no final newline, an empty file, files of k x 16384 bytes (kseq's phantom empty line)"""
    rng = np.random.RandomState(7 + len(" ".join(flags)))
    lines = []
    for i in range(3000):
        L = int(rng.choice([0, 1, 3, 20, 101, 400, 5000], p=[.05, .05, .1, .3, .38, .1, .02]))
        s = "".join(rng.choice(list("ACGTNacgtn"), size=L, p=[.22, .22, .22, .22, .03, .02, .02, .02, .02, .01]))
        lines.append(s + ("\r" if rng.rand() < 0.1 else ""))
    big = "\n".join(lines)
    pad = lambda t, n: t + "A" * ((-len(t) - 1) % n) + "\n"                 # a multiple of n bytes, newline last
    cases = {"many": big + "\n", "no_final_newline": big, "empty": "", "newline": "\n", "one": "ACGT", "k16384_nl": pad(big[:40000], 16384), "k16384_no_nl": pad(big[:40000], 16384)[:-1] + "C",
             "exactly_one_buffer": ("A" * 15 + "\n") * 1024, "one_line_longer_than_all_blocks": "ACGT" * 9000 + "\n" + "GG\n"}
    for name, data in cases.items():
        data = data.encode()
        path = tmp_path / ("%s.txt" % name)
        path.write_bytes(data)
        for m in ("-m20k", "-m1g"):
            want = _dump_batches(flags + [m], data, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq")
            for chunk, thr in (("64", "3"), ("1000", "2"), ("16384", "4"), ("70000", "5"), ("0", "4")):
                env = {"RB2_PARSE_THREADS": thr, "RB2_PARSE_TRACE": "1"}
                if chunk != "0":
                    env["RB2_PARSE_CHUNK"] = chunk
                got, err = _dump_batches_file(flags + [m], path, env, tmp_path, "direct")
                assert b"read by the workers" in err
                assert got == want, (name, flags, m, chunk, thr, len(data), len(got), len(want))
        got, err = _dump_batches_file(flags + ["-m1g"], path, {"RB2_PARSE_THREADS": "4", "RB2_PARSE_CHUNK": "1000", "RB2_NO_DIRECT_BLOCKS": "1", "RB2_PARSE_TRACE": "1"}, tmp_path, "reader")
        assert b"read by the workers" not in err and got == _dump_batches(flags + ["-m1g"], data, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq"), name


def test_line_reader_batches_match_python_model(tmp_path):
    """the dumped batch of the threaded reader is what helpers.encode_batch builds (main.c:200-237)"""
    codes = H.splitmix_bases(3000, 50, seed=2)
    text = H.reads_to_text(codes)
    got = _dump_batches(["-L", "-m1g"], text, {"RB2_PARSE_THREADS": "4", "RB2_PARSE_CHUNK": "5000"}, tmp_path, "model")
    want = H.encode_batch_fixed(codes, True, True).tobytes()
    assert got[8:] == want and int.from_bytes(got[:8], "little") == len(want)


# ---- the FASTQ reader in batch mode: blocks of whole four-line records parsed by worker threads; anything that is not strict
# ---- four-line FASTQ sends the rest of the input back to the sequential (kseq-exact) reader ------------------------------------

def _fastq_inputs():
    rng = np.random.RandomState(12)
    recs = []
    for i in range(1500):
        L = int(rng.choice([0, 1, 2, 30, 101, 250], p=[.04, .04, .06, .3, .46, .1]))
        s = "".join(rng.choice(list("ACGTNacgtn"), size=L, p=[.22, .22, .22, .22, .03, .02, .02, .02, .02, .01]))
        if rng.rand() < 0.05 and L >= 2 and L % 2 == 0:
            h = s[:L // 2].upper().replace("N", "A")
            s = h + h[::-1].translate(str.maketrans("ACGT", "TGCA"))
        q = "".join(chr(33 + int(x)) for x in rng.randint(0, 42, size=L))        # '@' (33+31), '+' and '>' occur inside quality strings
        recs.append((s, q))
    def fq(recs, eol="\n", last_nl=True):
        t = "".join("@r%d some comment%s%s%s+%s%s%s" % (i, eol, s, eol, "" if i % 3 else "r%d" % i, eol, q + eol) for i, (s, q) in enumerate(recs))
        return (t if last_nl else t[:-len(eol)]).encode()
    strict = fq(recs)
    out = {"strict": strict, "strict_crlf": fq(recs, "\r\n"), "strict_no_final_newline": fq(recs, last_nl=False), "empty": b"", "one": fq(recs[:1]),
           "header_only": b"@x\n", "truncated_record": strict[:len(strict) // 2]}
    # not strict somewhere in the middle: a sequence wrapped over two lines (quality too), a FASTA record, a blank line, qualities too short / too long
    k = len(recs) // 2
    s0, q0 = "ACGTACGTAC", "IIIIIIIIII"
    mid = {"multiline": "@m\n%s\n%s\n+\n%s\n%s\n" % (s0[:4], s0[4:], q0[:7], q0[7:]), "fasta": ">f\nACGTTTGA\nCCA\n", "blank": "\n",
           "qual_short": "@s\nACGTAC\n+\nIII\n", "qual_long": "@l\nACG\n+\nIIIII\n", "seq_starts_with_plus": "@p\n+CGT\n+\nIIII\n"}
    for name, frag in mid.items():
        out["mid_" + name] = fq(recs[:k]) + frag.encode() + fq(recs[k:])
    out["starts_with_garbage"] = b"xx\n" + strict
    out["starts_with_fasta"] = b">f\nACGT\n" + strict
    return out


@pytest.mark.parametrize("flags", [["-R"], [], ["-F"], ["-N"], ["-q", "20"], ["-x", "5", "-q", "15"], ["-C"], ["-s", "-R"]])
def test_parallel_fastq_reader_equals_sequential(flags, tmp_path):
    """the threaded FASTQ reader (blocks of whole four-line records, verified record by record) gives the batch stream of the
    sequential kseq-exact reader for strict inputs AND for inputs that stop being strict somewhere (fallback to the sequential reader
    from the failing block on), for block sizes down to 64 bytes"""
    for name, data in _fastq_inputs().items():
        for m in ("-m30k", "-m1g"):
            want = _dump_batches(flags + [m], data, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq")
            for chunk, thr in (("64", "3"), ("1500", "2"), ("100000", "5"), ("0", "4")):
                env = {"RB2_PARSE_THREADS": thr}
                if chunk != "0":
                    env["RB2_PARSE_CHUNK"] = chunk
                got = _dump_batches(flags + [m], data, env, tmp_path, "par")
                assert got == want, (name, flags, m, chunk, thr, len(data), len(got), len(want))


def _fasta_inputs():
    rng = np.random.RandomState(21)
    recs = []
    for i in range(900):
        L = int(rng.choice([0, 1, 2, 30, 101, 400, 3000], p=[.04, .04, .06, .3, .36, .15, .05]))
        recs.append("".join(rng.choice(list("ACGTNacgtn"), size=L, p=[.22, .22, .22, .22, .03, .02, .02, .02, .02, .01])))
    def fa(recs, width=0, eol="\n", last_nl=True, blank=False):
        out = []
        for i, s in enumerate(recs):
            out.append(">r%d%s" % (i, " a comment > with marks @ +" if i % 4 == 0 else "") + eol)
            if width:
                for k in range(0, len(s), width):
                    out.append(s[k:k + width] + eol)
                    if blank and k % (3 * width) == 0:
                        out.append(eol)
            else:
                out.append(s + eol)
        t = "".join(out)
        return (t if last_nl else t[:-len(eol)]).encode()
    plain = fa(recs)
    out = {"one_line": plain, "wrapped60": fa(recs, 60), "wrapped7_blank_lines": fa(recs, 7, blank=True), "crlf": fa(recs[:300], 60, "\r\n"),
           "no_final_newline": fa(recs, 60, last_nl=False), "one": fa(recs[:1]), "header_only": b">x\n", "header_no_newline": b">x", "two_headers": b">a\n>b\nACGT\n",
           "one_record_longer_than_many_blocks": fa(["".join(rng.choice(list("ACGT"), size=50_000))], 80) + fa(recs[:50], 60),
           "truncated": plain[:len(plain) // 2]}
    k = len(recs) // 2
    mid = {"fastq": "@q\nACGTAC\n+\nIIIIII\n", "plus_line": ">p\nACGT\n+\nIIII\n", "at_line": ">p\nACGT\n@x\nAC\n+\nII\n", "cr_inside": ">c\nAC\rGT\nA\r\nC\n",
           "spaces_and_tabs": ">s t\nAC GT\n\tA C\n", "gt_inside_a_line": ">g\nAC>GT\nAC\n"}
    for name, frag in mid.items():
        out["mid_" + name] = fa(recs[:k], 60) + frag.encode() + fa(recs[k:], 60)
    out["starts_with_garbage"] = b"xx\n" + plain
    out["starts_with_newline"] = b"\n" + plain
    return out


@pytest.mark.parametrize("flags", [["-R"], [], ["-F"], ["-N"], ["-x", "5"], ["-C"], ["-s", "-R"]])
def test_parallel_fasta_reader_equals_sequential(flags, tmp_path):
    """the threaded FASTA reader (blocks cut in front of a '>' line; header, then sequence lines concatenated) gives the batch stream
    of the sequential kseq-exact reader -- wrapped and blank lines, records longer than many blocks, header-only records -- and
    hands over to it (from the failing block on) where kseq's grammar does more: '+' / '@' lines, carriage returns, an unterminated
    last line"""
    for name, data in _fasta_inputs().items():
        for m in ("-m30k", "-m1g"):
            want = _dump_batches(flags + [m], data, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq")
            for chunk, thr in (("64", "3"), ("1500", "2"), ("100000", "5"), ("0", "4")):
                env = {"RB2_PARSE_THREADS": thr}
                if chunk != "0":
                    env["RB2_PARSE_CHUNK"] = chunk
                got = _dump_batches(flags + [m], data, env, tmp_path, "par")
                assert got == want, (name, flags, m, chunk, thr, len(data), len(got), len(want))


def test_parallel_fasta_reader_really_runs_and_falls_back(tmp_path):
    ins = _fasta_inputs()
    for name, expect in (("wrapped60", False), ("mid_plus_line", True), ("crlf", True), ("starts_with_garbage", False)):
        f = tmp_path / "b.bin"
        p = subprocess.run([CLI, "-R", "-m1g", "-"], input=ins[name], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, RB2_DUMP_BATCHES=str(f), RB2_PARSE_THREADS="4", RB2_PARSE_CHUNK="4096", RB2_PARSE_TRACE="1"))
        assert p.returncode == 0
        assert (b"not plain FASTA" in p.stderr) == expect, (name, p.stderr.decode()[-300:])
        if name == "wrapped60":
            import re
            mm = re.search(rb"(\d+) blocks of FASTA records parsed by 4 threads", p.stderr)
            assert mm and int(mm.group(1)) > 10 and b"then the sequential" not in p.stderr, p.stderr.decode()[-300:]
        if name == "starts_with_garbage":
            assert b"parsed by" not in p.stderr


def test_named_files_plain_and_gz(tmp_path):
    """a named plain file is read with read(2) by the reader thread (zlib bypassed), a .gz through zlib; the fallback of the FASTQ
    reader continues the plain file from where the reader thread stopped -- same batches as the sequential reader in all cases"""
    import gzip
    ins = _fastq_inputs()
    lines = H.reads_to_text(H.splitmix_bases(4000, 60, seed=8))
    fas = _fasta_inputs()
    cases = [("strict", ins["strict"], ["-R"]), ("mid_multiline", ins["mid_multiline"], ["-R"]), ("mid_fasta", ins["mid_fasta"], []), ("lines", lines, ["-L", "-R"]),
             ("fasta", fas["wrapped60"], ["-R"]), ("fasta_mid_plus", fas["mid_plus_line"], [])]
    for name, data, flags in cases:
        plain = tmp_path / (name + ".txt")
        plain.write_bytes(data)
        gz = tmp_path / (name + ".txt.gz")
        with gzip.open(gz, "wb") as fp:
            fp.write(data)
        outs = []
        for path in (plain, gz):
            for thr, extra in (("1", {}), ("4", {"RB2_PARSE_CHUNK": "3000"}), ("4", {"RB2_PARSE_CHUNK": "3000", "RB2_NO_DIRECT_READ": "1"})):
                f = tmp_path / "o.bin"
                if f.exists():
                    f.unlink()
                e = dict(os.environ, RB2_DUMP_BATCHES=str(f), RB2_NO_RESERVE="1", RB2_PARSE_THREADS=thr)
                e.update(extra)
                p = subprocess.run([CLI] + flags + ["-m25k", str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
                assert p.returncode == 0, p.stderr.decode()[-300:]
                outs.append(f.read_bytes())
        assert all(o == outs[0] for o in outs), name


def test_big_named_fastq_and_fasta_blocks_of_16_mib(tmp_path):
    """default 16 MiB blocks on files of several blocks: the reader thread fills a block with four preads side by side (rb2_par_pread);
    same batches as the sequential reader, also when the input stops being strict in the third block"""
    fq, fa = _fastq_inputs(), _fasta_inputs()
    big_fq = fq["strict"] * 140                                       # ~45 MB
    cases = [("fq", big_fq, ["-R"]), ("fq_broken", big_fq[:36 << 20] + b"@m\nACGT\nAC\n+\nIIII\nII\n" + fq["strict"], ["-R"]), ("fa", fa["wrapped60"] * 70, ["-R"])]
    for name, data, flags in cases:
        path = tmp_path / (name + ".txt")
        path.write_bytes(data)
        want, _ = _dump_batches_file(flags + ["-m8m"], path, {"RB2_PARSE_THREADS": "1"}, tmp_path, "seq")
        got, err = _dump_batches_file(flags + ["-m8m"], path, {"RB2_PARSE_THREADS": "5", "RB2_PARSE_TRACE": "1"}, tmp_path, "par")
        assert b"parsed by 5 threads" in err and (b"then the sequential" in err) == (name == "fq_broken"), err.decode()[-300:]
        assert got == want, name


def test_parallel_fastq_reader_really_runs_and_falls_back(tmp_path):
    """the strict input is parsed by the workers (no fallback message), the broken one reports where it went sequential"""
    ins = _fastq_inputs()
    for name, expect in (("strict", False), ("mid_multiline", True), ("starts_with_fasta", False)):
        f = tmp_path / "b.bin"
        p = subprocess.run([CLI, "-R", "-m1g", "-"], input=ins[name], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, RB2_DUMP_BATCHES=str(f), RB2_PARSE_THREADS="4", RB2_PARSE_CHUNK="4096", RB2_PARSE_TRACE="1"))
        assert p.returncode == 0
        assert (b"not four-line FASTQ" in p.stderr) == expect, (name, p.stderr.decode()[-300:])
        if name == "strict":
            import re
            mm = re.search(rb"(\d+) blocks of four-line FASTQ records parsed by 4 threads", p.stderr)
            assert mm and int(mm.group(1)) > 10 and b"then the sequential" not in p.stderr, p.stderr.decode()[-300:]
        if name == "starts_with_fasta":                               # starts with '>': the FASTA workers take it, and hand over at the first '@' line
            assert b"not plain FASTA" in p.stderr and b"then the sequential reader" in p.stderr


def test_fmr_dump_to_a_file_equals_dump_to_a_pipe(golden, tmp_path):
    """mr_dump writes the six ropes of a regular file from six threads (pwrite at known offsets); a pipe -- and
    RB2_DUMP_SEQUENTIAL=1 -- takes the reference's sequential fwrite path.  Same bytes, and the reference restores them."""
    text = H.reads_to_text(H.splitmix_bases(3000, 101, 42))
    for so in ("", "s", "r"):
        piped = cli(["-LRb" + so, "-m0"], text)
        f = tmp_path / ("par%s.fmr" % so)
        for thr in ("1", "6", "16"):                         # (small indexes take the sequential path unless asked)
            env = dict(os.environ, RB2_DUMP_THREADS=thr)
            assert subprocess.run([CLI, "-LRb" + so, "-m0", "-o", str(f), "-"], input=text, stderr=subprocess.DEVNULL, env=env).returncode == 0
            assert f.read_bytes() == piped
        g = tmp_path / ("seq%s.fmr" % so)
        env = dict(os.environ, RB2_DUMP_SEQUENTIAL="1")
        assert subprocess.run([CLI, "-LRb" + so, "-m0", "-o", str(g), "-"], input=text, stderr=subprocess.DEVNULL, env=env).returncode == 0
        assert g.read_bytes() == piped


def test_restore_runs_then_host_operations(hostlib, tmp_path):
    """mr_restore_runs keeps a restored .fmr as run bytes (what a GPU build needs); a host operation that comes first --
    a rank query, a dump -- makes mr_sync_host bulk-load the trees from them.  Ranks and the BWT equal mr_restore's."""
    L = hostlib
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    for fn in ("mr_restore", "mr_restore_runs"):
        getattr(L, fn).restype = C.c_void_p; getattr(L, fn).argtypes = [C.c_void_p]
    L.mr_rank2a.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.mr_dump.argtypes = [C.c_void_p, C.c_void_p]
    L.mr_destroy.argtypes = [C.c_void_p]
    text = H.reads_to_text(H.splitmix_bases(2500, 101, 7)) + b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n" * 40
    for so in ("", "s"):
        f = tmp_path / ("in%s.fmr" % so)
        f.write_bytes(cli(["-LRb" + so, "-m0"], text))
        want_bwt = cli(["-m0", "-i", str(f)], b"")
        ranks = {}
        for fn in ("mr_restore", "mr_restore_runs"):
            fp = libc.fopen(str(f).encode(), b"rb")
            mr = getattr(L, fn)(fp)
            libc.fclose(fp)
            assert mr
            out = []
            for x in (0, 1, 1000, 123456, 2500 * 102 + 40 * 90):
                cx = (C.c_int64 * 6)()
                L.mr_rank2a(mr, x, -1, cx, None)
                out.append(list(cx))
            ranks[fn] = out
            g = tmp_path / ("%s%s.fmr" % (fn, so))
            fo = libc.fopen(str(g).encode(), b"wb")
            L.mr_dump(mr, fo)
            libc.fclose(fo)
            L.mr_destroy(mr)
            assert cli(["-m0", "-i", str(g)], b"") == want_bwt
        assert ranks["mr_restore"] == ranks["mr_restore_runs"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_bulk_loader_threads_build_the_same_tree(hostlib, tmp_path, seed):
    """rope_load_runs_mt cuts the run stream where nothing merges, brings the segments into canonical form in parallel, derives
    the leaf starts from the codec's continuation marks and fills the leaves in parallel: the dump must equal rope_load_runs'
    byte for byte -- on streams with runs that merge, empty runs, 2/4/8-byte runs and long stretches of one-byte runs."""
    from ropebwt2_amd.hipbwt import encode_runs
    L = hostlib
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.rope_init.restype = C.c_void_p; L.rope_init.argtypes = [C.c_int, C.c_int]
    L.rope_load_runs.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rope_load_runs_mt.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int]
    L.rope_dump.argtypes = [C.c_void_p, C.c_void_p]
    L.rope_destroy.argtypes = [C.c_void_p]
    rng = np.random.RandomState(seed)
    n = 600_000
    sym = rng.randint(0, 6, size=n).astype(np.uint8)
    ln = rng.randint(1, 16, size=n).astype(np.uint8)
    same = rng.rand(n) < 0.02                                 # neighbours that merge
    sym[1:][same[1:]] = sym[:-1][same[1:]]
    ln[rng.rand(n) < 0.003] = 0                               # empty runs
    plain = (ln << 3 | sym).astype(np.uint8)
    parts, at = [], 0
    for cut in sorted(rng.randint(0, n, size=40)):            # wide runs in between (2, 4 and 8 byte codes)
        parts.append(plain[at:cut].tobytes())
        wide = np.repeat(np.uint8(rng.randint(0, 6)), int(rng.choice([20, 300, 70_000, 600_000])))
        parts.append(encode_runs(wide))
        at = cut
    parts.append(plain[at:].tobytes())
    stream = b"".join(parts)
    os.environ["RB2_LOAD_MIN_SEG"] = "4096"
    try:
        outs = []
        for thr in (0, 2, 5, 16):
            r = L.rope_init(64, 512) if thr != 5 else L.rope_init(6, 64)
            if thr == 0:
                L.rope_load_runs(r, stream, len(stream))
            else:
                L.rope_load_runs_mt(r, stream, len(stream), thr)
            f = tmp_path / ("t%d.bin" % thr)
            fp = libc.fopen(str(f).encode(), b"wb")
            L.rope_dump(r, fp)
            libc.fclose(fp)
            L.rope_destroy(r)
            outs.append(f.read_bytes())
        assert outs[0] == outs[1] == outs[3]
        r = L.rope_init(6, 64)                                 # small leaves and buckets: many levels
        L.rope_load_runs(r, stream, len(stream))
        f = tmp_path / "small.bin"
        fp = libc.fopen(str(f).encode(), b"wb"); L.rope_dump(r, fp); libc.fclose(fp); L.rope_destroy(r)
        assert f.read_bytes() == outs[2]
    finally:
        del os.environ["RB2_LOAD_MIN_SEG"]


def _mixed_run_stream(seed, n):
    from ropebwt2_amd.hipbwt import encode_runs
    rng = np.random.RandomState(seed)
    if n == 0:
        return b""
    sym = rng.randint(0, 6, size=n).astype(np.uint8)
    ln = rng.randint(1, 16, size=n).astype(np.uint8)
    same = rng.rand(n) < 0.02
    sym[1:][same[1:]] = sym[:-1][same[1:]]
    ln[rng.rand(n) < 0.003] = 0
    plain = (ln << 3 | sym).astype(np.uint8)
    parts, at = [], 0
    for cut in sorted(rng.randint(0, n, size=min(30, n))):
        parts.append(plain[at:cut].tobytes())
        parts.append(encode_runs(np.repeat(np.uint8(rng.randint(0, 6)), int(rng.choice([20, 300, 70_000, 600_000])))))
        at = cut
    parts.append(plain[at:].tobytes())
    return b"".join(parts)


@pytest.mark.parametrize("n,max_nodes,block_len,thr", [(0, 64, 512, 4), (1, 64, 512, 1), (300, 64, 512, 3), (40_000, 64, 512, 4), (40_000, 6, 64, 5), (40_000, 4, 32, 2),
                                                        (2_500_000, 64, 512, 16), (2_500_000, 10, 96, 7), (2_500_000, 5, 30, 3)])
def test_dump_without_trees_equals_the_dump_of_the_tree(hostlib, tmp_path, n, max_nodes, block_len, thr):
    """rope_rdump_*: rope_dump's bytes straight from the run stream (no arena, no nodes) -- header positions follow from the leaf
    count alone.  Against rope_load_runs + rope_dump for empty streams, one leaf, one bucket, and 2-14 levels of buckets."""
    L = hostlib
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.rope_init.restype = C.c_void_p; L.rope_init.argtypes = [C.c_int, C.c_int]
    L.rope_load_runs.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.rope_dump.argtypes = [C.c_void_p, C.c_void_p]
    L.rope_destroy.argtypes = [C.c_void_p]
    L.rope_rdump_prepare.restype = C.c_void_p; L.rope_rdump_prepare.argtypes = [C.c_char_p, C.c_int64, C.c_int, C.c_int, C.c_int]
    L.rope_rdump_size.restype = C.c_int64; L.rope_rdump_size.argtypes = [C.c_void_p]
    L.rope_rdump_write.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    stream = _mixed_run_stream(n % 7 + thr, n)
    r = L.rope_init(max_nodes, block_len)
    L.rope_load_runs(r, stream, len(stream))
    f = tmp_path / "tree.bin"
    fp = libc.fopen(str(f).encode(), b"wb"); L.rope_dump(r, fp); libc.fclose(fp); L.rope_destroy(r)
    want = f.read_bytes()
    d = L.rope_rdump_prepare(stream, len(stream), max_nodes, block_len, thr)
    assert L.rope_rdump_size(d) == len(want)
    g = tmp_path / "direct.bin"
    fd = os.open(g, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    os.write(fd, b"#" * 11)
    assert L.rope_rdump_write(d, fd, 11) == 0
    os.close(fd)
    assert g.read_bytes() == b"#" * 11 + want


def test_mr_dump_of_restored_run_bytes_needs_no_trees(hostlib, tmp_path):
    """mr_restore_runs + mr_dump to a regular file: the run bytes go straight into leaf records (dump_without_trees); the same
    bytes as through the host trees (RB2_DUMP_VIA_TREES=1), and the index still answers rank queries afterwards"""
    L = hostlib
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.mr_restore_runs.restype = C.c_void_p; L.mr_restore_runs.argtypes = [C.c_void_p]
    L.mr_rank2a.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.mr_dump.argtypes = [C.c_void_p, C.c_void_p]
    L.mr_destroy.argtypes = [C.c_void_p]
    text = H.reads_to_text(H.splitmix_bases(4000, 101, 11)) + b"A" * 90 + b"\n" + (b"C" * 70 + b"\n") * 300
    for so, extra in (("", []), ("s", ["-l", "64", "-n", "6"])):
        f = tmp_path / ("in%s.fmr" % so)
        f.write_bytes(cli(["-LRb" + so, "-m0"] + extra, text))
        outs = {}
        for mode in ("direct", "trees"):
            if mode == "trees":
                os.environ["RB2_DUMP_VIA_TREES"] = "1"
            try:
                fp = libc.fopen(str(f).encode(), b"rb")
                mr = L.mr_restore_runs(fp)
                libc.fclose(fp)
                g = tmp_path / ("%s%s.fmr" % (mode, so))
                fo = libc.fopen(str(g).encode(), b"wb")
                L.mr_dump(mr, fo)
                libc.fclose(fo)
                cx = (C.c_int64 * 6)()
                L.mr_rank2a(mr, 5000, -1, cx, None)
                outs[mode] = (g.read_bytes(), list(cx))
                L.mr_destroy(mr)
            finally:
                os.environ.pop("RB2_DUMP_VIA_TREES", None)
        assert outs["direct"] == outs["trees"]
        assert cli(["-m0", "-i", str(tmp_path / ("direct%s.fmr" % so))], b"") == cli(["-m0", "-i", str(f)], b"")


class _MrItr(C.Structure):
    """mritr_t (include/mrope.h): the caller owns it, the library only sees a pointer"""
    _fields_ = [("r", C.c_void_p), ("a", C.c_int), ("to_free", C.c_int),
                ("rope", C.c_void_p), ("pa", C.c_void_p * 80), ("ia", C.c_int * 80), ("d", C.c_int)]


def _walk_blocks(L, mr, to_free):
    L.mr_itr_first.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.mr_itr_next_block.restype = C.c_void_p; L.mr_itr_next_block.argtypes = [C.c_void_p]
    it = _MrItr()
    L.mr_itr_first(mr, C.byref(it), to_free)
    blocks = []
    while True:
        p = L.mr_itr_next_block(C.byref(it))
        if not p:
            break
        n = C.cast(p, C.POINTER(C.c_uint16))[0]
        blocks.append(C.string_at(p + 2, n))
    return blocks


@pytest.mark.parametrize("so,extra,to_free", [("", [], 0), ("s", ["-l", "64", "-n", "6"], 0), ("r", ["-l", "40", "-n", "4"], 1), ("", [], 1)])
def test_block_iterator_of_restored_run_bytes_builds_no_trees(hostlib, tmp_path, so, extra, to_free):
    """mr_itr_first / mr_itr_next_block (mrope.c:111-130) on an index that only exists as run bytes: the leaf blocks are cut from
    the run stream -- the same blocks, in the same order, as the walk over the bulk-loaded trees (RB2_ITR_VIA_TREES=1) -- and
    the host trees are never built (mr_host_resident); an empty rope still yields its one empty leaf"""
    L = hostlib
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p; libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    L.mr_restore_runs.restype = C.c_void_p; L.mr_restore_runs.argtypes = [C.c_void_p]
    L.mr_host_resident.argtypes = [C.c_void_p]
    L.mr_destroy.argtypes = [C.c_void_p]
    text = H.reads_to_text(H.splitmix_bases(3000, 101, 23)) + b"A" * 90 + b"\n" + (b"G" * 70 + b"\n") * 200   # no N: rope 5 stays empty
    f = tmp_path / "in.fmr"
    f.write_bytes(cli(["-LRb" + so, "-m0"] + extra, text))
    got = {}
    for mode in ("direct", "trees", "direct_no_lookahead"):
        env = {"trees": "RB2_ITR_VIA_TREES", "direct_no_lookahead": "RB2_ITR_NO_LOOKAHEAD"}.get(mode)
        if env:
            os.environ[env] = "1"
        try:
            fp = libc.fopen(str(f).encode(), b"rb")
            mr = L.mr_restore_runs(fp)
            libc.fclose(fp)
            assert L.mr_host_resident(mr) == 0
            got[mode] = _walk_blocks(L, mr, to_free)
            assert L.mr_host_resident(mr) == (1 if mode == "trees" else 0)
            if not to_free and mode == "direct":                # a second walk, and one that is left half-way
                assert _walk_blocks(L, mr, 0) == got[mode]
                it = _MrItr()
                L.mr_itr_first(mr, C.byref(it), 0)
                assert L.mr_itr_next_block(C.byref(it))
            L.mr_destroy(mr)
        finally:
            if env:
                os.environ.pop(env, None)
    assert got["direct"] == got["trees"] == got["direct_no_lookahead"]
    assert len(got["direct"]) > 12 and b"" in got["direct"]
