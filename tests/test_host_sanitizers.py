"""The plain-C host layer (csrc/host/*.c) under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 suggests
exactly this for the reference's tree code).  CPU only: the binary is linked against aborting stubs of rb2_hip_*, and only
runs paths that never reach the GPU (-m0 = mr_insert1, restore/dump, the .fmd / .fmr / text writers, the read filters)."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

import helpers as H

HOST = os.path.join(H.ROOT, "ropebwt2_amd", "csrc", "host")
INC = os.path.join(H.ROOT, "include")


@pytest.fixture(scope="module")
def san_cli(tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".c")]
    objs = []
    flags = ["-O1", "-g", "-std=gnu99", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I" + INC, "-I" + HOST]
    for s in srcs:
        o = str(d / (os.path.basename(s) + ".o"))
        subprocess.run(["gcc"] + flags + ["-c", "-o", o, s], check=True)
        objs.append(o)
    und = subprocess.run(["nm", "-u"] + objs, stdout=subprocess.PIPE, check=True).stdout.decode()
    names = sorted(set(re.findall(r"\bU (rb2_hip_\w+)", und)))
    stub = d / "stubs.c"
    stub.write_text("#include <stdlib.h>\n#include <stdio.h>\n" + "".join(
        "void %s(void) { fprintf(stderr, \"stub %s called\\n\"); abort(); }\n" % (n, n) for n in names))
    exe = str(d / "ropebwt2_san")
    p = subprocess.run(["gcc"] + flags + ["-o", exe] + objs + [str(stub), "-lz", "-lpthread"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if p.returncode != 0:
        pytest.skip("sanitizer runtime not available here: " + p.stdout.decode()[-300:])
    return exe


def run(exe, flags, data, env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    e.update(env or {})
    p = subprocess.run([exe] + flags + ["-"], input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-3000:]
    return p.stdout


def test_sanitized_host_layer_m0(san_cli, golden, tmp_path):
    kat = golden["kat_input"].encode()
    for flag in ("-LR", "-LRs", "-LRr", "-L", "-Lr", "-LRN", "-LRT"):
        assert run(san_cli, [flag, "-m0"], kat).decode().strip() == golden["kat"][flag]
    assert run(san_cli, ["-LRd", "-m0"], kat).hex() == golden["kat_fmd_hex"]
    g = golden["sets"]["10k_x_101"]
    text = H.reads_to_text(H.splitmix_bases(3000, g["read_len"], g["seed"]))
    a = run(san_cli, ["-LRsd", "-m0"], text)
    # small leaves / buckets: splits at every level; the parallel .fmd writer with tiny segments and several threads
    b = run(san_cli, ["-LRsd", "-m0", "-l", "32", "-n", "4"], text, env={"RB2_FMD_SEGMENT": "300", "RB2_FMD_THREADS": "3"})
    assert a == b
    # the same streamed to a regular file by the writer thread while the stitcher works (tiny write steps)
    out = tmp_path / "streamed.fmd"
    run(san_cli, ["-LRsd", "-m0", "-o", str(out)], text, env={"RB2_FMD_SEGMENT": "300", "RB2_FMD_THREADS": "3", "RB2_FMD_OUT_STEP": "16"})
    assert out.read_bytes() == a
    fmr = tmp_path / "x.fmr"
    fmr.write_bytes(run(san_cli, ["-LRsb", "-m0", "-l", "64", "-n", "6"], text))
    more = H.reads_to_text(H.splitmix_bases(500, 60, 9))
    c = run(san_cli, ["-LRd", "-m0", "-i", str(fmr)], more)
    if H.have_ref():
        assert c == H.run_ref(["-LRd", "-m0", "-i", str(fmr)], more)


def test_sanitized_reader_and_filters(san_cli):
    from test_host_layer import make_fastx
    fq, fa = make_fastx()
    for data in (fq, fa, b"", b">", b"@a\nACG\n+", b">x y\nAC GT\r\nA\tC\n>z\n\r\nAC\n", b"ACGT\r\n\r\nNNNN\n"):
        for flags in (["-m0", "-d"], ["-m0", "-q", "20"], ["-m0", "-N", "-r"], ["-m0", "-C", "-s"], ["-m0", "-F"]):
            run(san_cli, flags, data)


def test_sanitized_parallel_line_reader(san_cli, tmp_path):
    """the threaded -L reader (batches dumped to a file instead of being inserted) under ASan/UBSan"""
    text = H.reads_to_text(H.splitmix_bases(5000, 101, 3)) + b"ACGTNNAC\r\n\nAC"
    dump = tmp_path / "b.bin"
    for chunk in ("100", "4096", "1000000"):
        if dump.exists():
            dump.unlink()
        run(san_cli, ["-L", "-m50k", "-x", "3"], text, env={"RB2_DUMP_BATCHES": str(dump), "RB2_PARSE_THREADS": "4", "RB2_PARSE_CHUNK": chunk})
        assert dump.stat().st_size > len(text)


def test_sanitized_fmr_written_by_six_threads(san_cli, tmp_path):
    """mr_dump to a regular file (six pwrite threads, rope_dump_size / rope_dump_at) under ASan/UBSan, against the piped bytes"""
    text = H.reads_to_text(H.splitmix_bases(2000, 80, 5))
    for extra in ([], ["-l", "64", "-n", "6"]):
        piped = run(san_cli, ["-LRsb", "-m0"] + extra, text)
        f = tmp_path / "p.fmr"
        run(san_cli, ["-LRsb", "-m0", "-o", str(f)] + extra, text, env={"RB2_DUMP_THREADS": "5"})
        assert f.read_bytes() == piped


def _noncanonical_fmr(pairs_per_leaf=160, leaves=60):
    """an .fmr whose leaves hold NON-canonical run streams: a 2-byte run of 255 x followed by a 1-byte run of 1 x (together a
    4-byte run of 256: the canonical form is LONGER than the input), and 4-byte + 1-byte pairs that cross 2^19 (-> 8 bytes)"""
    import struct

    def enc(c, l):
        if l < 16:
            return bytes([l << 3 | c])
        n = 2 if l < 256 else 4 if l < (1 << 19) else 8
        tail = []
        for _ in range(n - 1):
            tail.append(0x80 | (l & 0x3f)); l >>= 6
        return bytes([{2: 0xC0, 4: 0xE0, 8: 0xF0}[n] | l << 3 | c]) + bytes(reversed(tail))
    out = b"RB\x02\x00"
    for rope in range(6):
        out += struct.pack("<ii", 64, 512)
        out += struct.pack("<Bh", 1, leaves)
        for lf in range(leaves):
            body, cnt = b"", [0] * 6
            for k in range(pairs_per_leaf):
                c = 1 + (k + lf + rope) % 4
                if k % 40 == 7 and len(body) < 480:
                    body += enc(c, (1 << 19) - 1) + enc(c, 1); cnt[c] += 1 << 19
                else:
                    body += enc(c, 255) + enc(c, 1); cnt[c] += 256
                if len(body) > 500:
                    break
            out += struct.pack("<6q", *cnt) + struct.pack("<H", len(body)) + body
    return out


def test_sanitized_parallel_loader_growing_canonical_form(san_cli, tmp_path):
    """rope_load_runs_mt sized the canonical form of a segment as input + 16 bytes; merging adjacent runs of one
    symbol can grow the stream (255 + 1 -> a 4-byte run).  Reached through `-i file -b` (mr_restore_runs -> mr_sync_host)."""
    f = tmp_path / "nc.fmr"
    f.write_bytes(_noncanonical_fmr())
    seq = run(san_cli, ["-b", "-i", str(f)], b"", env={"RB2_LOAD_THREADS": "1"})
    for thr in ("2", "4", "8"):
        par = run(san_cli, ["-b", "-i", str(f)], b"", env={"RB2_LOAD_THREADS": thr, "RB2_LOAD_MIN_SEG": "500"})
        assert par == seq
        g = tmp_path / ("direct%s.fmr" % thr)                # to a regular file: leaf records straight from the run bytes, no trees (rope_rdump_*)
        run(san_cli, ["-b", "-i", str(f), "-o", str(g)], b"", env={"RB2_LOAD_THREADS": thr, "RB2_LOAD_MIN_SEG": "500"})
        assert g.read_bytes() == seq
    if H.have_ref():
        ref = subprocess.run([H.REF_BIN, "-d", "-i", str(f), "/dev/null"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        ours = run(san_cli, ["-d", "-i", str(f)], b"", env={"RB2_LOAD_THREADS": "4", "RB2_LOAD_MIN_SEG": "500"})
        if ref.returncode == 0:
            assert ours == ref.stdout


def test_sanitized_parallel_fastq_reader(san_cli, tmp_path):
    """the threaded FASTQ reader and its fallback to the sequential reader (raw blocks handed back) under ASan/UBSan"""
    from test_host_layer import _fastq_inputs
    ins = _fastq_inputs()
    dump = tmp_path / "b.bin"
    for name in ("strict", "strict_crlf", "strict_no_final_newline", "mid_multiline", "mid_qual_long", "mid_blank", "truncated_record", "header_only", "empty"):
        for chunk in ("64", "3000", "1000000"):
            if dump.exists():
                dump.unlink()
            run(san_cli, ["-m40k", "-q", "10", "-C"], ins[name], env={"RB2_DUMP_BATCHES": str(dump), "RB2_PARSE_THREADS": "4", "RB2_PARSE_CHUNK": chunk})


def test_sanitized_parallel_fasta_reader(san_cli, tmp_path):
    """the threaded FASTA reader (blocks cut in front of '>' lines, carried-over records longer than a block) and its fallback under ASan/UBSan"""
    from test_host_layer import _fasta_inputs
    ins = _fasta_inputs()
    dump = tmp_path / "b.bin"
    for name in ("wrapped60", "wrapped7_blank_lines", "crlf", "no_final_newline", "header_only", "header_no_newline", "two_headers", "one_record_longer_than_many_blocks",
                 "truncated", "mid_fastq", "mid_cr_inside", "mid_gt_inside_a_line"):
        for chunk in ("64", "3000", "1000000"):
            if dump.exists():
                dump.unlink()
            run(san_cli, ["-m40k", "-C"], ins[name], env={"RB2_DUMP_BATCHES": str(dump), "RB2_PARSE_THREADS": "4", "RB2_PARSE_CHUNK": chunk})


def test_sanitized_line_blocks_read_by_the_workers(san_cli, tmp_path):
    """-L on a named plain file: the workers pread their own blocks (pjob_fill_direct) -- long lines across blocks, empty file, no final newline -- under ASan/UBSan"""
    rng = np.random.RandomState(3)
    lines = ["".join(rng.choice(list("ACGTN"), size=int(rng.choice([0, 1, 50, 101, 3000])))) for _ in range(800)]
    dump = tmp_path / "b.bin"
    for name, data in (("many", "\n".join(lines) + "\n"), ("no_nl", "\n".join(lines)), ("empty", ""), ("long", "ACGT" * 5000 + "\nA\n"), ("k16384", ("A" * 15 + "\n") * 1024)):
        path = tmp_path / (name + ".txt")
        path.write_bytes(data.encode())
        for chunk in ("64", "1000", "16384"):
            if dump.exists():
                dump.unlink()
            e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", RB2_DUMP_BATCHES=str(dump), RB2_PARSE_THREADS="4",
                     RB2_PARSE_CHUNK=chunk, RB2_NO_RESERVE="1")
            p = subprocess.run([san_cli, "-L", "-m30k", "-C", str(path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
            assert p.returncode == 0, p.stderr.decode()[-2000:]
            assert b"ERROR: AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr.decode()[-2000:]
