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
