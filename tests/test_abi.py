"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports
every symbol include/rb2_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ropebwt2_amd import build_all, load_hip_lib
    build_all()
    return load_hip_lib()


def declared_symbols(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rb2_hip_\w+)\s*\(", txt)))


def test_exports_every_declared_symbol(lib):
    syms = declared_symbols("rb2_hip.h")
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "librb2hip.so does not export %s" % s
    from ropebwt2_amd.hipbwt import ABI_SYMBOLS
    assert sorted(ABI_SYMBOLS) == syms, "python mirror and header disagree"


def test_layout_constants(lib):
    from ropebwt2_amd import HipBwt
    lay = HipBwt.layout()
    assert lay["leaf_syms"] % 64 == 0 and lay["tile_leaves"] >= 1 and lay["string_tile"] % 64 == 0


def test_kernel_names(lib):
    from ropebwt2_amd import K_NAMES
    for i, n in enumerate(K_NAMES):
        assert lib.rb2_hip_kernel_name(i).decode() == n


def test_code_object_targets_gfx950():
    from ropebwt2_amd.build import lib_path
    data = open(lib_path("librb2hip.so"), "rb").read()
    assert b"gfx950" in data
    for k in (b"k_merge", b"k_prep", b"k_advance", b"k_sym"):
        assert k in data


def test_no_gpu_fails_loudly(lib):
    """Without a device the engine must abort with a message -- never fall back to a CPU path."""
    if lib.rb2_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from ropebwt2_amd.hipbwt import load_hip_lib\n"
            "L = load_hip_lib(); L.rb2_hip_create(0, 0)\n") % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0
    assert b"no usable HIP device" in p.stderr
    from ropebwt2_amd import HipBwt
    with pytest.raises(RuntimeError):
        HipBwt(0)


def test_product_never_touches_oracle():
    """oracle/ is test infrastructure: nothing under ropebwt2_amd/ or include/ may reference it."""
    bad = []
    for base in ("ropebwt2_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"liboracle|bcr_oracle|oracle/_ref|orc_insert", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_fatal_handler_is_called_before_abort(lib):
    """rb2_hip_set_fatal_handler: a host program is told (with the message) before the engine gives up, and may leave its own way"""
    if lib.rb2_hip_device_count() > 0:
        code_tail = "L.rb2_hip_create(99, 0)\n"                  # no such device
    else:
        code_tail = "L.rb2_hip_create(0, 0)\n"
    code = ("import sys, os, ctypes as C; sys.path.insert(0, %r)\n"
            "from ropebwt2_amd.hipbwt import load_hip_lib\n"
            "L = load_hip_lib()\n"
            "CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)\n"
            "def h(user, msg):\n"
            "    sys.stdout.write('handler: ' + msg.decode()); sys.stdout.flush(); os._exit(7)\n"
            "cb = CB(h)\n"
            "L.rb2_hip_set_fatal_handler(cb, None)\n") % ROOT + code_tail
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 7, (p.returncode, p.stderr.decode()[-300:])
    assert b"handler: [rb2_hip] no usable HIP device" in p.stdout
