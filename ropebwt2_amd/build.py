"""In-tree build of the native libraries (hipcc cross-compiles gfx950 without a GPU).

    ropebwt2_amd/lib/librb2hip.so   HIP kernels + engine, C ABI of include/rb2_hip.h
    ropebwt2_amd/lib/libropebwt2.so plain-C host layer: mrope/rope/rle API + FMD/FMR writers
    ropebwt2_amd/bin/ropebwt2       CLI with the reference's flags

The .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
BINDIR = os.path.join(PKG, "bin")
INC = os.path.join(ROOT, "include")


def lib_path(name):
    return os.path.join(LIBDIR, name)


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise RuntimeError("build failed: " + " ".join(cmd))
    return r.stdout


def build_hip(force=False):
    os.makedirs(LIBDIR, exist_ok=True)
    out = lib_path("librb2hip.so")
    srcs = [os.path.join(CSRC, f) for f in ("rb2_engine.hip", "rb2_kernels.h", "rb2_merge.h", "rb2_device.h", "rb2_multi.h")] + [os.path.join(INC, "rb2_hip.h")]
    if not force and _newer(out, srcs):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I" + INC, "-I" + CSRC,
          "-Wno-unused-value", "-o", out, srcs[0], "-ldl", "-lpthread"])
    return out


def build_host(force=False):
    """plain-C host layer + CLI (only if their sources exist yet)."""
    host = os.path.join(CSRC, "host")
    if not os.path.isdir(host):
        return None
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    csrcs = sorted(os.path.join(host, f) for f in os.listdir(host) if f.endswith(".c") and f != "main.c")
    hdrs = [os.path.join(INC, f) for f in os.listdir(INC)] + [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h")]
    out = lib_path("libropebwt2.so")
    if force or not _newer(out, csrcs + hdrs + [lib_path("librb2hip.so")]):
        _run(["gcc", "-O2", "-g", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-fPIC", "-shared", "-I" + INC, "-I" + host, "-o", out] + csrcs +
             ["-L" + LIBDIR, "-lrb2hip", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    main_c = os.path.join(host, "main.c")
    exe = os.path.join(BINDIR, "ropebwt2")
    if os.path.exists(main_c) and (force or not _newer(exe, [main_c, out] + hdrs)):
        _run(["gcc", "-O2", "-g", "-std=gnu99", "-Wall", "-Werror=implicit-function-declaration", "-I" + INC, "-I" + host, "-o", exe, main_c,
              "-L" + LIBDIR, "-lropebwt2", "-lrb2hip", "-Wl,-rpath,$ORIGIN/../lib", "-lz", "-lpthread"])
    gen_c = os.path.join(ROOT, "tools", "synth_reads.c")      # synthetic read generator (SURVEY.md 8c stream) for tests / bench
    gen = os.path.join(BINDIR, "synth_reads")
    if os.path.exists(gen_c) and (force or not _newer(gen, [gen_c])):
        _run(["gcc", "-O2", "-std=c99", "-Wall", "-o", gen, gen_c])
    return out


def build_all(force=False):
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", sorted(os.listdir(LIBDIR)))
