#!/usr/bin/env python3
"""bench.py -- Gsymbols/s inserted by the gfx950 BCR engine (BASELINE.json metric).

Workload at N = 1: BASELINE.json configs[1] -- 100 M x 101 bp synthetic reads, `-bsR` (RLO, forward
strand), `-m4g`, 1 x MI355X.  `-m4g` cuts the job into three batches (40,844,297 / 40,844,297 /
18,311,406 reads; main.c:136, 238).  One "step" = one pass of the hot path (mr_insert_multi,
/root/reference/mrope.c:258-345) over one of these batches, inserted into the index left by the previous
step of the same job.  `--steps K` cycles through the three batches: step k is batch k % 3 of job k // 3,
and every job starts on an EMPTY index (rb2_hip_reset) -- the job never grows beyond configs[1], whatever K
is.  The default (--steps 3) is exactly one configs[1] job.

Inputs are generated on the device (splitmix64 stream of SURVEY.md 8c, tools/synth_reads.c is the same
stream as text) before the timed region, so `value` is whole-job symbols / wall time with inputs resident
in HBM.  Beside it the line reports
  value_host_api   the same job through rb2_hip_insert_multi on HOST buffers (what mr_insert_multi does:
                   one PCIe crossing per batch, no capacity hint) -- PCIe-inclusive, never `value`
  whole_process    the CLI end to end on the same reads (text in, 6.0 GB .fmd out): main.c:340's number
  cpu_baseline     the real reference (oracle/_ref) on a bounded sample of the same stream

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: ONE index, its 31 sub-ropes sharded over the ranks (rb2_hip_default_owners), per round a sum of the
31x6 count matrix and an exchange of 24-byte string records.  The round loop runs INSIDE librb2hip.so
(rb2_hip_multi_*, csrc/rb2_multi.h) -- Python only hands over one batch at a time:
  under torch.distributed.run  one process per GPU; every process is one rank of an RCCL group
                               (rb2_hip_multi_create_rank; the ncclUniqueId travels through torch.distributed):
                               ncclAllReduce + grouped ncclSend/ncclRecv
  plain `python bench.py --gpus N`  ONE process drives N devices over the PEER transport (peer access + device
                               events, no host synchronisation between rounds); RB2_BENCH_DEVICES=0,0,0,0 lists the
                               devices explicitly (virtual ranks on one GPU)
  --mode weak (default)   the job grows with N: N x 100 M reads in three batches of -m(4N)g, so every GPU
                          keeps ~40.8 M strings per round and 1/N of an N-times larger index ("weak";
                          N = 8 is within a factor 1.5 of BASELINE.json configs[2]'s 1.2 B reads)
  --mode strong           the configs[1] job itself, split N ways ("strong")
  --mode independent      one separate BWT per GPU over disjoint read slices, no collectives ("weak")
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SYMBOL = 50.0        # compulsory HBM traffic per inserted symbol
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8 TB/s spec
GEN = os.path.join(ROOT, "ropebwt2_amd", "bin", "synth_reads")
CLI = os.path.join(ROOT, "ropebwt2_amd", "bin", "ropebwt2")
# the real reference on the FULL configs[1] job, same kind of box (profiles/r01_configs1_cli_vs_reference.json)
CPU_FULL_CONFIG = {"value": 0.0293, "unit": "Gsymbols/s", "insert_s": 348.5, "real_s": 391.4, "threads": 5, "measured_in_round": 5, "constant": True,
                   "age": "a constant: re-measured in round 5 on an MI355X box host (profiles/r05_cpu_full_config.json: 348.5 s of inserts; round 1 had 346.9 s, "
                          "real time 391.4 s) -- the full job takes 6.5 min of host time, more than the default run may spend; `bench.py --cpu-full-config` "
                          "measures it again on the box at hand",
                   "source": "profiles/r05_cpu_full_config.json (oracle/_ref/ropebwt2 -L -R -b -s -m4g on all 100 M reads, MI355X box host)"}


def batch_reads(mem_arg_bytes, read_len):
    m = int(mem_arg_bytes * 0.97) + 1            # main.c:136
    per = read_len + 1
    return -(-m // per)                          # main.c:238: flush once buf.l >= m


def cpu_baseline(read_len, so_flag, sample_reads, budget_s=120):
    """Time the reference CLI (oracle/_ref, built from /root/reference by oracle/Makefile) on a
    bounded sample of the same read stream; falls back to the plain-C port (oracle/liboracle.so)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "ropebwt2")
    ncpu = os.cpu_count() or 1
    if os.path.exists(ref) and os.path.exists(GEN):
        try:
            g = subprocess.Popen([GEN, str(sample_reads), str(read_len), "42"], stdout=subprocess.PIPE)
            cmd = [ref, "-L", "-R", "-b"] + ([so_flag] if so_flag else []) + ["-m4g", "-o", "/dev/null", "-"]
            p = subprocess.run(cmd, stdin=g.stdout,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=budget_s * 4)
            g.wait()
            tot_s, tot_t = 0, 0.0
            for m in re.finditer(r"inserted (\d+) symbols in ([0-9.]+) sec", p.stderr.decode()):
                tot_s += int(m.group(1)); tot_t += float(m.group(2))
            if tot_t > 0:
                return {"value": tot_s / tot_t / 1e9, "unit": "Gsymbols/s", "cores": min(5, ncpu), "kind": "reference",
                        "sample": "%d x %d bp reads of the same splitmix64 stream, ropebwt2 -L -R -b %s -m4g (5 threads: 4 workers + master, mrope.c:287-296); "
                                  "%.1f s insert time; host has %d cores" % (sample_reads, read_len, so_flag, tot_t, ncpu),
                        "full_config": CPU_FULL_CONFIG}
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("[bench] reference baseline failed: %r\n" % (e,))
    # plain-C port (single thread)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    n = min(sample_reads, 50000)
    codes = helpers.splitmix_bases(n, read_len)
    buf = helpers.encode_batch_fixed(codes)
    o = helpers.Oracle({"-s": 1, "-r": 2}.get(so_flag, 0))
    t = time.time(); o.insert_multi(buf); dt = time.time() - t
    return {"value": len(buf) / dt / 1e9, "unit": "Gsymbols/s", "cores": 1, "kind": "port",
            "sample": "%d x %d bp reads, oracle/bcr_oracle.c orc_insert_multi, 1 thread" % (n, read_len),
            "full_config": CPU_FULL_CONFIG}


def host_api_rate(HipBwt, so, dev, bufs_dev, sizes, pinned=False):
    """the same job through the host-buffer entry point (rb2_hip_insert_multi): every batch crosses PCIe inside the
    call and nothing is reserved up front -- what a caller of mr_insert_multi gets (mrope.c:258 takes a host buffer)"""
    import numpy as np
    b = HipBwt(so, dev)
    host = []
    for p, n in zip(bufs_dev, sizes):
        a = np.empty(n, np.uint8)
        b.L.rb2_hip_memcpy(b.h, a.ctypes.data, p, n, 1)
        host.append(a)
    reg = []
    if pinned:                                   # the caller registers its batch buffers once, before the clock (what a long-lived caller -- our CLI -- can do)
        for a in host:
            if b.L.rb2_hip_host_register(a.ctypes.data, a.nbytes) == 0:
                reg.append(a)
        if len(reg) != len(host):
            for a in reg:
                b.L.rb2_hip_host_unregister(a.ctypes.data)
            b.close()
            return None
    b.sync()
    t0 = time.perf_counter()
    for a in host:
        b.insert_multi(a)
    b.sync()
    dt = time.perf_counter() - t0
    ok = int(b.counts().sum()) == sum(sizes)
    for a in reg:
        b.L.rb2_hip_host_unregister(a.ctypes.data)
    b.close()
    if pinned:
        return {"value": sum(sizes) / dt / 1e9, "unit": "Gsymbols/s", "seconds": dt, "counts_ok": ok,
                "what": "the same, the caller's buffers registered with the runtime (rb2_hip_host_register = hipHostRegister) once, before the clock: PCIe-inclusive"}
    # (rb2_hip_prefetch does not help THIS pattern -- batches fired back to back from pageable memory: a copy that runs beside the
    # insert kernels takes twice as long and slows them, tools/prefetch_probe.py; it pays where a batch is assembled over seconds,
    # as in the CLI: profiles/r03_configs2_full_cli_1gpu.txt)
    return {"value": sum(sizes) / dt / 1e9, "unit": "Gsymbols/s", "seconds": dt, "counts_ok": ok,
            "what": "one configs[1] job through rb2_hip_insert_multi on pageable host buffers, no rb2_hip_reserve: PCIe-inclusive"}


def whole_process(reads, read_len, so_flag, batch_gib):
    """CLI wall-clock, text file in -> .fmd file out (main.c:340 'Real time'), plus the CLI's own per-batch insert lines
    (main.c:241).  Input and output live in /dev/shm (as in profiles/r01_configs1_cli_vs_reference.json, where the reference
    needed 391 s for the same job); the text is generated before the clock starts.  Falls back to a pipe from the generator
    into the CLI (generator-bound) when /dev/shm cannot hold the 10 GB text + 6 GB .fmd."""
    if not (os.path.exists(GEN) and os.path.exists(CLI)):
        return None
    flags = ["-LRd" + so_flag.strip("-"), "-m%gg" % batch_gib]
    txt, out = "/dev/shm/rb2_bench_%d.txt" % os.getpid(), "/dev/shm/rb2_bench_%d.fmd" % os.getpid()
    mode = "file"
    fmr = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize < reads * (read_len + 1) * 1.8 + (4 << 30):
            raise OSError("not enough room in /dev/shm")
        with open(txt, "wb") as fp:
            subprocess.run([GEN, str(reads), str(read_len), "42"], stdout=fp, check=True)
        t0 = time.perf_counter()
        p = subprocess.run([CLI] + flags + ["-o", out, txt], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=1800)
        dt = time.perf_counter() - t0
        fmd_bytes = os.path.getsize(out) if os.path.exists(out) else 0
        os.unlink(out)
        # the config's literal flags: -b = the .fmr (ropebwt2's own binary dump: device -> six host B+ trees -> file)
        fflags = ["-LRb" + so_flag.strip("-"), "-m%gg" % batch_gib]
        t1 = time.perf_counter()
        pf = subprocess.run([CLI] + fflags + ["-o", out, txt], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=1800)
        dtf = time.perf_counter() - t1
        ef = pf.stderr.decode()
        mh = re.search(r"written as \.fmr in ([0-9.]+) sec", ef)
        mcf = re.search(r"constructed FM-index in ([0-9.]+) sec", ef)
        if pf.returncode == 0 and os.path.exists(out):
            fmr = {"real_s": dtf, "read_parse_insert_s": float(mcf.group(1)) if mcf else None, "device_to_fmr_file_s": float(mh.group(1)) if mh else None,
                   "fmr_bytes": os.path.getsize(out), "flags": " ".join(fflags)}
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("[bench] whole-process leg: %r; using a pipe from the generator\n" % (e,))
        mode, fmd_bytes = "pipe", None
        t0 = time.perf_counter()
        g = subprocess.Popen([GEN, str(reads), str(read_len), "42"], stdout=subprocess.PIPE)
        p = subprocess.run([CLI] + flags + ["-o", "/dev/null", "-"], stdin=g.stdout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=1800)
        g.wait()
        dt = time.perf_counter() - t0
    finally:
        for f in (txt, out):
            if os.path.exists(f):
                os.unlink(f)
    err = p.stderr.decode()
    ins = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"inserted (\d+) symbols in ([0-9.]+) sec", err)]
    syms = sum(s for s, _ in ins)
    if p.returncode != 0 or syms == 0:
        sys.stderr.write("[bench] whole-process leg failed (rc %d)\n%s" % (p.returncode, err[-400:]))
        return None
    m = re.search(r"Real time: ([0-9.]+) sec", err)
    mc = re.search(r"constructed FM-index in ([0-9.]+) sec", err)
    ms = re.search(r"run-length coded in ([0-9.]+) sec \(\+ ([0-9.]+) sec", err)
    return {"value": syms / dt / 1e9, "unit": "Gsymbols/s", "real_s": dt, "cli_real_s": float(m.group(1)) if m else None,
            "read_parse_insert_s": float(mc.group(1)) if mc else None, "insert_s": sum(t for _, t in ins),
            "insert_gsym_per_s": syms / sum(t for _, t in ins) / 1e9,
            "export_and_encode_s": float(ms.group(1)) + float(ms.group(2)) if ms else None, "fmd_bytes": fmd_bytes, "input": mode,
            "fmr_output": fmr,
            "reference_same_job_real_s": CPU_FULL_CONFIG["real_s"],
            "what": "ropebwt2 %s -o out.fmd reads.txt (%d x %d bp text, %s; text parse + PCIe + insert + export + parallel .fmd encode + write; one process)"
                    % (" ".join(flags), reads, read_len, "both in /dev/shm" if mode == "file" else "piped from synth_reads, output to /dev/null")}


class Single:
    """one engine (HipBwt) behind the small interface the job loop needs"""

    def __init__(self, so, dev):
        from ropebwt2_amd import HipBwt
        self.b = HipBwt(so, dev)
        self.eng = self.b

    def alloc(self, nbytes):
        return [self.b.dev_alloc(nbytes)]

    def free(self, ptrs):
        self.b.dev_free(ptrs[0])

    def synth(self, ptrs, first, n, L, seed, strand=0):
        self.b.synth_reads(ptrs[0], first, n, L, seed=seed, strand=strand)

    def insert(self, ptrs, nbytes):
        self.b.insert_multi_dev(ptrs[0], nbytes)

    def reserve(self, *a):
        self.b.reserve(*a)

    def __getattr__(self, k):                  # sync, reset, counts, close
        return getattr(self.b, k)


class Multi:
    """N ranks behind one handle (MultiBwt): the ranks this process drives each need the batch text on their device"""

    def __init__(self, so, devices, transport, rank=None, nranks=None, nccl_id=None):
        from ropebwt2_amd import MultiBwt
        self.m = MultiBwt(so, devices, transport, rank=rank, nranks=nranks, nccl_id=nccl_id)
        self.devices = list(devices)
        self.engs = [self.m.engine(k) for k in range(self.m.n)]
        self.eng = self.engs[0]
        self.lead = [self.devices.index(d) for d in self.devices]       # first local rank on the same device

    def alloc(self, nbytes):
        own = {k: self.engs[k].dev_alloc(nbytes) for k in set(self.lead)}
        return [own[k] for k in self.lead]

    def free(self, ptrs):
        for k in set(self.lead):
            self.engs[k].dev_free(ptrs[k])

    def synth(self, ptrs, first, n, L, seed, strand=0):
        for k in set(self.lead):
            self.engs[k].synth_reads(ptrs[k], first, n, L, seed=seed, strand=strand)
            self.engs[k].sync()

    def insert(self, ptrs, nbytes):
        self.m.insert_multi_dev(list(ptrs), nbytes)

    def reserve(self, *a):
        self.m.reserve(*a)

    def __getattr__(self, k):
        return getattr(self.m, k)


def src_sha():
    """identifies the kernels a committed PMC summary was taken from"""
    import hashlib
    h = hashlib.sha1()
    for f in ("rb2_merge.h", "rb2_kernels.h", "rb2_device.h"):
        h.update(open(os.path.join(ROOT, "ropebwt2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def measure_traffic(budget_s=240):
    """HBM bytes per k_merge launch, measured NOW: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass,
    MI355X_MICROARCH.md) of one configs[1] job of this same script, kernel-trace only; FETCH_SIZE doubled per the guide's
    gfx950 note.  None when rocprofv3 is missing or a pass fails (the caller then falls back to the committed summary)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix="rb2_pmc_", dir="/tmp")
    res, per_kernel = {}, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
            p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=budget_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                sys.stderr.write("[bench] PMC pass %s failed (rc %d)\n%s\n" % (ctr, p.returncode, p.stderr.decode()[-300:]))
                return None
            tot, n = 0.0, 0
            for r in csv.DictReader(open(files[0])):
                kn = r["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1].strip()
                if kn == "k_merge":
                    tot += float(r["Counter_Value"]); n += 1
                per_kernel.setdefault(kn, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})[ctr] += float(r["Counter_Value"])
            res[ctr] = (tot, n)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("[bench] PMC passes failed: %r\n" % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (fe, n), (wr, n2) = res["FETCH_SIZE"], res["WRITE_SIZE"]
    if n == 0 or n != n2:
        return None
    # every kernel of the job, per ROUND (= per k_merge launch): where the bytes of a round go
    table = {k: round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / n / 1e9, 4) for k, v in per_kernel.items() if k.startswith("k_")}
    table = dict(sorted(table.items(), key=lambda kv: -kv[1]))
    return {"bytes_per_launch": (2 * fe + wr) * 1024 / n, "fetch_bytes_per_launch_corrected": 2 * fe * 1024 / n, "write_bytes_per_launch": wr * 1024 / n,
            "traffic_GB_per_round": table, "all_kernels_GB_per_round": round(sum(table.values()), 4),
            "launches": n, "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 3 --warmup 0 "
                                     "--no-cpu-baseline --no-extras` (one configs[1] job, %d k_merge launches); counters in KiB, FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md)" % n}


def secondary_configs2(so_name="rclo", reads=1_200_000_000, L=101, batch_gib=10.0, name="configs[2] shape"):
    """BASELINE.json configs[2]'s shape on ONE GPU (1.2 B x 101 bp, RCLO, -R, ropebwt2's default -m10g: 12 batches on a growing
    index): how the rate falls as the index grows.  Batches are generated on the device just before they are inserted; only the
    inserts are timed.  Result checked through the count matrix (sums, LF consistency)."""
    from ropebwt2_amd import HipBwt
    so = {"io": 0, "rlo": 1, "rclo": 2}[so_name]
    per_batch = batch_reads(batch_gib * 1024 ** 3, L)
    b = HipBwt(so, 0)
    try:
        total = reads * (L + 1)
        b.reserve(per_batch * (L + 1), per_batch, total)
        buf = b.dev_alloc(per_batch * (L + 1) + 64)
        done, times = 0, []
        while done < reads:
            n = min(per_batch, reads - done)
            b.synth_reads(buf, done, n, L, seed=42 if L == 101 else 44)
            b.sync()
            t0 = time.perf_counter()
            b.insert_multi_dev(buf, n * (L + 1))
            b.sync()
            times.append(time.perf_counter() - t0)
            done += n
        c = b.counts()
        sizes, occ = c.sum(axis=1), c.sum(axis=0)
        ok = int(c.sum()) == total and int(c[:, 0].sum()) == reads and all(int(sizes[a]) == int(occ[a]) for a in range(1, 6))
        st = b.sparse_stats()
        b.dev_free(buf)
    finally:
        b.close()
    return {"value": total / sum(times) / 1e9, "unit": "Gsymbols/s", "insert_s": sum(times), "batch_s": [round(t, 3) for t in times], "counts_ok": bool(ok),
            "layout": st, "what": "%s on 1 GPU: %d x %d bp, %s, forward strand, -m%gg (%d batches, %.1f G symbols), inputs generated on the device, inserts timed"
                                  % (name, reads, L, so_name.upper(), batch_gib, len(times), total / 1e9)}


def secondary_configs3(reads=1_000_000, L=10_000, batch_gib=10.0):
    """BASELINE.json configs[3]'s shape at one tenth on ONE GPU (1 M x 10 kbp, input order, forward strand, -m10g: one batch of
    10,001 rounds of a million symbols each -- the latency-bound regime; the full config is ten such batches on a growing index,
    profiles/r03_configs3_full_1gpu.json).  Inputs generated on the device, the insert timed; checked through the count matrix."""
    from ropebwt2_amd import HipBwt
    b = HipBwt(0, 0)
    try:
        total = reads * (L + 1)
        b.reserve(total, reads, total)
        buf = b.dev_alloc(total + 64)
        b.synth_reads(buf, 0, reads, L, seed=44)
        b.sync()
        t0 = time.perf_counter()
        b.insert_multi_dev(buf, total)
        b.sync()
        dt = time.perf_counter() - t0
        c = b.counts()
        sizes, occ = c.sum(axis=1), c.sum(axis=0)
        ok = int(c.sum()) == total and int(c[:, 0].sum()) == reads and all(int(sizes[a]) == int(occ[a]) for a in range(1, 6))
        st = b.layout_stats()
        b.dev_free(buf)
    finally:
        b.close()
    # roofline of the whole job: the algorithmic bytes of an inserted symbol (50 B) over the wall time against the 8 TB/s peak -- the in-place rounds
    # move a leaf (two or three 128-byte lines in and out), the directory lines around it and 50 B of string state per symbol; there is no single dominant launch
    return {"value": total / dt / 1e9, "unit": "Gsymbols/s", "insert_s": dt, "rounds": L + 1, "us_per_round": dt / (L + 1) * 1e6, "counts_ok": bool(ok), "layout": st,
            "roofline_frac": round(50.0 * total / dt / 8e12, 4), "roofline": {"bound": "hbm", "achieved": round(50.0 * total / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                                              "basis": "50 algorithmic bytes per inserted symbol over the wall time of the job (all kernels, all rounds)"},
            "what": "configs[3] shape at one tenth on 1 GPU: %d x %d bp, input order, forward strand, one -m10g batch (%.1f G symbols), inputs generated on the device, insert timed"
                    % (reads, L, total / 1e9)}


def secondary_coverage(L=101):
    """The compressible regime ropebwt2 exists for: 30 M OVERLAPPING reads (windows of one random 100 Mb genome, 30x; the job of
    golden_large.json coverage30x, whose .fmd md5 from the real reference the test-suite checks through the CLI) on one GPU, RLO,
    -m1g batches.  Reports the rate, the HBM bytes the index holds per symbol (packed bit planes: 0.375 + directory, whatever the
    data) and, beside it, what the real reference needed for the same job (tools/ref_footprint.py: its run-length B+ trees)."""
    from ropebwt2_amd import HipBwt
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_large.json"))).get("coverage30x")
    if g is None:
        return None
    reads, glen = g["n_reads"], g["genome_len"]
    per_batch = batch_reads(1.0 * 1024 ** 3, L)
    b = HipBwt(1, 0)
    try:
        total = reads * (L + 1)
        b.reserve(per_batch * (L + 1), per_batch, total)
        buf = b.dev_alloc(per_batch * (L + 1) + 64)
        done, dt = 0, 0.0
        while done < reads:
            n = min(per_batch, reads - done)
            b.synth_reads(buf, done, n, L, seed=g["seed"], genome_len=glen)
            b.sync()
            t0 = time.perf_counter()
            b.insert_multi_dev(buf, n * (L + 1))
            b.sync()
            dt += time.perf_counter() - t0
            done += n
        c = b.counts()
        ok = int(c.sum()) == total and int(c[:, 0].sum()) == reads
        runs = sum(int(b.L.rb2_hip_rope_bytes(b.h, k)) for k in range(6))            # one byte per run of <= 15 symbols: what the exported stream costs
        b.dev_free(buf)
    finally:
        b.close()
    lay = HipBwt.layout()
    leaf_b = lay["leaf_syms"] * 3 // 8
    held = (leaf_b + 32.0 + 32.0 / 32) / lay["leaf_syms"]          # leaf + LeafMeta x 2 + superblock record share, per symbol, one pool side
    ref = g.get("reference_run")
    return {"value": total / dt / 1e9, "unit": "Gsymbols/s", "insert_s": dt, "counts_ok": bool(ok), "symbols": total,
            "hbm_bytes_per_symbol": {"one_pool_side": round(held, 4), "both_sides_of_a_dense_round": round(2 * held, 4)},
            "run_bytes_per_symbol_exported": round(runs / total, 4),
            "reference": ref, "footprint_ratio_vs_reference_rss": round(2 * held / ref["rss_bytes_per_symbol"], 2) if ref else None,
            "what": "coverage30x: %d x %d bp windows of one random %d bp genome (30x), RLO, -m1g batches on 1 GPU; inputs generated on the device, inserts timed" % (reads, L, glen)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads in the whole job (configs[1]: 100 M)")
    ap.add_argument("--read-len", type=int, default=101)
    ap.add_argument("--batch", type=float, default=4.0, help="-m in GiB")
    ap.add_argument("--order", default="rlo", choices=["io", "rlo", "rclo"])
    ap.add_argument("--mode", default="weak", choices=["weak", "strong", "independent"], help="what N > 1 ranks do (see module docstring)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the value_host_api / whole_process / secondary / PMC legs (profiling runs)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the configs[2]-shape leg")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic in-run (use the committed summary if it matches the sources)")
    ap.add_argument("--cpu-sample-reads", type=int, default=10_000_000, help="reads of the reference's bounded sample (10 M: ~25 s of inserts)")
    ap.add_argument("--configs3-full", action="store_true", help="also run BASELINE.json configs[3] at FULL size on one GPU (10 M x 10 kbp, ten -m10g batches, ~40 s): secondary.configs3_full_1gpu")
    ap.add_argument("--cpu-full-config", action="store_true", help="re-measure cpu_baseline.full_config: the reference on ALL reads of configs[1] (~6.5 min of host time)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        sys.stderr.write("[bench] WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE\n" % (world, args.gpus))
    # one process may drive several devices itself (no launcher): --gpus N, or an explicit device list (virtual ranks)
    local_devices = None
    if world == 1:
        if os.environ.get("RB2_BENCH_DEVICES"):
            local_devices = [int(x) for x in os.environ["RB2_BENCH_DEVICES"].split(",")]
        elif args.gpus > 1:
            local_devices = list(range(args.gpus))
        if local_devices is not None and len(local_devices) < 2:
            local_devices = None
    n_ranks = world if world > 1 else (len(local_devices) if local_devices else 1)
    n_gpus = world if world > 1 else (len(set(local_devices)) if local_devices else 1)

    import torch
    dist = None
    # RB2_BENCH_FORCE_MULTI=1: run the multi-GPU code path (process group, ncclUniqueId through torch.distributed, one rank of an RCCL
    # group per process, the round loop inside librb2hip.so) even with ONE process -- how that path is exercised on a one-GPU box
    force_multi = bool(os.environ.get("RB2_BENCH_FORCE_MULTI")) and "RANK" in os.environ
    if world > 1 or force_multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from ropebwt2_amd import HipBwt, MultiBwt, build_all
    if rank == 0:
        build_all()
    if dist is not None:
        dist.barrier()
    sharded = (n_ranks > 1 or force_multi) and args.mode in ("weak", "strong")
    if sharded and args.mode == "weak":                 # per-GPU work fixed: N times the reads, N times the batch
        args.reads *= n_gpus
        args.batch *= n_gpus
    so = {"io": 0, "rlo": 1, "rclo": 2}[args.order]
    so_flag = {"io": "", "rlo": "-s", "rclo": "-r"}[args.order]
    L = args.read_len
    per_batch = batch_reads(args.batch * 1024 ** 3, L)
    dev = local_rank if world > 1 else 0
    driver = "single engine"

    def fresh_nccl_id():
        """every RCCL communicator needs an id of its own (the warm-up handle and the measured one are two communicators):
        rank 0 makes one, torch.distributed hands it round"""
        box = [MultiBwt.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def make(so_, dev_):
        nonlocal driver
        if not sharded:
            return Single(so_, dev_)
        if world > 1 or force_multi:
            driver = "librb2hip round loop, one process per GPU, RCCL C API (ncclAllReduce + grouped ncclSend/ncclRecv)"
            return Multi(so_, [dev_], "rccl", rank=rank, nranks=world, nccl_id=fresh_nccl_id())
        tr = os.environ.get("RB2_BENCH_TRANSPORT", "peer")
        driver = "librb2hip round loop, one process, %d ranks on devices %s, %s transport" % (len(local_devices), local_devices, tr.upper())
        return Multi(so_, local_devices, tr)

    # ---- warm-up: same kernels (and collectives) on a small scratch index (untimed)
    for _ in range(max(0, args.warmup)):
        w = make(so, dev)
        n = 200_000
        p = w.alloc(n * (L + 1))
        for i in range(2):
            w.synth(p, i * n, n, L, 7)
            w.sync()
            w.insert(p, n * (L + 1))
        w.free(p)
        w.close()

    # ---- the job: its -m batches (sharded: the same batches on every rank, one index; independent: this rank's own
    # ---- slice of the read stream, its own index).  Step k = batch k % nb of job k // nb; a job starts on an empty index.
    bwt = make(so, dev)
    # hipEvents inside the timed region: around the k_merge launches only (level 2: the roofline's `achieved` needs their duration, measured live);
    # the other kernel groups are timed in a pass of their own behind the clock (sixteen events per round cost the timed job 1-2 %).
    # RB2_BENCH_FULLPROF=1: every group inside the timed region, as in rounds 1-5.
    full_prof = os.environ.get("RB2_BENCH_FULLPROF", "0") == "1"
    bwt.eng.profile(1 if full_prof else 2)
    first = 0 if sharded or world == 1 else rank * args.reads
    job, done = [], 0
    while done < args.reads:
        n = min(per_batch, args.reads - done)
        job.append((first + done, n))
        done += n
    nb = len(job)
    bufs = []
    for (f, n) in job:                            # inputs resident in HBM before the clock starts
        p = bwt.alloc(n * (L + 1))
        bwt.synth(p, f, n, L, 42)
        bufs.append(p)
    sizes = [n * (L + 1) for _, n in job]
    # capacity hint (rb2_hip_reserve): the job's size is known up front, as it is to `ropebwt2 -m`; without it
    # the engine grows its buffers batch by batch (hipMalloc + copy + hipFree inside the timed region)
    tot_syms = sum(sizes)
    bwt.reserve(max(sizes), max(n for _, n in job), tot_syms)              # (a Multi handle divides by its active ranks itself)
    bwt.sync()
    bwt.eng.profile_get(reset=True)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        j = k % nb
        if j == 0 and k > 0:
            bwt.reset()                           # next job: empty index, same buffers
        bwt.insert(bufs[j], sizes[j])
    bwt.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    symbols = sum(sizes[k % nb] for k in range(args.steps))
    last = (args.steps - 1) % nb + 1 if args.steps else 0      # batches in the index at the end (last job, maybe partial)
    prof = bwt.eng.profile_get()
    counts = bwt.counts()
    ok_counts = int(counts.sum()) == sum(sizes[:last]) and int(counts[:, 0].sum()) == sum(n for _, n in job[:last])
    prof_all, prof_all_steps = prof, args.steps
    if not full_prof:                                          # the per-group breakdown: one more job, every group timed, behind the clock
        bwt.eng.profile(1)
        bwt.eng.profile_get(reset=True)
        bwt.reset()
        for j in range(nb):
            bwt.insert(bufs[j], sizes[j])
        bwt.sync()
        prof_all, prof_all_steps = bwt.eng.profile_get(), nb
    mstats = bwt.stats() if isinstance(bwt, Multi) else None
    host_api = host_api_pinned = None
    if rank == 0 and n_ranks == 1 and not args.no_extras:
        bwt.reset()
        host_api = host_api_rate(HipBwt, so, dev, [p[0] for p in bufs], sizes)
        try:
            host_api_pinned = host_api_rate(HipBwt, so, dev, [p[0] for p in bufs], sizes, pinned=True)
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("[bench] pinned host-API leg failed: %r\n" % (e,))
    for p in bufs:
        bwt.free(p)
    bwt.close()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total_symbols = symbols if (sharded or n_ranks == 1) else symbols * n_gpus
    active, owners = 1, None
    if sharded:
        owners = MultiBwt.default_owners(n_ranks)
        active = len(set(owners))
    mk = prof["k_merge"]
    units = mk["units"] / (active if sharded else 1)
    ach = ALG_BYTES_PER_SYMBOL * units / (mk["ms"] * 1e-3) / 1e9 if mk["ms"] > 0 else 0.0
    njobs = -(-args.steps // nb)
    reads_job = sum(n for _, n in job)
    name = "configs[1]" if not (sharded and args.mode == "weak") else "configs[1] x %d (weak scaling of one sharded index)" % n_gpus
    out = {
        "metric": "Gsymbols/s inserted (wall-clock), bit-identical .fmd",
        "value": total_symbols / dt / 1e9,
        "unit": "Gsymbols/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / max(1, args.steps),
        "higher_is_better": True, "scaling": "strong" if (sharded and args.mode == "strong") else "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: %d x %d bp synthetic reads (splitmix64 seed 42), -b%sR -m%gg, %d x MI355X; step = one -m batch "
                               "(%s reads); %d steps = %d job(s) of %d batches, each job on an empty index%s"
                               % (name, reads_job, L, so_flag.strip("-"), args.batch, n_gpus, "/".join(str(n) for _, n in job), args.steps, njobs, nb,
                                  "" if args.steps % nb == 0 else " (the last job stops after batch %d)" % (args.steps % nb)),
                   "reads_per_job": reads_job * (1 if (sharded or n_ranks == 1) else n_gpus), "jobs": args.steps / nb,
                   "symbols": total_symbols, "symbols_per_gpu": symbols // (n_gpus if sharded else 1),
                   "parallelism": "1 GPU" if n_ranks == 1 else
                                  ("31 sub-ropes (b,x) sharded over %d of %d ranks on %d GPU(s) (owner map %s); per round: sum of the 31x6 count matrix + exchange of 24 B string records"
                                   % (active, n_ranks, n_gpus, owners)) if sharded else "independent BWT per GPU (read stream sliced by rank)",
                   "driver": driver, "multi_stats": mstats,
                   "counts_ok": bool(ok_counts)},
        "roofline": {"bound": "hbm", "kernel": "k_merge", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": None,
                     # the pipeline figure: the same algorithmic bytes over the WHOLE round (all kernels, wall clock) instead of over k_merge alone
                     "whole_round_achieved": ALG_BYTES_PER_SYMBOL * total_symbols / dt / 1e9 / max(1, n_gpus),
                     "whole_round_frac": ALG_BYTES_PER_SYMBOL * total_symbols / dt / 1e9 / max(1, n_gpus) / HBM_PEAK_GBS,
                     "avg_launch_ms": mk["ms"] / max(1, mk["launches"]), "launches": mk["launches"],
                     "algorithmic_bytes_per_symbol": ALG_BYTES_PER_SYMBOL,
                     "algorithmic_bytes_per_launch": ALG_BYTES_PER_SYMBOL * units / max(1, mk["launches"]),
                     "note": None if not sharded else "rank 0 only; units per launch approximated by strings / active ranks"},
        "kernels_ms": {k: round(v["ms"], 3) for k, v in prof_all.items()},
        "kernels_ms_steps": prof_all_steps,
        "kernels_ms_note": ("hipEvent scopes of every kernel group over the timed steps" if full_prof else
                            "hipEvent scopes of every kernel group over ONE more job (%d steps) run behind the clock; inside the timed region only the k_merge launches carry events "
                            "(roofline.avg_launch_ms)" % nb),
    }
    # HBM bytes per k_merge launch: measured now (two PMC passes of one configs[1] job -- the per-launch average does not depend on
    # K because every job is the same job); else the committed summary of the same command, but only if it was taken from the
    # kernels that are being benchmarked (sha of the kernel sources)
    is_cfg1 = n_ranks == 1 and args.reads == 100_000_000 and args.batch == 4.0 and args.order == "rlo" and L == 101
    if is_cfg1 and not args.no_extras and not args.no_pmc:
        tr = measure_traffic()
        if tr is not None:
            out["roofline"]["traffic"] = tr["bytes_per_launch"]
            out["roofline"]["traffic_GB_per_round"] = tr.pop("traffic_GB_per_round")
            out["roofline"]["all_kernels_GB_per_round"] = tr.pop("all_kernels_GB_per_round")
            out["roofline"]["traffic_detail"] = tr
    traffic_file = os.path.join(ROOT, "profiles", "k_merge_traffic.json")
    if out["roofline"]["traffic"] is None and os.path.exists(traffic_file) and is_cfg1:
        try:
            tf = json.load(open(traffic_file))
            if tf.get("src_sha") == src_sha():
                out["roofline"]["traffic"] = tf.get("bytes_per_launch")
                out["roofline"]["traffic_detail"] = {"source": tf.get("source"), "src_sha": tf.get("src_sha")}
            else:
                out["roofline"]["traffic_detail"] = {"stale": "profiles/k_merge_traffic.json was taken from other kernel sources (sha %s, now %s): not used"
                                                             % (tf.get("src_sha"), src_sha())}
        except Exception:  # noqa: BLE001
            pass
    # achieved HBM rate per kernel group: the PMC bytes per round (a separate, serialised run of the same job) over the in-run hipEvent time per round
    tgr = out["roofline"].get("traffic_GB_per_round")
    if tgr and mk["launches"]:
        groups = {"k_sym": ["k_sym"], "k_tscan": ["k_tscan1", "k_tscan2", "k_tscan3", "k_tfix", "k_tscan_setup", "k_setup"], "k_prep": ["k_prep"], "k_part": ["k_part"],
                  "k_merge": ["k_merge"], "k_meta": ["k_meta_sb", "k_sbscan3", "k_sbscan2"], "k_advance": ["k_advance"]}
        per = {}
        for g, ks in groups.items():
            gb = sum(tgr.get(k, 0.0) for k in ks)
            ms = prof_all[g]["ms"] / max(1, prof_all["k_merge"]["launches"]) if g in prof_all else 0.0
            if gb > 0 and ms > 0:
                per[g] = {"GB_per_round": round(gb, 4), "ms_per_round": round(ms, 4), "TB_per_s": round(gb / ms, 3), "frac_of_peak": round(gb / ms / (HBM_PEAK_GBS / 1e3), 3)}
        out["roofline"]["per_kernel"] = per
        out["roofline"]["per_kernel_note"] = ("GB_per_round: FETCH x2 + WRITE of the PMC passes (the x2 is calibrated for wide streaming reads, MI355X_MICROARCH.md: it overstates kernels "
                                              "that gather 2-16 byte items); ms_per_round: hipEvent scopes (kernels_ms_note), launch gaps inside a scope included")
    if host_api is not None:
        out["value_host_api"] = host_api
    if host_api_pinned is not None:
        out["value_host_api_pinned"] = host_api_pinned
    if n_ranks == 1 and not args.no_extras:
        wp = whole_process(args.reads, L, so_flag, args.batch)
        if wp is not None:
            out["whole_process"] = wp
        if not args.no_secondary and is_cfg1:
            try:
                out["secondary"] = {"configs2_shape_1gpu": secondary_configs2()}
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("[bench] secondary leg failed: %r\n" % (e,))
            try:
                out.setdefault("secondary", {})["configs3_shape_tenth_1gpu"] = secondary_configs3()
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("[bench] secondary long-read leg failed: %r\n" % (e,))
            if args.configs3_full:
                try:
                    out.setdefault("secondary", {})["configs3_full_1gpu"] = secondary_configs2("io", 10_000_000, 10_000, 10.0, name="configs[3] at full size")
                except Exception as e:  # noqa: BLE001
                    sys.stderr.write("[bench] full configs[3] leg failed: %r\n" % (e,))
            try:
                cv = secondary_coverage()
                if cv is not None:
                    out.setdefault("secondary", {})["coverage30x_1gpu"] = cv
            except Exception as e:  # noqa: BLE001
                sys.stderr.write("[bench] secondary coverage leg failed: %r\n" % (e,))
    if not args.no_cpu_baseline and n_ranks == 1:
        out["cpu_baseline"] = cpu_baseline(L, so_flag, args.cpu_sample_reads)
        if args.cpu_full_config and is_cfg1:
            full = cpu_baseline(L, so_flag, args.reads, budget_s=900)
            if full.get("kind") == "reference":
                out["cpu_baseline"]["full_config"] = {"value": full["value"], "unit": "Gsymbols/s", "threads": 5, "constant": False, "sample": full["sample"], "measured": "in this run"}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
