#!/usr/bin/env python3
"""Where the instructions of a kernel are: VALU / SALU / LDS / VMEM instruction counts per SOURCE LINE, from the device assembly with line tables.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only --cuda-device-only -S -Iinclude -Iropebwt2_amd/csrc \
          -o /tmp/eng.s ropebwt2_amd/csrc/rb2_engine.hip
    python tools/line_valu.py /tmp/eng.s _ZN3rb27k_mergeILb0EjEE [first_label]

Static counts (a loop body counts once, both sides of a branch count): read them next to the SQ counters of the same kernel
(tools/collect_sq_bench.sh: instructions per wave as executed).  first_label: count from that basic block on (e.g. the full-window path
of k_merge).  Lines are the INNERMOST inlined location; file numbers are the .file numbers of the assembly.
"""
import collections
import re
import sys


def main():
    path, sym = sys.argv[1], sys.argv[2]
    first = sys.argv[3] if len(sys.argv) > 3 else None
    files, body, on = {}, [], False
    for l in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
        if l.startswith(sym):
            on = True
        elif on and l.startswith(".Lfunc_end"):
            break
        if on:
            body.append(l.rstrip("\n"))
    if first:
        idx = [i for i, l in enumerate(body) if l.startswith(first + ":")]
        body = body[idx[0]:] if idx else body
    cur, cnt = None, collections.OrderedDict()
    for l in body:
        t = l.strip()
        m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        k = "VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else None
        if k:
            cnt.setdefault(cur, collections.Counter())[k] += 1
    tot = collections.Counter()
    for (f, ln), c in sorted(cnt.items(), key=lambda kv: -kv[1]["VALU"]) if cnt else []:
        tot.update(c)
        print("%-28s %5d  VALU %4d  SALU %4d  LDS %3d  VMEM %3d" % (files.get(f, str(f)).split("/")[-1], ln, c["VALU"], c["SALU"], c["LDS"], c["VMEM"]))
    print("total", dict(tot))


if __name__ == "__main__":
    main()
