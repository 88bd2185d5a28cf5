#!/bin/bash
# 8 virtual ranks on one GPU (PEER transport, the owner map an 8-GPU run uses) building configs[1] (strong): the bench line, then the same
# run under rocprofv3 --kernel-trace: per-kernel totals and how much of the wall time the GPU ran no kernel at all.
#   usage (GPU box): bash tools/vranks_profile.sh [tag]   -> gpurun_out/vranks_<tag>.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04}
O=$R/gpurun_out/vranks_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export RB2_BENCH_DEVICES=${RB2_BENCH_DEVICES:-0,0,0,0,0,0,0,0}
{
echo "# RB2_BENCH_DEVICES=$RB2_BENCH_DEVICES python bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras"
python $R/bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value', round(d['value'],2), 'Gsym/s  ms_per_step', round(d['ms_per_step'],1), d['config']['multi_stats'], {k: round(v) for k,v in d['kernels_ms'].items()})"
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o kt -- python $R/bench.py --mode strong --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $O/kt.log 2>&1
DB=$(ls $O/*kt*results.db $O/*/*kt*results.db 2>/dev/null | head -1)
python - "$DB" <<'P'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
# the timed region: from the first k_sym after the last k_synth to the end
last_synth = max((e for n, s, e in rows if 'k_synth' in n), default=rows[0][1])
rows = [r for r in rows if r[1] >= last_synth]
t0, t1 = rows[0][1], max(e for _, _, e in rows)
busy, cur_s, cur_e = 0, None, None
for n, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = {}
for n, s, e in rows:
    k = n.split('(')[0].replace('void ', '')
    k = k.split('<')[0]
    a = tot.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
print("timed region %.1f ms, some kernel running %.1f ms (%.1f %%), sum of kernel durations %.1f ms" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), sum(v[1] for v in tot.values()) / 1e6))
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-22s %8d launches %9.1f ms  avg %8.1f us" % (k, c, d / 1e6, d / c / 1e3))
P
} > $R/gpurun_out/vranks_$TAG.txt 2>&1
rm -rf $O
cat $R/gpurun_out/vranks_$TAG.txt
