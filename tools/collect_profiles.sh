#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM traffic counters for k_merge.
# Outputs land in gpurun_out/prof/, to be summarised into profiles/ by tools/rocprof_summary.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. plain bench line (with the CPU baseline leg)
timeout 900 python $R/bench.py > $O/bench.json 2> $O/bench.err
# 2. kernel trace + stats of the same command (no baseline leg)
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o ktrace -- python $R/bench.py --no-cpu-baseline > $O/ktrace.log 2>&1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits), kernel-trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch --output-format csv -- python $R/bench.py --no-cpu-baseline --warmup 0 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write --output-format csv -- python $R/bench.py --no-cpu-baseline --warmup 0 > $O/write.log 2>&1
ls -la $O | head -30
tail -1 $O/bench.json | cut -c1-600
