#!/bin/bash
# Run on the GPU box (via gpurun): the DRIVER's bench command + rocprofv3 kernel stats + HBM traffic counters of the same job.
#   usage: collect_profiles.sh [tag]          outputs in gpurun_out/prof_<tag>/, summaries (ready to commit) in gpurun_out/prof_<tag>/summary/
# The profiled runs use --no-extras --no-cpu-baseline: the timed region (and therefore every kernel launch of the job) is the same,
# only the host-API / CLI / CPU-reference legs that follow it are skipped.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03}
STEPS=${STEPS:-20}
WARM=${WARM:-5}
PMC_STEPS=${PMC_STEPS:-4}   # the counter passes cost ~15 ms per dispatch (11 per round, 102 rounds per step): a few steps are a few minutes; per-launch figures do not depend on it
O=$R/gpurun_out/prof_$TAG
S=$O/summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
# 1. the driver's line (with all legs)
timeout 1200 python $R/bench.py --gpus 1 --steps $STEPS --warmup $WARM > $O/bench.json 2> $O/bench.err
# 2. kernel trace + stats of the same job
timeout 900 rocprofv3 --kernel-trace --stats -d $O -o ktrace -- python $R/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras > $O/ktrace.log 2>&1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limits), kernel-trace only
timeout 1500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O -o fetch --output-format csv -- python $R/bench.py --steps $PMC_STEPS --warmup 0 --no-cpu-baseline --no-extras > $O/fetch.log 2>&1
timeout 1500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O -o write --output-format csv -- python $R/bench.py --steps $PMC_STEPS --warmup 0 --no-cpu-baseline --no-extras > $O/write.log 2>&1
# 4. summaries
cp $O/bench.json $S/${TAG}_bench.json
DB=$(ls $O/*ktrace*results.db $O/*/*ktrace*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $S/${TAG}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras (MI355X)" $((STEPS * 102))
F=$(ls $O/*fetch*counter_collection.csv $O/*/*fetch*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $O/*write*counter_collection.csv $O/*/*write*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/traffic_summary.py $F $W $S/${TAG}_hbm_traffic.csv $S/k_merge_traffic.json "python bench.py --steps $PMC_STEPS --warmup 0 --no-cpu-baseline --no-extras"
# only the summaries travel back (gpurun merges at most 64 MiB)
find $O -maxdepth 2 -type f ! -path "$S/*" ! -name "bench.json" ! -name "*.log" -delete 2>/dev/null
ls -la $O $S | head -40
tail -1 $O/bench.json | cut -c1-1500
