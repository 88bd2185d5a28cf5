#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats + HBM traffic counters of the long-read job (configs[3] at one tenth: 1 M x 10 kbp).
#   usage: collect_longreads_profile.sh [tag] [reads] [kernel regex]      summaries in gpurun_out/prof_<tag>/summary/
# The counter passes cost ~15 ms per dispatch they instrument (85 k dispatches in this job: 20 min a pass); the third argument restricts them to the kernels
# whose name matches (default: the leaf merges and the descent -- the byte-bound kernels of an in-place round; the other rows of the traffic table stay empty)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_long}
READS=${2:-1000000}
KRE=${3:-k_merge_leaf|k_part_sparse|k_advance}
O=$R/gpurun_out/prof_$TAG
S=$O/summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44"
[ -z "${SKIP_KTRACE:-}" ] && timeout 400 rocprofv3 --kernel-trace --stats -d $O -o ktrace -- $CMD > $O/ktrace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$KRE" -d $O -o fetch --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$KRE" -d $O -o write --output-format csv -- $CMD > $O/write.log 2>&1
DB=$(ls $O/*ktrace*results.db $O/*/*ktrace*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB $S/${TAG}_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44 (MI355X)"
F=$(ls $O/*fetch*counter_collection.csv $O/*/*fetch*counter_collection.csv 2>/dev/null | head -1)
W=$(ls $O/*write*counter_collection.csv $O/*/*write*counter_collection.csv 2>/dev/null | head -1)
[ -n "$F" ] && [ -n "$W" ] && python $R/tools/traffic_summary.py $F $W $S/${TAG}_hbm_traffic.csv $S/${TAG}_k_merge_traffic.json "python tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44"
find $O -maxdepth 2 -type f ! -path "$S/*" ! -name "*.log" -delete 2>/dev/null
tail -2 $O/ktrace.log | cut -c1-400
head -14 $S/${TAG}_kernel_stats.csv
head -12 $S/${TAG}_hbm_traffic.csv
