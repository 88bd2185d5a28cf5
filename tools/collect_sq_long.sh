#!/bin/bash
# Run on the GPU box: SQ (shader) counters of the in-place round's kernels in the long-read job (configs[3] at one tenth), restricted to the kernels whose name
# matches $2 (a counter pass costs ~15 ms per instrumented dispatch).   usage: collect_sq_long.sh [tag] [kernel regex] [reads]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_longreads}
KRE=${2:-k_merge_leaf|k_part_sparse}
READS=${3:-1000000}
O=$R/gpurun_out/prof_${TAG}_sq
S=$O/summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
PASS_A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
PASS_B="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
LONG="python $R/tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44"
timeout 900 rocprofv3 --pmc $PASS_A --kernel-trace --kernel-include-regex "$KRE" -d $O -o longA --output-format csv -- $LONG > $O/longA.log 2>&1
timeout 900 rocprofv3 --pmc $PASS_B --kernel-trace --kernel-include-regex "$KRE" -d $O -o longB --output-format csv -- $LONG > $O/longB.log 2>&1
A=$(ls $O/*longA*counter_collection.csv $O/*/*longA*counter_collection.csv 2>/dev/null | head -1)
B=$(ls $O/*longB*counter_collection.csv $O/*/*longB*counter_collection.csv 2>/dev/null | head -1)
[ -n "$A$B" ] && python $R/tools/pmc_summary.py $A $B > $S/${TAG}_sq_counters.csv
find $O -maxdepth 2 -type f ! -path "$S/*" ! -name "*.log" -delete 2>/dev/null
head -5 $S/*_sq_counters.csv | cut -c1-600
