#!/bin/bash
# Run on the GPU box: SQ (shader) counters per kernel for the driver's bench job only (collect_sq.sh also does the long-read job: 85 k dispatches a pass at
# ~15 ms each; tools/collect_sq_long.sh restricts that one to a few kernels).   usage: collect_sq_bench.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_driver}
O=$R/gpurun_out/prof_${TAG}_sq
S=$O/summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
PASS_A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
PASS_B="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
BENCH="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --pmc $PASS_A --kernel-trace -d $O -o benchA --output-format csv -- $BENCH > $O/benchA.log 2>&1
timeout 600 rocprofv3 --pmc $PASS_B --kernel-trace -d $O -o benchB --output-format csv -- $BENCH > $O/benchB.log 2>&1
A=$(ls $O/*benchA*counter_collection.csv $O/*/*benchA*counter_collection.csv 2>/dev/null | head -1)
B=$(ls $O/*benchB*counter_collection.csv $O/*/*benchB*counter_collection.csv 2>/dev/null | head -1)
[ -n "$A$B" ] && python $R/tools/pmc_summary.py $A $B > $S/${TAG}_sq_counters.csv
find $O -maxdepth 2 -type f ! -path "$S/*" ! -name "*.log" -delete 2>/dev/null
head -8 $S/*_sq_counters.csv | cut -c1-400
