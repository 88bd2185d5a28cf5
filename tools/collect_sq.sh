#!/bin/bash
# Run on the GPU box: SQ (shader) counters per kernel for the driver's bench job and for the long-read job -- instruction mix
# (VALU / SALU / LDS / SMEM per wave), busy and wait cycles, LDS bank conflicts.  Counters in their own runs, kernel-trace only.
#   usage: collect_sq.sh [tag]          summaries in gpurun_out/prof_<tag>/summary/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_sq}
O=$R/gpurun_out/prof_$TAG
S=$O/summary
mkdir -p $O $S
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
PASS_A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
PASS_B="SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run() {   # name, counters, command...
	local name=$1 ctr=$2; shift 2
	timeout 1200 rocprofv3 --pmc $ctr --kernel-trace -d $O -o $name --output-format csv -- "$@" > $O/$name.log 2>&1
}
BENCH="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras"
LONG="python $R/tools/scale_check.py --reads 300000 --read-len 10000 --order io --seed 44"
run benchA "$PASS_A" $BENCH
run benchB "$PASS_B" $BENCH
run longA "$PASS_A" $LONG
run longB "$PASS_B" $LONG
for j in bench long; do
	A=$(ls $O/*${j}A*counter_collection.csv $O/*/*${j}A*counter_collection.csv 2>/dev/null | head -1)
	B=$(ls $O/*${j}B*counter_collection.csv $O/*/*${j}B*counter_collection.csv 2>/dev/null | head -1)
	[ -n "$A$B" ] && python $R/tools/pmc_summary.py $A $B > $S/${TAG}_${j}_sq_counters.csv
done
# the raw per-dispatch CSVs are hundreds of MB: only the summaries travel back
find $O -maxdepth 2 -type f ! -path "$S/*" ! -name "sq_counters.txt" -delete 2>/dev/null
head -12 $S/*_sq_counters.csv | cut -c1-400
