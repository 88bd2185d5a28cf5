#!/bin/bash
# On the GPU box: instruction counts per wave (one --pmc pass) and HBM traffic (two passes) of the dense job's kernels for alternative
# library builds / settings.   usage: ab_counters.sh <tag>:<lib or ->[:ENV=val] ...      summaries in gpurun_out/abc/
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/abc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras"
for spec in "$@"; do
  IFS=: read tag lib envs <<< "$spec"
  ( if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$R/ropebwt2_amd/lib/librb2hip_$lib.so; fi
    [ -n "${envs:-}" ] && export $envs
    for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
      n=${pass%%:*}; c=${pass#*:}
      rm -rf $O/raw; mkdir -p $O/raw
      timeout 900 rocprofv3 --pmc $c --kernel-trace -d $O/raw -o $n --output-format csv -- $BENCH > $O/${tag}_$n.log 2>&1
      f=$(find $O/raw -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && python $R/tools/pmc_summary.py $f | grep "kernel,\|k_merge\|k_advance\|k_prep\|k_sym\|k_part\|k_meta" > $O/${tag}_$n.csv
    done
    rm -rf $O/raw )
done
for f in $O/*_sq.csv $O/*_fetch.csv $O/*_write.csv; do echo "== $f"; cut -c1-260 $f; done
