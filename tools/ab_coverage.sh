#!/bin/bash
# A/B on the GPU box, coverage30x (30 M overlapping reads: intervals stay non-empty, groups stay large -- the non-AE tile kernels): bench.py's secondary_coverage()
# with ropebwt2_amd/lib/librb2hip_<tag>.so and with the working tree's library, alternating.   usage: ab_coverage.sh <tag> [repeats]
TAG=$1; REP=${2:-2}
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python -c "
import bench, json
r = bench.secondary_coverage()
print('$1', round(r['value'], 2), 'Gsym/s', round(r['insert_s'], 4), 's', r['counts_ok'])" 2>/dev/null | tail -1; }
for i in $(seq $REP); do RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$TAG.so one $TAG; one tree; done
