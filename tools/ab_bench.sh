#!/bin/bash
# A/B on the GPU box: the driver's bench job (no extras) with the working tree's library and with ropebwt2_amd/lib/librb2hip_<tag>.so (tools/build_variant.sh),
# alternating, value + per-kernel-group milliseconds per step (hipEvent scopes).   usage: ab_bench.sh <tag> [steps] [repeats]
TAG=$1; STEPS=${2:-9}; REP=${3:-2}
cd ${GRAFT_REPO_ROOT:-/root/repo}
one() { python bench.py --steps $STEPS --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); n=d.get('kernels_ms_steps', d['steps'])
print('$1', round(d['value'],2), {k: round(v/n,2) for k,v in d['kernels_ms'].items() if v})"; }
for i in $(seq $REP); do RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$TAG.so one $TAG; one tree; done
