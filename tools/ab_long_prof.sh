#!/bin/bash
# like ab_long.sh, with per-kernel times (hipEvent scopes: the job itself runs slower under them)
READS=${READS:-1000000}
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  env $envs python tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44 --profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$spec', round(d['insert_s'],3), 's', {k: round(v) for k, v in d['kernels_ms'].items()})"
done
