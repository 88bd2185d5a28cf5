#!/bin/bash
# A/B of library builds (and engine settings) on one box, long-read shape: scale_check 1 M x 10 kbp with each
# ropebwt2_amd/lib/librb2hip_<tag>.so given ("-" = the shipped build); "tag:ENV=VALUE" also sets an environment variable
READS=${READS:-1000000}
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [ "$spec" != "$lib" ] && envs=${spec#*:}
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  env $envs python tools/scale_check.py --reads $READS --read-len 10000 --order io --seed 44 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$spec', round(d['insert_s'],3), 's', round(d['gsym_per_s'],3), 'Gsym/s', d['layout'])"
done
