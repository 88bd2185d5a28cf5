#!/bin/bash
# A/B of library builds on one box: bench.py --steps 6 with each ropebwt2_amd/lib/librb2hip_<tag>.so given ("-" = the shipped build)
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', round(d['value'],2), 'k_merge ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), {k: round(v) for k, v in d['kernels_ms'].items() if v > 0})"
done
