#!/bin/bash
# A/B of library builds on the coverage30x job (every round has non-empty intervals): tools/ab_cov.sh <tag|-> ...
for lib in "$@"; do
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  python tools/scale_check.py --reads 30000000 --read-len 101 --order rlo --genome-len 100000000 --batch 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'insert_s', round(d['insert_s'],4), 'Gsym/s', round(d['gsym_per_s'],2), d['counts_ok'])"
  python tools/scale_check.py --reads 30000000 --read-len 101 --order rlo --genome-len 100000000 --batch 1 --profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', '  profiled', round(d['insert_s'],4), {k: round(v,1) for k, v in d['kernels_ms'].items() if v > 0})"
done
