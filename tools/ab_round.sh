timeout 900 python -m pytest tests/test_hip_parity.py tests/test_leaf_split_gpu.py -m gpu -x -q 2>&1 | grep -a -E "passed|failed|rror" | head -5
for lib in base - base - -; do
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  python tools/scale_check.py --reads 1000000 --read-len 10000 --order io --seed 44 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', round(d['insert_s'],3), round(d['gsym_per_s'],3), d['layout'], d['counts_ok'], d['lf_ok'], d['rank_ok'])"
done
