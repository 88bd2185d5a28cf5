# one A/B turn on a GPU box: the sparse-layout parity tests of the shipped build, then the long-read job with librb2hip_base.so and with the shipped build

timeout 900 python -m pytest tests/test_hip_parity.py tests/test_leaf_split_gpu.py -m gpu -x -q 2>&1 | tail -3
for lib in base - base -; do
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  python tools/scale_check.py --reads 1000000 --read-len 10000 --order io --seed 44 --profile 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', round(d['insert_s'],3), round(d['gsym_per_s'],3), {k: round(v) for k,v in d['kernels_ms'].items()}, d['layout'], d['counts_ok'], d['lf_ok'], d['rank_ok'])"
done
