# one A/B turn on a GPU box for the N-ranks-behind-one-handle path: its tests, then configs[1] built by 8 virtual ranks (strong) with librb2hip_base.so and with the shipped build
timeout 1200 python -m pytest tests/test_multi.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | grep -a -E "passed|failed|rror" | head -5
for lib in - base - base; do
  if [ "$lib" != "-" ]; then export RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_$lib.so; else unset RB2_HIP_LIB; fi
  RB2_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('vranks8 $lib', round(d['value'],2), d['config']['multi_stats'], d['config']['counts_ok'])"
done
unset RB2_HIP_LIB
bash tools/ab_libs.sh base - 2>&1
