timeout 900 python -m pytest tests/test_multi.py tests/test_leaf_split_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab_libs.sh base - base - 2>&1
RB2_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('vranks8 new', round(d['value'],2), d['config']['multi_stats'])"
RB2_HIP_LIB=$PWD/ropebwt2_amd/lib/librb2hip_base.so RB2_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('vranks8 base', round(d['value'],2), d['config']['multi_stats'])"
RB2_BENCH_DEVICES=0,0,0,0,0,0,0,0 python bench.py --mode strong --steps 6 --warmup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('vranks8 new', round(d['value'],2), d['config']['multi_stats'])"
