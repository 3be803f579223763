run() { echo -n "$1: "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -x -q -m gpu -k "not gemm and not wgrad" 2>&1 | grep -E "passed|failed"
run X=1
run CRCT_LN_BWD_NO_COMBINE=1
run X=1
run CRCT_LN_BWD_NO_COMBINE=1
run X=1
run CRCT_LN_BWD_NO_COMBINE=1
