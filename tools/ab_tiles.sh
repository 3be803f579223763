run() { echo -n "$1 $2: "; env $1 $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run CRCT_GEMM_GROUP=4
run CRCT_GEMM_GROUP=8
run CRCT_GEMM_GROUP=6
run CRCT_GEMM_GROUP=4 CRCT_GEMM_DGRAD=15
run CRCT_GEMM_GROUP=4 CRCT_GEMM_DGRAD=13
run CRCT_GEMM_GROUP=4 CRCT_GEMM_DGRAD=4
run CRCT_GEMM_GROUP=4 CRCT_GEMM_FWD=15
run CRCT_GEMM_GROUP=4 CRCT_GEMM_WGRAD=4
run CRCT_GEMM_GROUP=4
