run() { echo -n "$1 $2 $3: "; env $1 $2 $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
run X=0
run CRCT_GEMM_DGRAD_LONGK=15
run CRCT_GEMM_DGRAD_LONGK=15 CRCT_GEMM_FWD_LONGK=15
run CRCT_GEMM_DGRAD_LONGK=15 CRCT_GEMM_FWD_LONGK=10
run CRCT_GEMM_DGRAD=15
run CRCT_GEMM_DGRAD=15 CRCT_GEMM_FWD_LONGK=15
run X=1
run CRCT_GEMM_DGRAD_LONGK=15
