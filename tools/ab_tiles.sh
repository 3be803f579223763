run() { echo -n "$1: "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run X=1
run CRCT_LN_BWD_BLOCKS=512
run CRCT_LN_BWD_BLOCKS=1024
run CRCT_LN_BWD_BLOCKS=128
run X=1
run CRCT_LN_BWD_BLOCKS=512
