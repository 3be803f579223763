# developer A/B of step-level knobs (same box)
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --no-h2d-leg $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],3))"; }
run base A=1
EXTRA="--opt-early 1" run early A=1
EXTRA="--adamw-wgs 128" run wgs128 A=1
EXTRA="--adamw-wgs 512" run wgs512 A=1
EXTRA="--adamw-wgs 0" run wgs0 A=1
EXTRA="--no-opt-overlap" run noov A=1
EXTRA="" run group12 CRCT_GEMM_GROUP=12
EXTRA="" run group9 CRCT_GEMM_GROUP=9
EXTRA="" run streams1 CRCT_STREAMS=1
EXTRA="" run base2 A=1
