run() { echo -n "$*: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run
run --fuse-zero-grad
run
run --fuse-zero-grad
run --adamw-wgs 512
run --adamw-wgs 128
