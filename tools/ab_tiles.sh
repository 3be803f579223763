run() { echo -n "$*: "; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run --input resident
run --input prefetch
run --input sync
run --batch 64 --vis 100 --tokens 40
