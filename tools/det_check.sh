# developer check: is the final loss of N identical runs bit-identical?  usage: bash tools/det_check.sh <runs> [ENV=VAL ...]
n=$1; shift
for i in $(seq 1 $n); do env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-h2d-leg --profile-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', repr(d['config']['final_loss']), round(d['ms_per_step'],3))"; done
