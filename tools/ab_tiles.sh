cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > /tmp/ks.log 2>&1
f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'embed_text_bwd' in n or 'embed_image_bwd' in n or 'gather_sum' in n: print('   ', n.split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MaxNs'])/1e3)
PY
cd $GRAFT_REPO_ROOT; python -m pytest tests/test_step_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed"; for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done
