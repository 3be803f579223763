cd /tmp && export TMPDIR=/tmp
for nb in 256 128 64 32; do
  rm -rf /tmp/ks; CRCT_EMBED_BWD_BLOCKS=$nb rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > /tmp/ks.log 2>&1
  f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1)
  echo "== embed blocks $nb"; grep -E "embed_text_bwd|embed_image_bwd|gather_sum" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-40,200-400 
  python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'embed_text_bwd' in n or 'embed_image_bwd' in n or 'gather_sum' in n: print('   ', n.split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3)
PY
done
cd $GRAFT_REPO_ROOT; for nb in 256 64; do CRCT_EMBED_BWD_BLOCKS=$nb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$nb', d['ms_per_step'])"; done
