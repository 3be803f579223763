# End-of-round evidence, collected on the GPU box in one call:  bash tools/collect_profiles.sh <tag>
# writes gpurun_out/<tag>_*; the PMC table is also put under profiles/ of the box's copy so that the bench line that
# follows reads its roofline.traffic from the same build.
tag=${1:-r2}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --profile-steps 0 --no-h2d-leg"
rm -rf /tmp/pf /tmp/pw /tmp/ks
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $BENCH --steps 3 --warmup 2 > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $BENCH --steps 3 --warmup 2 > /tmp/pw.log 2>&1
f=$(find /tmp/pf -name "*counter_collection.csv" | head -1); w=$(find /tmp/pw -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_traffic.py $f $w $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.txt 2>&1
cp $O/${tag}_pmc_traffic.json $R/profiles/${tag}_pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- $BENCH --steps 10 --warmup 3 > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/${tag}_rocprof_kernel_stats_bench.csv
python $R/tools/trace_timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > $O/${tag}_step_timeline.txt 2>&1
cd $R
python tools/step_phases.py > $O/${tag}_step_phases.txt 2>&1
python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err
python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline > $O/${tag}_bench_n1_fp8.json 2> $O/${tag}_bench_n1_fp8.err
python bench.py --steps 20 --warmup 5 --batch 64 --vis 100 --tokens 40 --no-cpu-baseline > $O/${tag}_bench_n1_longctx.json 2> $O/${tag}_bench_n1_longctx.err
tail -c 1500 $O/${tag}_bench_n1.json
