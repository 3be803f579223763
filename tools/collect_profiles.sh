# End-of-round evidence, collected on the GPU box in one call:  bash tools/collect_profiles.sh <tag>
# writes gpurun_out/<tag>_*; the PMC tables are also put under profiles/ of the box's copy so that the bench lines that
# follow read their roofline.traffic from the same build.
tag=${1:-r6}
# optional second argument: space-separated stages (pmc pmcfp8 pmclc pmcpq stats labs bench); default: all
stages=${2:-"pmc pmcfp8 pmclc pmcpq stats labs bench"}
want() { case " $stages " in *" $1 "*) return 0;; esac; return 1; }
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --profile-steps 0 --no-h2d-leg"
# --sustained-s 0: the launch log covers the timed region; tools/pmc_sites.py matches it to the LAST GEMM dispatches of the process
PB="$BENCH --steps 3 --warmup 2 --sustained-s 0"
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/p4 /tmp/p5 /tmp/p6 /tmp/p7 /tmp/ks /tmp/kl
cc() { find $1 -name "*counter_collection.csv" | head -1; }
LC="--batch 64 --vis 100 --tokens 40"
if want pmc; then
# per-site counters of configs[1]: SQ pass, L2 hit / miss pass, fabric read / write passes, instruction-cache / TLB pass
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/p1 -- $PB --launch-log /tmp/log1.json > /tmp/p1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/p2 -- $PB --launch-log /tmp/log2.json > /tmp/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p3 -- $PB --launch-log /tmp/log3.json > /tmp/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p4 -- $PB --launch-log /tmp/log4.json > /tmp/p4.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH --kernel-trace --output-format csv -d /tmp/p5 -- $PB --launch-log /tmp/log5.json > /tmp/p5.log 2>&1
python3 $R/tools/pmc_sites.py /tmp/log1.json $O/${tag}_pmc_sites.json $(cc /tmp/p1) $(cc /tmp/p2) $(cc /tmp/p3) $(cc /tmp/p4) $(cc /tmp/p5) > $O/${tag}_pmc_sites.txt 2>&1
cp $O/${tag}_pmc_sites.json $R/profiles/${tag}_pmc_sites.json
fi
if want pmcfp8; then
# the same for configs[4] (fp8): SQ pass + fabric read / write passes
rm -rf /tmp/q1 /tmp/q3 /tmp/q4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/q1 -- $PB --dtype fp8 --launch-log /tmp/logq1.json > /tmp/q1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/q3 -- $PB --dtype fp8 --launch-log /tmp/logq3.json > /tmp/q3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/q4 -- $PB --dtype fp8 --launch-log /tmp/logq4.json > /tmp/q4.log 2>&1
python3 $R/tools/pmc_sites.py /tmp/logq1.json $O/${tag}_pmc_sites_fp8.json $(cc /tmp/q1) $(cc /tmp/q3) $(cc /tmp/q4) > $O/${tag}_pmc_sites_fp8.txt 2>&1
cp $O/${tag}_pmc_sites_fp8.json $R/profiles/${tag}_pmc_sites_fp8.json
fi
if want pmclc; then
# long-context configuration (configs[3]): fabric traffic only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p6 -- $PB $LC --launch-log /tmp/log6.json > /tmp/p6.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p7 -- $PB $LC --launch-log /tmp/log7.json > /tmp/p7.log 2>&1
python3 $R/tools/pmc_sites.py /tmp/log6.json $O/${tag}_pmc_sites_longctx.json $(cc /tmp/p6) $(cc /tmp/p7) > $O/${tag}_pmc_sites_longctx.txt 2>&1
cp $O/${tag}_pmc_sites_longctx.json $R/profiles/${tag}_pmc_sites_longctx.json
fi
if want pmcpq; then
# the reference's own PlotQA shape (bench.py --workload plotqa-real: B 80, V 44, T 124, F_v 1024): SQ pass + fabric traffic
PQ="--workload plotqa-real"
rm -rf /tmp/r1 /tmp/r3 /tmp/r4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d /tmp/r1 -- $PB $PQ --launch-log /tmp/logr1.json > /tmp/r1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/r3 -- $PB $PQ --launch-log /tmp/logr3.json > /tmp/r3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/r4 -- $PB $PQ --launch-log /tmp/logr4.json > /tmp/r4.log 2>&1
python3 $R/tools/pmc_sites.py /tmp/logr1.json $O/${tag}_pmc_sites_plotqa_real.json $(cc /tmp/r1) $(cc /tmp/r3) $(cc /tmp/r4) > $O/${tag}_pmc_sites_plotqa_real.txt 2>&1
cp $O/${tag}_pmc_sites_plotqa_real.json $R/profiles/${tag}_pmc_sites_plotqa_real.json
fi
if want stats; then
# kernel stats + timeline of the bench workload
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- $BENCH --steps 10 --warmup 3 > /tmp/ks.log 2>&1
cp $(find /tmp/ks -name "*kernel_stats.csv" | head -1) $O/${tag}_rocprof_kernel_stats_bench.csv
python3 $R/tools/trace_timeline.py $(find /tmp/ks -name "*kernel_trace.csv" | head -1) > $O/${tag}_step_timeline.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kl -- $BENCH --steps 10 --warmup 3 $LC > /tmp/kl.log 2>&1
cp $(find /tmp/kl -name "*kernel_stats.csv" | head -1) $O/${tag}_longctx_kernel_stats.csv
rm -rf /tmp/kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- $BENCH --steps 10 --warmup 3 --workload plotqa-real > /tmp/kp.log 2>&1
cp $(find /tmp/kp -name "*kernel_stats.csv" | head -1) $O/${tag}_plotqa_real_kernel_stats.csv
fi
cd $R
F="RCCL\|HIP ver\|ROCm ver\|Hostname\|Librccl\|amdgpu.ids\|socket.cpp"
if want labs; then
python tools/step_phases.py 2>&1 | grep -v "$F" > $O/${tag}_step_phases.txt
python tools/step_phases.py --exchange bf16 --stats 1 2>&1 | grep -v "$F" > $O/${tag}_step_phases_forced_exchange.txt
python tools/step_phases.py --dtype fp8 2>&1 | grep -v "$F" > $O/${tag}_step_phases_fp8.txt
CRCT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --batch 40 --no-cpu-baseline --profile-steps 0 --no-h2d-leg --sustained-s 0 2> $O/${tag}_bench_n2_shared_gpu.err | grep '^{' > $O/${tag}_bench_n2_shared_gpu.json
python tools/lab/wgrad_fp8_lab.py 2>&1 | grep -v "$F" > $O/${tag}_wgrad_fp8_lab.txt
python tools/lab/fp8_draws.py 2>&1 | grep -v "$F" > $O/${tag}_fp8_parity_draws.txt
./tools/lab/tr8_probe.bin > $O/${tag}_tr8_probe.txt 2>&1
fi
if want bench; then
python bench.py --steps 20 --warmup 5 2> $O/${tag}_bench_n1.err | grep '^{' > $O/${tag}_bench_n1.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --no-cpu-baseline 2> $O/${tag}_bench_n1_fp8.err | grep '^{' > $O/${tag}_bench_n1_fp8.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --fp8-bf16-wgrad --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_fp8_bf16_wgrad.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --fp8-forward-only --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_fp8_forward_only.json
python bench.py --steps 20 --warmup 5 --dtype fp8 --fp8-bf16-forward --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_fp8_bf16_forward.json
python bench.py --steps 20 --warmup 5 $LC --no-cpu-baseline 2> $O/${tag}_bench_n1_longctx.err | grep '^{' > $O/${tag}_bench_n1_longctx.json
python bench.py --steps 20 --warmup 5 $LC --dtype fp8 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_longctx_fp8.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange --exchange-pack-all 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_forced_exchange_pack_all.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange --dtype fp8 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_forced_exchange_fp8.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange 2> $O/${tag}_bench_n1_forced_exchange.err | grep '^{' > $O/${tag}_bench_n1_forced_exchange.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange --grad-dtype fp32 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_forced_exchange_fp32.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-exchange --ghost-ranks 8 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_ghost8.json
python bench.py --steps 20 --warmup 5 --workload plotqa-real 2> $O/${tag}_bench_n1_plotqa_real.err | grep '^{' > $O/${tag}_bench_n1_plotqa_real.json
# the residual stream stored as bf16 (rounds 1 - 5; params['residual_fp32'] = False): the speed the default gives up for its gradient fidelity
python bench.py --steps 20 --warmup 5 --residual-bf16 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_residual_bf16.json
python bench.py --steps 20 --warmup 5 --residual-bf16 --no-cpu-baseline --workload plotqa-real 2>/dev/null | grep '^{' > $O/${tag}_bench_n1_plotqa_real_residual_bf16.json
for f in bench_n1 bench_n1_fp8 bench_n1_fp8_bf16_wgrad bench_n1_fp8_forward_only bench_n1_fp8_bf16_forward bench_n1_longctx bench_n1_longctx_fp8 bench_n1_forced_exchange bench_n1_forced_exchange_pack_all bench_n1_forced_exchange_fp8 bench_n1_forced_exchange_fp32 bench_n1_ghost8 bench_n1_plotqa_real bench_n1_residual_bf16 bench_n1_plotqa_real_residual_bf16; do python -c "import json; d=json.load(open('$O/${tag}_$f.json')); print('$f', round(d['ms_per_step'],3), round(d['value']), 'ffn frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'])"; done
fi
head -45 $O/${tag}_pmc_sites.txt
