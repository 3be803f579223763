# developer A/B of GEMM shape-class configurations on the whole step (same box, same process conditions)
run() { CRCT_GEMM_CLS="$2" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), d['config']['final_loss'])"; }
OLD="tw=12,tn=12,tnl=15,vw=12,vm=12,vml=4"
run old "$OLD"
run tn15 "tw=12,tn=15,tnl=15,vw=12,vm=12,vml=15"
run vm15 "tw=12,tn=12,tnl=15,vw=12,vm=15,vml=15"
run tw15 "tw=15,tn=12,tnl=15,vw=12,vm=12,vml=15"
run vw15 "tw=12,tn=12,tnl=15,vw=15,vm=12,vml=15"
run all15 "tw=15,tn=15,tnl=15,vw=15,vm=15,vml=15"
run tn3 "tw=12,tn=3,tnl=15,vw=12,vm=12,vml=15"
run old2 "$OLD"
run tnvm15 "tw=12,tn=15,tnl=15,vw=12,vm=15,vml=15"
