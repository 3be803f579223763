# A/B of two builds of the library on the same box: tools/libcrct_alt.so against the in-tree one (developer script)
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
cp cqa-crct_amd/crct/libcrct_hip.so /tmp/main.so
for i in 1 2 3; do
  cp /tmp/main.so cqa-crct_amd/crct/libcrct_hip.so; run main
  cp tools/libcrct_alt.so cqa-crct_amd/crct/libcrct_hip.so; run alt
done
if [ "$1" = "test-alt" ]; then python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or wgrad" 2>&1 | tail -2; fi
cp /tmp/main.so cqa-crct_amd/crct/libcrct_hip.so
