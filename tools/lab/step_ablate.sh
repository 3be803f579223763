#!/bin/bash
# tools/lab/step_ablate.py with the LAB build of the library in the package's place (restored afterwards)
cp cqa-crct_amd/crct/libcrct_hip.so /tmp/main_lib.so
cp tools/lab/libcrct_hip.so cqa-crct_amd/crct/libcrct_hip.so
for d in 0 2 1 3 64 128 192 0; do CRCT_GEMM_DBG=$d python tools/lab/step_ablate.py 30 2>&1 | grep "ms per step"; done
cp /tmp/main_lib.so cqa-crct_amd/crct/libcrct_hip.so
