#!/bin/bash
# tools/lab/step_ablate.py with the LAB build of the library (wrong-result ablations, CRCT_GEMM_DBG) in the package's place.  The
# product library is restored on EVERY exit path (trap), so an interrupted run cannot leave the lab build installed.
set -u
backup=$(mktemp /tmp/crct_main_lib.XXXXXX.so)
cp cqa-crct_amd/crct/libcrct_hip.so "$backup"
trap 'cp "$backup" cqa-crct_amd/crct/libcrct_hip.so; rm -f "$backup"' EXIT
cp tools/lab/libcrct_hip.so cqa-crct_amd/crct/libcrct_hip.so
for d in 0 2 1 3 64 128 192 0; do CRCT_GEMM_DBG=$d python tools/lab/step_ablate.py 30 2>&1 | grep "ms per step"; done
