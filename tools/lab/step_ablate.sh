#!/bin/bash
# The training step (configs[1]) on the LAB build of the library (make -C cqa-crct_amd/csrc lab -> tools/lab/libcrct_lab.so; wrong-result
# ablations, TIMING ONLY): CRCT_GEMM_DBG bits 1 = no epilogue, 2 = no operand DMA after the prologue, 64 = no activation / dropout arithmetic,
# 128 = no side inputs / outputs.  The lab library is loaded through tools/lab/step_time.py --lib for the child process only: the package's
# library is never replaced.
set -u
cd "$(dirname "$0")/../.."
for d in 0 2 1 3 64 128 192 0; do echo -n "CRCT_GEMM_DBG=$d: "; CRCT_GEMM_DBG=$d python tools/lab/step_time.py --lib tools/lab/libcrct_lab.so --reps 2 2>&1 | grep step_time | sed 's/.*libcrct_lab.so: //'; done
