cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name shape tile generic
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/pmc_$1 -- ./tools/gemm_lab.bin 3 $2 $3 $4 > /dev/null 2>&1
  done
}
run ffnup_t3_gen 1 3 1
run ffnup_t1_gen 1 1 1
run ffnup_t0_pipe 1 0 0
run big_t1_gen 13 1 1
ls gpurun_out/pmc_ffnup_t3_gen/*/ | head
