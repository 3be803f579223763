# usage (on the GPU box): bash tools/pmc_gemm.sh <name> <shape index> <tile> ; prints per-kernel counter sums
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shape=$2; tile=$3
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/pmc_$name -- ./tools/gemm_lab.bin 3 $shape $tile 0 > /dev/null 2>&1
done
python3 - gpurun_out/pmc_$name <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(sys.argv[1], {k: round(v / max(n[k], 1)) for k, v in sorted(acc.items())})
PY
rm -rf gpurun_out/pmc_$name
