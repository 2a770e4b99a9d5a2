# PMC passes of the split kernels (gemm_split.hip) on the large layer shape: where the cycles go under the exact bf16 split (precision 3),
# the two-chunk one (2) and the fp16 two-way split (4).  usage (on the GPU box): bash tools/split_pmc.sh
set -x
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
ARGS=""
for P in 3 4 2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/p_split_a_$P -- python3 tools/gemm_one.py $P 16384 1024 1024 1 1 2 > $R/p_split_a_$P.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/p_split_b_$P -- python3 tools/gemm_one.py $P 16384 1024 1024 1 1 2 > $R/p_split_b_$P.log 2>&1
  ARGS="$ARGS a$P=$R/p_split_a_$P b$P=$R/p_split_b_$P"
done
python3 tools/pmc_summary.py $R/r04_split_pmc.json $ARGS
find $R -name "*.db" -delete; find $R -name "*agent_info.csv" -delete
