# PMC passes of the rigid-body step kernel (4096 envs, four lanes per env): instruction mix and where the waves wait
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/p_rigid_a -- python3 tools/rigid_bench.py > $R/p_rigid_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/p_rigid_b -- python3 tools/rigid_bench.py > $R/p_rigid_b.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $R/p_rigid_c -- python3 tools/rigid_bench.py > $R/p_rigid_c.log 2>&1
python3 tools/pmc_summary.py $R/r03_rigid_pmc.json a=$R/p_rigid_a b=$R/p_rigid_b c=$R/p_rigid_c
find $R -name "*.db" -delete; find $R -name "*agent_info.csv" -delete
