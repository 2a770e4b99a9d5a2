set -x
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/p_bench -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > $R/p_bench.log 2>&1
for P in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/p_trace_$P -- python3 tools/gemm_step_replay.py $P > $R/p_trace_$P.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/p_fetch_$P -- python3 tools/gemm_step_replay.py $P > $R/p_fetch_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/p_write_$P -- python3 tools/gemm_step_replay.py $P > $R/p_write_$P.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/p_mfma_bf16 -- python3 tools/gemm_bf16_one.py 16384 1024 1024 1 1 2 > $R/p_mfma_bf16.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/p_mfma_fp32 -- python3 tools/gemm_one.py 0 > $R/p_mfma_fp32.log 2>&1
python3 tools/pmc_summary.py $R/r03_gemm_step_pmc.json trace_fp32=$R/p_trace_fp32 fetch_fp32=$R/p_fetch_fp32 write_fp32=$R/p_write_fp32 trace_bf16=$R/p_trace_bf16 fetch_bf16=$R/p_fetch_bf16 write_bf16=$R/p_write_bf16
python3 tools/pmc_summary.py $R/r03_gemm_mfma_util.json bf16=$R/p_mfma_bf16 fp32=$R/p_mfma_fp32
find $R/p_bench -name "*kernel_stats.csv" -exec cp {} $R/r03_bench_kernel_stats.csv \;
find $R -name "*.db" -delete; find $R -name "*agent_info.csv" -delete
du -sh $R | tail -1
