# FETCH_SIZE / WRITE_SIZE of the bf16-storage optimiser step at BASELINE configs[2]'s 16 384 envs (65 536-row minibatch)
set -x
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/q2_trace -- python3 tools/gemm_step_replay.py bf16 16384 > $R/q2_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/q2_fetch -- python3 tools/gemm_step_replay.py bf16 16384 > $R/q2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/q2_write -- python3 tools/gemm_step_replay.py bf16 16384 > $R/q2_write.log 2>&1
python3 tools/pmc_summary.py $R/r04_gemm_step_cfg2_pmc.json trace=$R/q2_trace fetch=$R/q2_fetch write=$R/q2_write
find $R -name "*.db" -delete; find $R -name "*agent_info.csv" -delete
