# The rocprofv3 passes of round 4 (on the GPU box: bash tools/prof_round4.sh): kernel trace of the bench; per optimiser step FETCH_SIZE /
# WRITE_SIZE of the GEMM-class launches in fp32, bf16 storage and f16x2; the env step at 65 536 envs (one clip / the 43-clip library).
# Counters in their own passes (--pmc with --kernel-trace only), the program itself behind `--`.
set -x
R=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/q_bench -- python3 bench.py --steps 3 --warmup 1 --no-alt --no-cpu-baseline > $R/q_bench.log 2>&1
ARGS=""
for P in fp32 bf16 f16x2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/q_trace_$P -- python3 tools/gemm_step_replay.py $P > $R/q_trace_$P.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/q_fetch_$P -- python3 tools/gemm_step_replay.py $P > $R/q_fetch_$P.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/q_write_$P -- python3 tools/gemm_step_replay.py $P > $R/q_write_$P.log 2>&1
  ARGS="$ARGS trace_$P=$R/q_trace_$P fetch_$P=$R/q_fetch_$P write_$P=$R/q_write_$P"
done
python3 tools/pmc_summary.py $R/r04_gemm_step_pmc.json $ARGS
EARGS=""
for M in 1x3600 43x3600; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/q_env_trace_$M -- python3 tools/env_bench.py 65536 --motion=synthetic:$M > $R/q_env_trace_$M.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/q_env_fetch_$M -- python3 tools/env_bench.py 65536 --motion=synthetic:$M > $R/q_env_fetch_$M.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/q_env_write_$M -- python3 tools/env_bench.py 65536 --motion=synthetic:$M > $R/q_env_write_$M.log 2>&1
  EARGS="$EARGS trace_$M=$R/q_env_trace_$M fetch_$M=$R/q_env_fetch_$M write_$M=$R/q_env_write_$M"
done
python3 tools/pmc_summary.py $R/r04_env_step_pmc.json $EARGS
find $R/q_bench -name "*kernel_stats.csv" -exec cp {} $R/r04_bench_kernel_stats.csv \;
find $R -name "*.db" -delete; find $R -name "*agent_info.csv" -delete
find $R -path "*q_*" -name "*.csv" -size +3M -delete
du -sh $R | tail -1
