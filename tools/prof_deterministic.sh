#!/bin/bash
# kernel-time breakdown of the update step with agent.deterministic off / on (GPU box): gpurun_out/det_{0,1}_stats.csv
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1; do
  DET_ONLY=$d rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/det_$d -o det -- python3 $R/tools/deterministic_cost.py fp32 > $R/gpurun_out/det_$d.log 2>&1
  f=$(find $R/gpurun_out/det_$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $R/gpurun_out/det_${d}_stats.csv
done
