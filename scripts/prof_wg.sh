#!/bin/bash
# Usage (on the GPU box, via gpurun): scripts/prof_wg.sh <tag>
# rocprofv3 kernel trace + stats of the per-GPU slices of BASELINE configs 4 and 5 (scripts/gpu_time_wg.py).
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wg -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_wg.py > $OUT/wg.log 2> $OUT/wg_err.log
find $OUT -name "*kernel_stats.csv" | head
