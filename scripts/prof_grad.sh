#!/bin/bash
# Usage (on the GPU box, via gpurun): scripts/prof_grad.sh <tag>
# rocprofv3 kernel trace + stats of the gradient paths: forward sensitivities at the C2 shape (gpu_time_grad.py) and the
# forward + reverse sweep pair at the config-5 slice (gpu_time_adjoint.py).
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sens -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_grad.py > $OUT/sens.log 2> $OUT/sens_err.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/adjoint -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_adjoint.py 1024 1000 > $OUT/adjoint.log 2> $OUT/adjoint_err.log
find $OUT -name "*kernel_stats.csv" | head
