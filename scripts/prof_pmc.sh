#!/bin/bash
# Usage (on the GPU box, via gpurun): scripts/prof_pmc.sh <tag>
# Kernel trace + stats in one run, then FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
# (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass).
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-saturation > $OUT/trace_bench.json 2> $OUT/trace_err.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-saturation > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch_err.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-saturation > $OUT/pmc_write_bench.json 2> $OUT/pmc_write_err.log
find $OUT -name "*.csv" | head -20
