#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== bench"; timeout 900 python bench.py > gpurun_out/r03_d_bench.json 2> gpurun_out/r03_d_bench.err; echo "bench rc $?"; tail -2 gpurun_out/r03_d_bench.err
for tgt in bench config3 config4 config5 config2_with_smoother; do
  echo "== prof $tgt"; bash scripts/prof_r02.sh r03_d_$tgt $tgt 2>&1 | tail -1 | cut -c1-200
done
