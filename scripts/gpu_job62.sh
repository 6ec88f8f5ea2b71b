#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== bench"; timeout 900 python bench.py > gpurun_out/r03_g_bench.json 2> gpurun_out/r03_g_bench.err; echo "bench rc $?"; tail -2 gpurun_out/r03_g_bench.err
for tgt in bench config3 config4 config5 config2_with_smoother grad config2_long_gap_grid config4_value_and_grad; do
  echo "== prof $tgt"; bash scripts/prof_r02.sh r03_g_$tgt $tgt 2>&1 | tail -1 | cut -c1-200
done
