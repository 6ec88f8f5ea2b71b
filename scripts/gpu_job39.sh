#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tgt in grad config2_long_gap_grid config4_value_and_grad; do
  echo "== prof $tgt"; bash scripts/prof_r02.sh r03_e_$tgt $tgt 2>&1 | tail -1 | cut -c1-300
done
